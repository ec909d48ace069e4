// logmel_any.hip -- at_logmel_f32 for every power-of-two n_fft other than 512 (64 .. 4096).
//
// The reference exposes n_fft / hop_length as configuration (audio_tokens_config.py:39-40; its README
// documents 1024 / 512) and hands them to torchaudio's MelSpectrogram (processors/spectrogram_generator.py:28-33).
// The tuned kernel (logmel.hip) is built around the code default, n_fft = 512 -- two 16-point stages in registers.
// This file is the general form: one wavefront per frame, the M = n_fft/2-point complex FFT of z[m] = x[2m] + i x[2m+1]
// as Stockham passes of radix 8 (then 4 or 2 for what is left of log2 M) with the butterflies in registers: three
// passes and two trips through LDS at n_fft = 1024 (round 2 did nine radix-2 stages, each a trip through LDS, with
// window and twiddles read from global memory); the first pass takes the windowed samples straight from global
// memory, twiddle and window tables sit in LDS.  Then the same even/odd untangling to the M + 1 power bins, the banded
// mel dot products, 10 log10.  Same arithmetic contract as the tuned kernel (fp32 throughout, |X|^2 as re^2 + im^2,
// clamp at 1e-10), same tolerance against the CPU restatement (tests/test_gpu_ops.py::test_logmel_other_nfft).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "at_internal.h"

namespace {

constexpr int WG = 256;

struct AnyParams {
    const float* wave;
    long n_clips, L, wave_stride;
    int n_fft, log2m, hop, T, n_mels;
    long n_frames;           // n_clips * T
    const float* win;        // n_fft
    const float* twm;        // M x (cos, -sin) of 2*pi*q/M
    const float* twn;        // M x (cos, -sin) of 2*pi*k/n_fft
    const int* fb_start;     // [n_mels] first bin / number of bins / offset into fb_wts
    const int* fb_len;
    const int* fb_off;
    const float* fb_wts;
    float* out;
    int frame_major;
};

struct cx {
    float re, im;
};
__device__ __forceinline__ cx cadd(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cx csub(cx a, cx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cx cmul(cx a, cx w) {
    return {__builtin_fmaf(a.re, w.re, -(a.im * w.im)), __builtin_fmaf(a.re, w.im, a.im * w.re)};
}
// forward DFTs in registers, natural order in and out
__device__ __forceinline__ void dft2(cx (&v)[2]) {
    const cx a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
__device__ __forceinline__ void dft4(cx& a, cx& b, cx& c, cx& d) {
    const cx t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = {t1.re + t3.im, t1.im - t3.re};  // t1 - i*t3
    d = {t1.re - t3.im, t1.im + t3.re};  // t1 + i*t3
}
__device__ __forceinline__ void dft4(cx (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void dft8(cx (&v)[8]) {
    constexpr float R2 = 0.70710678118654752f;
    dft4(v[0], v[2], v[4], v[6]);   // E[0..3] left at v[0], v[2], v[4], v[6]
    dft4(v[1], v[3], v[5], v[7]);   // O[0..3] left at v[1], v[3], v[5], v[7]
    const cx o0 = v[1];
    const cx o1 = {R2 * (v[3].re + v[3].im), R2 * (v[3].im - v[3].re)};      // * W8^1 = (1 - i)/sqrt 2
    const cx o2 = {v[5].im, -v[5].re};                                        // * W8^2 = -i
    const cx o3 = {R2 * (v[7].im - v[7].re), -R2 * (v[7].re + v[7].im)};     // * W8^3 = (-1 - i)/sqrt 2
    const cx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <int R>
__device__ __forceinline__ void dftR(cx (&v)[R]) {
    if constexpr (R == 8) dft8(v);
    else if constexpr (R == 4) dft4(v);
    else dft2(v);
}

// One Stockham pass of radix R over M points held by one wavefront: butterfly j (of M/R) takes the inputs
// j + t*M/R, multiplies input t by W_(NS*R)^(k*t) with k = j mod NS (NS = product of the radices before this pass),
// and leaves the R outputs at (j - k)*R + k + t*NS.  Natural order in, natural order out after the last pass.
// In place: a wavefront's LDS instructions execute in order, and every lane has read all its inputs into registers
// before the first store of the pass is issued (the wave barriers keep the compiler from mixing the two).
template <int M, int R, int NS, bool FIRST, typename Load>
__device__ __forceinline__ void fft_pass(int lane, float* z, const float* tw, Load load) {
    constexpr int NB = M / R;
    constexpr int PER = (NB + 63) / 64;
    cx v[PER][R];
#pragma unroll
    for (int b = 0; b < PER; b++) {
        const int j = lane + 64 * b;
        if (NB >= 64 * (b + 1) || j < NB) {
#pragma unroll
            for (int t = 0; t < R; t++) {
                if constexpr (FIRST) v[b][t] = load(j + t * NB);
                else v[b][t] = {z[2 * (j + t * NB)], z[2 * (j + t * NB) + 1]};
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < PER; b++) {
        const int j = lane + 64 * b;
        if (NB >= 64 * (b + 1) || j < NB) {
            const int k = j & (NS - 1);
            if constexpr (NS > 1) {
#pragma unroll
                for (int t = 1; t < R; t++) {
                    const int q = k * t * (M / (NS * R));
                    v[b][t] = cmul(v[b][t], {tw[2 * q], tw[2 * q + 1]});
                }
            }
            dftR<R>(v[b]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; t++) {
                z[2 * (j0 + t * NS)] = v[b][t].re;
                z[2 * (j0 + t * NS) + 1] = v[b][t].im;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// radix 8 while three bits are left, then 4 or 2
template <int M, int NS, bool FIRST, typename Load>
__device__ __forceinline__ void fft_passes(int lane, float* z, const float* tw, Load load) {
    if constexpr (NS < M) {
        constexpr int left = M / NS;
        constexpr int R = left >= 8 ? 8 : left;
        fft_pass<M, R, NS, FIRST>(lane, z, tw, load);
        fft_passes<M, NS * R, false>(lane, z, tw, load);
    }
}

template <int LOG2M>
__global__ void __launch_bounds__(WG) logmel_any_kernel(AnyParams p) {
    constexpr int M = 1 << LOG2M, N = 2 * M;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // W_M (2M) | W_N (2M) | window (N) | per wave: z (N) + power (M + 4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tw = sm;
    float* twn = sm + 2 * M;
    float* win = sm + 4 * M;
    for (int i = threadIdx.x; i < 2 * M; i += WG) {
        tw[i] = p.twm[i];
        twn[i] = p.twn[i];
        win[i] = p.win[i];
    }
    __syncthreads();
    float* z = sm + 6 * M + (size_t)wave * (N + M + 4);
    float* pw = z + N;
    for (long g = (long)blockIdx.x * (WG / 64) + wave; g < p.n_frames; g += (long)gridDim.x * (WG / 64)) {
        const long clip = g / p.T;
        const int t = (int)(g - clip * p.T);
        const float* w = p.wave + clip * p.wave_stride;
        const long s0 = (long)t * p.hop - M;              // center=True: n_fft/2 samples of reflection on each side
        const bool inner = s0 >= 0 && s0 + N <= p.L;      // no reflection anywhere in this frame
        // complex point m = (x[2m], x[2m+1]) of the windowed frame (torch: frames * window, fp32)
        auto load = [&](int m) -> cx {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                long q = s0 + 2 * m + e;
                if (!inner) {
                    if (q < 0) q = -q;                      // reflect, no edge repeat
                    if (q >= p.L) q = 2 * (p.L - 1) - q;
                    if (q < 0) q = 0;
                    if (q >= p.L) q = p.L - 1;
                }
                v[e] = w[q] * win[2 * m + e];
            }
            return {v[0], v[1]};
        };
        fft_passes<M, 1, true>(lane, z, tw, load);
        // even / odd untangling: X[k] = Ev + W_N^k * Od, Ev = (Z[k] + conj Z[M-k]) / 2, Od = -i (Z[k] - conj Z[M-k]) / 2
        for (int k = lane; k < M; k += 64) {
            const int kk = (M - k) & (M - 1);
            const float ar = z[2 * k], ai = z[2 * k + 1];
            const float br = z[2 * kk], bi = -z[2 * kk + 1];
            const float evr = 0.5f * (ar + br), evi = 0.5f * (ai + bi);
            const float dfr = 0.5f * (ar - br), dfi = 0.5f * (ai - bi);
            const float odr = dfi, odi = -dfr;
            const float wr = twn[2 * k], wi = twn[2 * k + 1];
            const float xr = evr + __builtin_fmaf(odr, wr, -(odi * wi));
            const float xi = evi + __builtin_fmaf(odr, wi, odi * wr);
            pw[k] = __builtin_fmaf(xr, xr, xi * xi);
            if (k == 0) {
                const float nq = ar - ai;                   // X[n_fft/2] = Re Z0 - Im Z0 (purely real)
                pw[M] = nq * nq;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int m = lane; m < p.n_mels; m += 64) {
            const float* wt = p.fb_wts + p.fb_off[m];
            const int st = p.fb_start[m], ln = p.fb_len[m];
            float acc = 0.0f;
            for (int q = 0; q < ln; q++) acc = __builtin_fmaf(pw[st + q], wt[q], acc);
            const float db = !(acc <= 1e-10f) ? 10.0f * log10f(acc) : -100.0f;   // (a NaN power stays NaN, as torch.clamp leaves it)
            if (p.frame_major) p.out[g * p.n_mels + m] = db;
            else p.out[(clip * p.n_mels + m) * p.T + t] = db;
        }
        __builtin_amdgcn_wave_barrier();   // the next frame overwrites z / pw
    }
}

template <int LOG2M>
int launch_any(at_ctx* ctx, const AnyParams& p, hipStream_t stream) {
    constexpr int M = 1 << LOG2M, N = 2 * M;
    const size_t lds = ((size_t)6 * M + (size_t)(WG / 64) * (N + M + 4)) * sizeof(float);
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&logmel_any_kernel<LOG2M>), lds); if (rcl_) return rcl_; }
    // persistent: as many workgroups as fit the LDS of the chip (at most eight per CU), each wave walking frames
    long per_cu = (long)(160 * 1024 / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    long grid = per_cu * ctx->n_cus;
    const long need = (p.n_frames + WG / 64 - 1) / (WG / 64);
    if (grid > need) grid = need;
    AT_LAUNCH(logmel_any_kernel<LOG2M>, dim3((unsigned)grid), dim3(WG), lds, stream, p);
    return AT_OK;
}

}  // namespace

// Tables for (sample_rate, n_fft, n_mels, filterbank values): window | W_M | W_N | start, len, off | band weights.
int at_logmel_any(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride, int sample_rate,
                  int n_fft, int hop, int n_mels, const float* fb_user_dev, float* out, int frame_major,
                  hipStream_t stream) {
    const int N = n_fft, M = N / 2, NBIN = M + 1;
    int log2m = 0;
    while ((1 << log2m) < M) log2m++;
    std::vector<float> fb((size_t)NBIN * n_mels);
    bool cached = ctx->ws[WS_LOGMEL_ANY] && ctx->any_sr == sample_rate && ctx->any_nfft == n_fft && ctx->any_nmels == n_mels &&
                  (ctx->any_user_copy != nullptr) == (fb_user_dev != nullptr);
    if (fb_user_dev) {   // a caller's filterbank is compared by value, never trusted by address (see logmel.hip)
        AT_HIP(hipStreamSynchronize(stream));
        AT_HIP(hipMemcpy(fb.data(), fb_user_dev, fb.size() * sizeof(float), hipMemcpyDeviceToHost));
        cached = cached && std::memcmp(ctx->any_user_copy, fb.data(), fb.size() * sizeof(float)) == 0;
    }
    const size_t nint = ((size_t)3 * n_mels + 3) & ~(size_t)3;
    const size_t head = (size_t)N + 2 * (size_t)M + 2 * (size_t)M;   // floats: window, W_M (M complex), W_N (M complex)
    const size_t cap = (head + nint + (size_t)NBIN * n_mels) * 4;
    char* base = static_cast<char*>(at_ws(ctx, WS_LOGMEL_ANY, cap, stream));
    if (!base) return AT_E_NOMEM;
    if (ctx->any_nfft != n_fft) cached = false;   // (a grown slot forgets what it held)
    if (!cached) {
        if (!fb_user_dev) {
            int rc = at_mel_filterbank_host(sample_rate, n_fft, n_mels, fb.data());
            if (rc) return rc;
        }
        std::vector<float> blob(head + nint, 0.0f);
        for (int i = 0; i < N; i++) blob[i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / N));   // periodic Hann
        for (int j = 0; j < M; j++) {
            blob[N + 2 * j] = (float)std::cos(2.0 * M_PI * j / M);
            blob[N + 2 * j + 1] = (float)-std::sin(2.0 * M_PI * j / M);
        }
        for (int k = 0; k < M; k++) {
            blob[N + 2 * M + 2 * k] = (float)std::cos(2.0 * M_PI * k / N);
            blob[N + 2 * M + 2 * k + 1] = (float)-std::sin(2.0 * M_PI * k / N);
        }
        int* ints = reinterpret_cast<int*>(blob.data() + head);
        std::vector<float> wts;
        for (int m = 0; m < n_mels; m++) {
            int lo = NBIN, hi = -1;
            for (int f = 0; f < NBIN; f++)
                if (fb[(size_t)f * n_mels + m] != 0.0f) { lo = f < lo ? f : lo; hi = f; }
            ints[m] = hi < 0 ? 0 : lo;
            ints[n_mels + m] = hi < 0 ? 0 : hi - lo + 1;
            ints[2 * n_mels + m] = (int)wts.size();
            for (int f = lo; f <= hi; f++) wts.push_back(fb[(size_t)f * n_mels + m]);
        }
        blob.insert(blob.end(), wts.begin(), wts.end());
        AT_HIP(hipDeviceSynchronize());   // a launch on any stream may still be reading the tables about to be replaced
        AT_HIP(hipMemcpy(base, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
        std::free(ctx->any_user_copy);
        ctx->any_user_copy = nullptr;
        if (fb_user_dev) {
            ctx->any_user_copy = static_cast<float*>(std::malloc(fb.size() * sizeof(float)));
            if (!ctx->any_user_copy) return at_fail(AT_E_NOMEM, "at_logmel_f32: out of host memory");
            std::memcpy(ctx->any_user_copy, fb.data(), fb.size() * sizeof(float));
        }
        ctx->any_sr = sample_rate; ctx->any_nfft = n_fft; ctx->any_nmels = n_mels;
    }
    AnyParams p;
    const float* f = reinterpret_cast<const float*>(base);
    p.wave = wave; p.n_clips = n_clips; p.L = L; p.wave_stride = wave_stride;
    p.n_fft = N; p.log2m = log2m; p.hop = hop; p.n_mels = n_mels;
    const int64_t T = at_num_frames(L, hop);
    p.T = (int)T;
    p.n_frames = n_clips * T;
    p.win = f; p.twm = f + N; p.twn = f + N + 2 * M;
    const int* ints = reinterpret_cast<const int*>(f + head);
    p.fb_start = ints; p.fb_len = ints + n_mels; p.fb_off = ints + 2 * n_mels;
    p.fb_wts = reinterpret_cast<const float*>(ints + nint);
    p.out = out; p.frame_major = frame_major;
    switch (log2m) {
        case 5: return launch_any<5>(ctx, p, stream);
        case 6: return launch_any<6>(ctx, p, stream);
        case 7: return launch_any<7>(ctx, p, stream);
        case 8: return launch_any<8>(ctx, p, stream);
        case 9: return launch_any<9>(ctx, p, stream);
        case 10: return launch_any<10>(ctx, p, stream);
        case 11: return launch_any<11>(ctx, p, stream);
    }
    return at_fail(AT_E_INVALID, "at_logmel_f32: n_fft=%d not supported", n_fft);
}
