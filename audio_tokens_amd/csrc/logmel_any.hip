// logmel_any.hip -- at_logmel_f32 for every power-of-two n_fft other than 512 (64 .. 4096).
//
// The reference exposes n_fft / hop_length as configuration (audio_tokens_config.py:39-40; its README
// documents 1024 / 512) and hands them to torchaudio's MelSpectrogram (processors/spectrogram_generator.py:28-33).
// The tuned kernel (logmel.hip) is built around the code default, n_fft = 512 -- two 16-point stages in registers.
// This file is the general form: one wavefront per frame, the n_fft/2-point complex FFT of z[m] = x[2m] + i x[2m+1]
// as radix-2 decimation-in-time stages in LDS (bit-reversed load, natural output), the same even/odd untangling to
// the n_fft/2 + 1 power bins, the banded mel dot products, 10 log10.  Same arithmetic contract as the tuned kernel
// (fp32 throughout, |X|^2 as re^2 + im^2, clamp at 1e-10), same tolerance against the CPU restatement
// (tests/test_gpu_ops.py::test_logmel_other_nfft); roughly a quarter of its speed, which is the price of a
// configuration the reference's defaults do not use.
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
    const float* twm;        // M/2 x (cos, -sin) of 2*pi*j/M
    const float* twn;        // M   x (cos, -sin) of 2*pi*k/n_fft
    const int* fb_start;     // [n_mels] first bin / number of bins / offset into fb_wts
    const int* fb_len;
    const int* fb_off;
    const float* fb_wts;
    float* out;
    int frame_major;
};

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

__global__ void __launch_bounds__(WG) logmel_any_kernel(AnyParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // per wave: n_fft floats (z) + M + 4 floats (power)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = p.n_fft, M = N >> 1;
    float* z = sm + (size_t)wave * (N + M + 4);
    float* pw = z + N;
    for (long g = (long)blockIdx.x * (WG / 64) + wave; g < p.n_frames; g += (long)gridDim.x * (WG / 64)) {
        const long clip = g / p.T;
        const int t = (int)(g - clip * p.T);
        const float* w = p.wave + clip * p.wave_stride;
        const long s0 = (long)t * p.hop - M;              // center=True: n_fft/2 samples of reflection on each side
        // windowed samples, complex point m = (x[2m], x[2m+1]), stored at the bit-reversed position
        for (int m = lane; m < M; m += 64) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                long q = s0 + 2 * m + e;
                if (q < 0) q = -q;                          // reflect, no edge repeat
                if (q >= p.L) q = 2 * (p.L - 1) - q;
                if (q < 0) q = 0;
                if (q >= p.L) q = p.L - 1;
                v[e] = w[q] * p.win[2 * m + e];             // torch: frames * window, fp32
            }
            const unsigned r = bitrev((unsigned)m, p.log2m);
            z[2 * r] = v[0];
            z[2 * r + 1] = v[1];
        }
        // radix-2 decimation in time: stage s combines blocks of half = 2^(s-1)
        for (int s = 1; s <= p.log2m; s++) {
            __builtin_amdgcn_wave_barrier();   // (a wave's LDS operations execute in order; keep the compiler from moving them)
            const int half = 1 << (s - 1);
            const int tstep = M >> s;          // twiddle index step: W_M^(pos * M / (2 half))
            for (int b = lane; b < M / 2; b += 64) {
                const int pos = b & (half - 1);
                const int i = ((b - pos) << 1) + pos, j = i + half;
                const float wr = p.twm[2 * (pos * tstep)], wi = p.twm[2 * (pos * tstep) + 1];
                const float ur = z[2 * i], ui = z[2 * i + 1];
                const float vr = z[2 * j], vi = z[2 * j + 1];
                const float tr = __builtin_fmaf(vr, wr, -(vi * wi)), ti = __builtin_fmaf(vr, wi, vi * wr);
                z[2 * i] = ur + tr;
                z[2 * i + 1] = ui + ti;
                z[2 * j] = ur - tr;
                z[2 * j + 1] = ui - ti;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // even / odd untangling: X[k] = Ev + W_N^k * Od, Ev = (Z[k] + conj Z[M-k]) / 2, Od = -i (Z[k] - conj Z[M-k]) / 2
        for (int k = lane; k < M; k += 64) {
            const int kk = (M - k) & (M - 1);
            const float ar = z[2 * k], ai = z[2 * k + 1];
            const float br = z[2 * kk], bi = -z[2 * kk + 1];
            const float evr = 0.5f * (ar + br), evi = 0.5f * (ai + bi);
            const float dfr = 0.5f * (ar - br), dfi = 0.5f * (ai - bi);
            const float odr = dfi, odi = -dfr;
            const float wr = p.twn[2 * k], wi = p.twn[2 * k + 1];
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
    const size_t head = (size_t)N + M + 2 * (size_t)M;   // floats: window, W_M (M/2 complex), W_N (M complex)
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
        for (int j = 0; j < M / 2; j++) {
            blob[N + 2 * j] = (float)std::cos(2.0 * M_PI * j / M);
            blob[N + 2 * j + 1] = (float)-std::sin(2.0 * M_PI * j / M);
        }
        for (int k = 0; k < M; k++) {
            blob[N + M + 2 * k] = (float)std::cos(2.0 * M_PI * k / N);
            blob[N + M + 2 * k + 1] = (float)-std::sin(2.0 * M_PI * k / N);
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
    p.win = f; p.twm = f + N; p.twn = f + N + M;
    const int* ints = reinterpret_cast<const int*>(f + head);
    p.fb_start = ints; p.fb_len = ints + n_mels; p.fb_off = ints + 2 * n_mels;
    p.fb_wts = reinterpret_cast<const float*>(ints + nint);
    p.out = out; p.frame_major = frame_major;
    const size_t lds = (size_t)(WG / 64) * (N + M + 4) * sizeof(float);
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&logmel_any_kernel), lds); if (rcl_) return rcl_; }
    long grid = 8L * ctx->n_cus;
    const long need = (p.n_frames + WG / 64 - 1) / (WG / 64);
    if (grid > need) grid = need;
    AT_LAUNCH(logmel_any_kernel, dim3((unsigned)grid), dim3(WG), lds, stream, p);
    return AT_OK;
}
