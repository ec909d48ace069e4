// resample.hip -- torchaudio.transforms.Resample(orig_freq, new_freq) on gfx950 (at_resample_f32).
//
// Replaces SpectrogramGenerator.resample (processors/spectrogram_generator.py:117-121 of
// danavery/audio-tokens): torchaudio 2.4.1 Resample with its defaults -- resampling_method
// "sinc_interp_hann", lowpass_filter_width = 6, rolloff = 0.99 -- i.e. a polyphase FIR: with
// orig/new reduced by their gcd, output sample i*new + j is the dot product of the j-th filter
// (2*width + orig taps) with the input window starting at i*orig - width (zero padded).
// HBM-bound in principle (4 B in + 4*new/orig B out per input sample); every input sample is
// re-read new*K/orig times from L1/L2.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "at_internal.h"

namespace {

constexpr int WG = 256;

__global__ void __launch_bounds__(WG) resample_kernel(const float* __restrict__ wave, long n_clips, long L,
                                                      long wave_stride, const float* __restrict__ taps, int orig,
                                                      int nw, int K, int width, long out_len, long out_stride,
                                                      float* __restrict__ out) {
    const long o = (long)blockIdx.x * WG + threadIdx.x;
    const long clip = blockIdx.y;
    if (o >= out_len) return;
    const long i = o / nw;
    const int j = (int)(o - i * nw);
    const float* w = wave + clip * wave_stride;
    const float* t = taps + j;  // device taps are stored [K][new]: consecutive lanes, consecutive phases
    const long s0 = i * orig - width;
    float acc = 0.0f;
    for (int k = 0; k < K; k++) {
        const long s = s0 + k;
        const float v = (s >= 0 && s < L) ? w[s] : 0.0f;
        acc = __builtin_fmaf(v, t[(size_t)k * nw], acc);
    }
    out[clip * out_stride + o] = acc;
}

// Tiled variant: the input span of TI consecutive i-steps is staged in LDS once and every thread
// keeps RI accumulators for one phase j, so a tap fetched from L1 feeds RI FMAs.  When the number
// of phases is large the RI steps of a thread are adjacent (all lanes of a wave then read the same
// LDS word: a broadcast); when it is small (44.1 kHz -> 22.05 kHz has ONE phase) they are
// interleaved so that consecutive lanes read consecutive steps.  The accumulation order is the
// same ascending-k fma chain as resample_kernel, so both produce identical bits.
template <int RI>
__global__ void __launch_bounds__(WG) resample_tiled_kernel(const float* __restrict__ wave, long L, long wave_stride,
                                                            const float* __restrict__ taps, int orig, int nw, int K,
                                                            int width, int TI, int interleave, long out_len,
                                                            long out_stride, float* __restrict__ out) {
    extern __shared__ float seg[];
    const long clip = blockIdx.y;
    const long i0 = (long)blockIdx.x * TI;
    const float* w = wave + clip * wave_stride;
    const int span = (TI - 1) * orig + K;
    const long s_base = i0 * orig - width;
    // eight independent (clamped, then masked) loads in flight per thread: the staging is otherwise
    // one exposed HBM latency per 1 KiB
    for (int t0 = threadIdx.x; t0 < span; t0 += WG * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const long s = s_base + t0 + u * WG;
            const long sc = s < 0 ? 0 : (s >= L ? L - 1 : s);
            v[u] = w[sc];
            if (s != sc) v[u] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (t0 + u * WG < span) seg[t0 + u * WG] = v[u];
    }
    __syncthreads();
    const int nblk = TI / RI;
    const int ntask = nw * nblk;
    float* o = out + clip * out_stride;
    for (int q = threadIdx.x; q < ntask; q += WG) {
        const int ib = q / nw;
        const int j = q - ib * nw;
        int li[RI];
#pragma unroll
        for (int r = 0; r < RI; r++) li[r] = interleave ? r * nblk + ib : ib * RI + r;
        float acc[RI];
#pragma unroll
        for (int r = 0; r < RI; r++) acc[r] = 0.0f;
        const float* tp = taps + j;
#pragma unroll 4
        for (int k = 0; k < K; k++) {
            const float t = tp[(size_t)k * nw];
#pragma unroll
            for (int r = 0; r < RI; r++) acc[r] = __builtin_fmaf(seg[li[r] * orig + k], t, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const long oi = (i0 + li[r]) * nw + j;
            if (oi < out_len) o[oi] = acc[r];
        }
    }
}

}  // namespace

extern "C" {

// torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann), evaluated in double and
// rounded to float as torchaudio does.  taps_host: [new][2*width + orig]; returns width via *width.
int at_resample_taps_host(int orig_freq, int new_freq, int* orig_out, int* new_out, int* width_out,
                          float* taps_host, int64_t taps_capacity) {
    AT_REQUIRE(orig_freq > 0 && new_freq > 0 && orig_out && new_out && width_out, "at_resample_taps_host: bad arguments");
    int a = orig_freq, b = new_freq;
    while (b) { const int t = a % b; a = b; b = t; }
    const int g = a;
    const int orig = orig_freq / g, nw = new_freq / g;
    const double lowpass_filter_width = 6.0, rolloff = 0.99;
    const double base_freq = std::fmin((double)orig, (double)nw) * rolloff;
    const int width = (int)std::ceil(lowpass_filter_width * orig / base_freq);
    const int K = 2 * width + orig;
    *orig_out = orig; *new_out = nw; *width_out = width;
    if (!taps_host) return AT_OK;  // size query
    AT_REQUIRE(taps_capacity >= (int64_t)nw * K, "at_resample_taps_host: taps buffer too small");
    const double scale = base_freq / orig;
    for (int j = 0; j < nw; j++) {
        for (int k = 0; k < K; k++) {
            double t = (double)(-j) / nw + (double)(k - width) / orig;
            t *= base_freq;
            if (t < -lowpass_filter_width) t = -lowpass_filter_width;
            if (t > lowpass_filter_width) t = lowpass_filter_width;
            const double c = std::cos(t * M_PI / lowpass_filter_width / 2.0);
            const double window = c * c;
            t *= M_PI;
            const double sinc = t == 0.0 ? 1.0 : std::sin(t) / t;
            taps_host[(size_t)j * K + k] = (float)(sinc * window * scale);
        }
    }
    return AT_OK;
}

int64_t at_resample_length(int64_t L, int orig_freq, int new_freq) {
    int a = orig_freq, b = new_freq;
    while (b) { const int t = a % b; a = b; b = t; }
    const int64_t orig = orig_freq / a, nw = new_freq / a;
    return (nw * L + orig - 1) / orig;  // ceil(new * length / orig)
}

int at_resample_f32(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride,
                    int orig_freq, int new_freq, float* out, int64_t out_stride, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && orig_freq > 0 && new_freq > 0 && n_clips >= 0 && L > 0, "at_resample_f32: bad arguments");
    if (n_clips == 0) return AT_OK;
    AT_REQUIRE(wave && out && wave_stride >= L && n_clips <= 65535, "at_resample_f32: bad arguments");
    AT_HIP(hipSetDevice(ctx->device));
    int orig = 0, nw = 0, width = 0;
    int rc = at_resample_taps_host(orig_freq, new_freq, &orig, &nw, &width, nullptr, 0);
    if (rc) return rc;
    const int K = 2 * width + orig;
    const int64_t out_len = at_resample_length(L, orig_freq, new_freq);
    AT_REQUIRE(out_stride >= out_len, "at_resample_f32: out_stride < output length %lld", (long long)out_len);
    if (!(ctx->rs_orig == orig_freq && ctx->rs_new == new_freq && ctx->ws[WS_RESAMPLE_TAPS])) {
        std::vector<float> taps((size_t)nw * K);
        rc = at_resample_taps_host(orig_freq, new_freq, &orig, &nw, &width, taps.data(), (int64_t)taps.size());
        if (rc) return rc;
        std::vector<float> tr((size_t)nw * K);
        for (int j = 0; j < nw; j++)
            for (int k = 0; k < K; k++) tr[(size_t)k * nw + j] = taps[(size_t)j * K + k];
        float* dev = static_cast<float*>(at_ws(ctx, WS_RESAMPLE_TAPS, tr.size() * sizeof(float), stream));
        if (!dev) return AT_E_NOMEM;
        AT_HIP(hipStreamSynchronize(stream));
        AT_HIP(hipMemcpy(dev, tr.data(), tr.size() * sizeof(float), hipMemcpyHostToDevice));
        ctx->rs_orig = orig_freq; ctx->rs_new = new_freq;
    }
    const float* taps = static_cast<const float*>(ctx->ws[WS_RESAMPLE_TAPS]);
    constexpr int RI = 4;
    constexpr int kSegFloats = 8192;  // 32 KiB of LDS per workgroup
    const bool force_simple = ctx->dbg.resample_simple != 0;  // test switch
    const long fit = ((long)kSegFloats - K) / orig + 1;  // i-steps whose input span fits the segment
    if (fit >= RI && !force_simple) {
        const long n_i = (out_len + nw - 1) / nw;
        long TI = fit - fit % RI;
        const long cap = ((n_i + RI - 1) / RI) * RI;  // no point in tiles longer than the clip
        if (TI > cap) TI = cap;
        const size_t lds = sizeof(float) * (size_t)((TI - 1) * orig + K);
        AT_LAUNCH(resample_tiled_kernel<RI>, dim3((unsigned)((n_i + TI - 1) / TI), (unsigned)n_clips),
                           dim3(WG), lds, stream, wave, (long)L, (long)wave_stride, taps, orig, nw, K, width, (int)TI,
                           nw < 32 ? 1 : 0, (long)out_len, (long)out_stride, out);
    } else {
        AT_LAUNCH(resample_kernel, dim3((unsigned)((out_len + WG - 1) / WG), (unsigned)n_clips), dim3(WG),
                           0, stream, wave, (long)n_clips, (long)L, (long)wave_stride, taps, orig, nw, K, width,
                           (long)out_len, (long)out_stride, out);
    }
    return AT_OK;
}

}  // extern "C"
