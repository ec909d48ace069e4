// l2norm.hip -- row L2 normalisation with numpy's exact fp32 arithmetic (at_l2norm_rows_f32).
//
// Replaces ClusterCreator.normalize_vectors (processors/cluster_creator.py:64-66) and
// SpecTokenizer.normalize_vectors (processors/spec_tokenizer.py:106-109):
//     norms = np.linalg.norm(vectors, axis=1, keepdims=True);  vectors / (norms + 1e-10)
// numpy evaluates this in fp32 as sqrt(add.reduce(x*x)) with its pairwise summation (8 running
// sums over blocks of <= 128, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), halves split at a
// multiple of 8 above 128), then one add and one IEEE division per element.  The same order is
// used here, so the result is bit-identical to numpy's.  HBM-bound: 8d bytes per row.
//
// Layout: a workgroup stages R rows in LDS (coalesced reads, row pitch d+1 so a row-per-thread
// walk is bank-conflict free), R threads reduce one row each, then all threads divide and store
// coalesced.
#include "at_internal.h"
#include "l2norm_core.h"

namespace {

constexpr int WG = 256;

// bad (optional): set to 1 when a row's norm is not finite, i.e. when the row holds a NaN or Inf (or its squares overflow)
__global__ void __launch_bounds__(WG) l2norm_rows_kernel(const float* __restrict__ x, long n, int d,
                                                         int rows_per_block, float* __restrict__ y, int* __restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // R*(d+1) floats + R floats
    const int R = rows_per_block;
    const int pitch = d + 1;
    float* den = sm + (size_t)R * pitch;
    const long row0 = (long)blockIdx.x * R;
    const int rows = (int)min((long)R, n - row0);
    const long total = (long)rows * d;
    const float* src = x + row0 * d;
    for (long e = threadIdx.x; e < total; e += WG) {
        const int r = (int)(e / d), c = (int)(e - (long)r * d);
        sm[(size_t)r * pitch + c] = src[e];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < rows; r += WG) {
        den[r] = l2n::row_denominator(sm + (size_t)r * pitch, d, 1);
        if (bad && !(den[r] < __builtin_inff())) atomicOr(bad, 1);
    }
    __syncthreads();
    float* dst = y + row0 * d;
    for (long e = threadIdx.x; e < total; e += WG) {
        const int r = (int)(e / d), c = (int)(e - (long)r * d);
        dst[e] = l2n::divide(sm[(size_t)r * pitch + c], den[r]);
    }
}

// Per-clip min-max scaling of a spectrogram (SpectrogramGenerator.normalize_spectrogram,
// processors/spectrogram_generator.py:129-131): (spec - min) / (max - min) with torch's fp32 operations -- two
// subtractions and one IEEE division per element.  One workgroup per clip: the reduction pass pulls the clip
// (441 KB at 64 x 1723) through L2, the scaling pass reads it from there.  NaN propagates as in torch.min / max.
__global__ void __launch_bounds__(1024) minmax_scale_kernel(float* __restrict__ x, long clip_elems) {
    __shared__ float s_lo[16], s_hi[16];
    __shared__ int s_nan[16];
    float* p = x + (size_t)blockIdx.x * clip_elems;
    const int t = threadIdx.x;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    int nan = 0;
    for (long e = t; e < clip_elems; e += 1024) {
        const float v = p[e];
        nan |= v != v;
        lo = __builtin_fminf(lo, v);
        hi = __builtin_fmaxf(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, off));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, off));
        nan |= __shfl_xor(nan, off);
    }
    if ((t & 63) == 0) { s_lo[t >> 6] = lo; s_hi[t >> 6] = hi; s_nan[t >> 6] = nan; }
    __syncthreads();
    for (int w = 0; w < 16; w++) {
        lo = __builtin_fminf(lo, s_lo[w]);
        hi = __builtin_fmaxf(hi, s_hi[w]);
        nan |= s_nan[w];
    }
    if (nan) lo = hi = __builtin_nanf("");
    const float range = hi - lo;
    for (long e = t; e < clip_elems; e += 1024) p[e] = __fdiv_rn(p[e] - lo, range);
}

}  // namespace

namespace {
// ClusterCreator / SpecTokenizer.apply_convolution (processors/cluster_creator.py:68-81, spec_tokenizer.py:92-104,115-121):
// nn.Conv1d(1, num_kernels, kernel_size, padding) along the mel axis of every frame, output laid out as the reference's
// transpose(1, 2).reshape does: feature = mel * num_kernels + kernel.  One thread per output value; a frame's n_mels
// inputs are read through L1 by the num_kernels * kernel_size threads that need them, the outputs (num_kernels x the
// input bytes: the pass is bound by its stores) leave coalesced.  Arithmetic: bias, then the taps in ascending order,
// one fmaf each.
__global__ void __launch_bounds__(256) conv1d_mel_kernel(const float* __restrict__ x, long n, int n_mels,
                                                         const float* __restrict__ w, const float* __restrict__ bias, int nk,
                                                         int ks, int pad, float* __restrict__ out) {
    extern __shared__ float wsh[];   // nk * ks weights, nk biases
    for (int i = threadIdx.x; i < nk * ks + nk; i += 256) wsh[i] = i < nk * ks ? w[i] : (bias ? bias[i - nk * ks] : 0.0f);
    __syncthreads();
    const long d_out = (long)n_mels * nk;
    const long total = n * d_out;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / d_out;
        const int f = (int)(e - row * d_out);
        const int m = f / nk, j = f - m * nk;
        const float* xr = x + row * n_mels;
        float acc = wsh[nk * ks + j];
        for (int t = 0; t < ks; t++) {
            const int mm = m + t - pad;
            const float v = (mm >= 0 && mm < n_mels) ? xr[mm] : 0.0f;
            acc = __builtin_fmaf(wsh[j * ks + t], v, acc);
        }
        out[e] = acc;
    }
}
}  // namespace

extern "C" int at_conv1d_mel_f32(at_ctx* ctx, const float* x, int64_t n, int n_mels, const float* weight, const float* bias_or_null,
                                 int num_kernels, int kernel_size, int padding, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && n >= 0 && n_mels > 0 && num_kernels > 0 && kernel_size > 0 && padding >= 0, "at_conv1d_mel_f32: bad arguments");
    AT_REQUIRE(2 * padding == kernel_size - 1, "at_conv1d_mel_f32: only 'same' convolutions (2 * padding == kernel_size - 1), as the reference's");
    AT_REQUIRE((num_kernels * kernel_size + num_kernels) * sizeof(float) <= 32 * 1024, "at_conv1d_mel_f32: too many weights");
    if (n == 0) return AT_OK;
    AT_REQUIRE(x && weight && out, "at_conv1d_mel_f32: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    const long total = n * (long)n_mels * num_kernels;
    long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks > 8L * ctx->n_cus) blocks = 8L * ctx->n_cus;
    if (blocks < 1) blocks = 1;
    AT_LAUNCH(conv1d_mel_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(num_kernels * kernel_size + num_kernels) * sizeof(float), stream, x,
              (long)n, n_mels, weight, bias_or_null, num_kernels, kernel_size, padding, out);
    return AT_OK;
}

extern "C" int at_minmax_scale_clips_f32(at_ctx* ctx, float* x, int64_t n_clips, int64_t clip_elems, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && n_clips >= 0 && n_clips <= 0x7fffffffL && clip_elems > 0, "at_minmax_scale_clips_f32: bad arguments");
    if (n_clips == 0) return AT_OK;
    AT_REQUIRE(x, "at_minmax_scale_clips_f32: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    AT_LAUNCH(minmax_scale_kernel, dim3((unsigned)n_clips), dim3(1024), 0, stream, x, (long)clip_elems);
    return AT_OK;
}

int* at_row_flag(at_ctx* ctx, hipStream_t stream) {
    const bool fresh = ctx->ws[WS_ROW_FLAG] == nullptr;
    int* f = static_cast<int*>(at_ws(ctx, WS_ROW_FLAG, 16, stream));
    if (f && fresh) {
        const hipError_t e = AT_HIP_TOLERATE(hipMemsetAsync(f, 0, 16, stream));
        if (e != hipSuccess) {
            at_fail(AT_E_HIP, "at_row_flag: hipMemsetAsync failed: %s", hipGetErrorString(e));
            return nullptr;
        }
    }
    return f;
}

namespace {
__global__ void row_flag_take_kernel(int* __restrict__ flag, int32_t* __restrict__ out) {
    *out = *flag != 0;
    *flag = 0;
}
}  // namespace

extern "C" int at_logmel_nonfinite_take(at_ctx* ctx, int32_t* flag_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && flag_out, "at_logmel_nonfinite_take: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    int* f = at_row_flag(ctx, stream);
    if (!f) return AT_E_NOMEM;
    AT_LAUNCH(row_flag_take_kernel, dim3(1), dim3(1), 0, stream, f, flag_out);
    return AT_OK;
}

extern "C" int at_l2norm_rows_f32(at_ctx* ctx, const float* x, int64_t n, int d, float* y,
                                  void* stream_) {
    return at_l2norm_rows_flagged(ctx, x, n, d, y, nullptr, static_cast<hipStream_t>(stream_));
}

int at_l2norm_rows_flagged(at_ctx* ctx, const float* x, int64_t n, int d, float* y, int* bad, hipStream_t stream) {
    AT_REQUIRE(ctx, "at_l2norm_rows_f32: ctx is null");
    AT_REQUIRE(n >= 0 && d > 0, "at_l2norm_rows_f32: bad sizes");
    if (n == 0) return AT_OK;
    AT_REQUIRE(x && y, "at_l2norm_rows_f32: null pointer");
    AT_REQUIRE(d <= 16384, "at_l2norm_rows_f32: d=%d not supported (max 16384)", d);
    AT_HIP(hipSetDevice(ctx->device));
    // rows per workgroup: as many as fit ~66 KB of LDS (2 workgroups per CU), at most 256
    int R = 256;
    while (R > 1 && (size_t)R * (d + 2) * sizeof(float) > 68 * 1024) R >>= 1;
    const size_t lds = (size_t)R * (d + 2) * sizeof(float);
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&l2norm_rows_kernel), lds); if (rcl_) return rcl_; }
    const long blocks = (n + R - 1) / R;
    AT_LAUNCH(l2norm_rows_kernel, dim3((unsigned)blocks), dim3(WG), lds, stream, x, (long)n, d,
                       R, y, bad);
    return AT_OK;
}
