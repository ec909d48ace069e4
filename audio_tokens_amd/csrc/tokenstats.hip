// tokenstats.hip -- rank-frequency statistics of the token histogram, without leaving the device.
//
// Reference call site: processors/spec_tokenizer.py:129-240 (analyze_tokens, plot_token_distribution,
// analyze_zipf_and_tail): a Python Counter over tokens.tolist() (O(frames) Python ints), a host sort of the
// counts, np.cumsum / np.searchsorted for the "80 % of occurrences" rank and scipy.stats.linregress over
// (ln rank, ln count) of the middle 80 % of the ranks.  Here: at_token_histogram_i64 (kmeans.hip) counts on the
// device; this file sorts the k counts (rocPRIM radix sort, descending, ties in ascending token id) and one
// workgroup computes the cumulative-share rank and the regression with linregress' formulas in double.
// Only k numbers ever reach the host, for the optional plots.
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "at_internal.h"

namespace {

constexpr int WG = 256;

__global__ void iota_kernel(int32_t* v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// sc: counts sorted descending.  stats: [0] total, [1] unique U, [2] #ranks with cumulative share < 0.8,
// [3] slope, [4] intercept, [5] r, [6] points fitted, [7] #ranks (< U) with cumulative share < 0.8 among the U
__global__ __launch_bounds__(WG) void token_stats_kernel(const unsigned long long* __restrict__ sc, int k,
                                                        double* __restrict__ stats) {
    __shared__ double sh[WG / 64];
    __shared__ unsigned long long seg[WG];
    const int t = threadIdx.x;
    const int per = (k + WG - 1) / WG, lo = min(k, t * per), hi = min(k, lo + per);
    unsigned long long mine = 0;
    double uniq = 0.0;
    for (int r = lo; r < hi; r++) {
        mine += sc[r];
        uniq += sc[r] > 0 ? 1.0 : 0.0;
    }
    seg[t] = mine;
    const double U = block_sum(uniq, sh);   // (barriers inside: seg is visible afterwards)
    unsigned long long before = 0, total = 0;
    for (int j = 0; j < WG; j++) {
        if (j < t) before += seg[j];
        total += seg[j];
    }
    // np.searchsorted(np.cumsum(freq) / total, 0.8): ranks whose cumulative share is still below 0.8
    double below = 0.0;
    unsigned long long cum = before;
    for (int r = lo; r < hi; r++) {
        cum += sc[r];
        if (sc[r] > 0 && (double)cum / (double)total < 0.8) below += 1.0;
    }
    below = block_sum(below, sh);
    // scipy.stats.linregress over ranks [int(0.1 U), int(0.9 U)): x = ln(rank), y = ln(count)
    const int nu = (int)U;
    const int s = (int)(0.1 * nu), e = (int)(0.9 * nu);
    const int m = e - s;
    double sx = 0.0, sy = 0.0;
    for (int r = s + t; r < e; r += WG) {
        sx += log((double)(r + 1));
        sy += log((double)sc[r]);
    }
    sx = block_sum(sx, sh);
    sy = block_sum(sy, sh);
    const double xm = m > 0 ? sx / m : 0.0, ym = m > 0 ? sy / m : 0.0;
    double sxx = 0.0, sxy = 0.0, syy = 0.0;
    for (int r = s + t; r < e; r += WG) {
        const double dx = log((double)(r + 1)) - xm, dy = log((double)sc[r]) - ym;
        sxx += dx * dx;
        sxy += dx * dy;
        syy += dy * dy;
    }
    sxx = block_sum(sxx, sh);
    sxy = block_sum(sxy, sh);
    syy = block_sum(syy, sh);
    if (t == 0) {
        stats[0] = (double)total;
        stats[1] = U;
        stats[2] = below;
        double slope = 0.0, icpt = 0.0, rv = 0.0;
        if (m >= 2 && sxx > 0.0) {
            slope = sxy / sxx;
            icpt = ym - slope * xm;
            rv = (sxx > 0.0 && syy > 0.0) ? sxy / sqrt(sxx * syy) : 0.0;
            rv = fmin(1.0, fmax(-1.0, rv));
        }
        stats[3] = slope;
        stats[4] = icpt;
        stats[5] = rv;
        stats[6] = (double)m;
        stats[7] = 0.0;
    }
}

}  // namespace

extern "C" int at_token_stats_f64(at_ctx* ctx, const int64_t* counts, int k, int64_t* sorted_counts,
                                  int32_t* sorted_tokens, double* stats, void* stream_) {
    AT_REQUIRE(ctx && counts && sorted_counts && sorted_tokens && stats && k > 0, "at_token_stats_f64: bad arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AT_HIP(hipSetDevice(ctx->device));
    int32_t* iota = static_cast<int32_t*>(at_ws(ctx, WS_TSTAT_IOTA, (size_t)k * 4, stream));
    if (!iota) return AT_E_NOMEM;
    AT_LAUNCH(iota_kernel, dim3((k + WG - 1) / WG), dim3(WG), 0, stream, iota, k);
    const unsigned long long* kin = reinterpret_cast<const unsigned long long*>(counts);
    unsigned long long* kout = reinterpret_cast<unsigned long long*>(sorted_counts);
    size_t tmp_bytes = 0;
    AT_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, kin, kout, iota, sorted_tokens, (size_t)k, 0, 64, stream));
    void* tmp = at_ws(ctx, WS_TSTAT_TMP, tmp_bytes, stream);
    if (!tmp) return AT_E_NOMEM;
    AT_HIP(rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, kin, kout, iota, sorted_tokens, (size_t)k, 0, 64, stream));
    AT_LAUNCH(token_stats_kernel, dim3(1), dim3(WG), 0, stream, kout, k, stats);
    return AT_OK;
}
