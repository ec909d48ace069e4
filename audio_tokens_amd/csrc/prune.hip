// prune.hip -- the bookkeeping that lets a Lloyd iteration skip most of the -2XC^T work without
// changing a single output bit (used by at_assign_pruned_f32, assign.hip).
//
// Replaces nothing in the reference by itself: it accelerates the search inside
// faiss.Kmeans.train (processors/cluster_creator.py:54-56 of danavery/audio-tokens) from the second
// Lloyd iteration on.
//
// Idea (Elkan's lemma, made safe for fp32): row x was assigned to centroid p in the previous
// iteration; its distance to the UPDATED c_p is evaluated exactly (same fmaf chains as the MFMA
// sweep) and is an upper bound bd of its final minimum.  For any other centroid c,
// |x - c| >= |c - c_p| - |x - c_p|, so c cannot reach bd when |c - c_p| > 2 |x - c_p|.  To be
// certain about the COMPUTED distances, which differ from the true ones by at most
// delta = (2d + 8) u (|x|^2 + max|c|^2), the radius is inflated: R = sqrt(bd + delta), skip iff
// (a lower bound of) |c - c_p| > 2 R.  Centroids are handled in groups of 32 (one MFMA
// accumulator) laid out by a kd-style spatial grouping, rows in tiles of 32 visited in
// (previous centroid, distance) order so that the outliers of a cluster sit together.
//   group_min_dist_kernel : dmin[p][g] = lower bound of min_{c in group g} |c - c_p|
//   visit order           : stable radix sort of (p << 8 | quantised distance, row)
//   prune_mask_kernel     : per row bd; per 32-row tile one bit per group ("some row needs it")
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "at_internal.h"

// Onesweep radix sort at every size: below a million items rocPRIM would switch to a merge sort of ~18
// small launches, which is what an iteration of a sharded (N-GPU) run would then mostly consist of;
// the keys here are 13-21 bits wide, two or three onesweep passes.
using at_radix_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int WG = 256;
constexpr float U32 = 5.9604645e-8f;  // 2^-24

// ---- dmin[p][g] ------------------------------------------------------------------------------
// grid (ng, ceil(k/256)); the group's <= 32 rows sit in LDS (read as broadcasts); one thread per p
// with its own row in registers.
template <int D>
__global__ void __launch_bounds__(WG) group_min_dist_kernel(const float* __restrict__ C, int k,
                                                            const int32_t* __restrict__ cperm, int ng,
                                                            float* __restrict__ dmin) {
    __shared__ __attribute__((aligned(16))) float grp[32 * D];
    __shared__ int member[32];
    const int g = blockIdx.x;
    if (threadIdx.x < 32) member[threadIdx.x] = cperm[g * 32 + threadIdx.x];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * (D / 4); e += WG) {
        const int m = member[e / (D / 4)];
        f32x4_t v = {0, 0, 0, 0};
        if (m >= 0) v = reinterpret_cast<const f32x4_t*>(C + (size_t)m * D)[e % (D / 4)];
        reinterpret_cast<f32x4_t*>(grp)[e] = v;
    }
    __syncthreads();
    int p = blockIdx.y * WG + threadIdx.x;
    const bool livep = p < k;
    if (!livep) p = k - 1;
    f32x4_t cp[D / 4];
#pragma unroll
    for (int q = 0; q < D / 4; q++) cp[q] = reinterpret_cast<const f32x4_t*>(C + (size_t)p * D)[q];
    float best = __builtin_inff();
    for (int m = 0; m < 32; m++) {
        if (member[m] < 0) continue;  // uniform
        const f32x4_t* cg = reinterpret_cast<const f32x4_t*>(grp + m * D);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < D / 4; q++) {
            const f32x4_t b = cg[q];
            const float d0 = cp[q][0] - b[0], d1 = cp[q][1] - b[1], d2 = cp[q][2] - b[2], d3 = cp[q][3] - b[3];
            s0 = __builtin_fmaf(d0, d0, s0);
            s1 = __builtin_fmaf(d1, d1, s1);
            s0 = __builtin_fmaf(d2, d2, s0);
            s1 = __builtin_fmaf(d3, d3, s1);
        }
        best = fminf(best, s0 + s1);
    }
    // the fp32 sum of squared fp32 differences is within (d+3)u of the true value: shave 2e-5
    if (livep) dmin[(size_t)p * ng + g] = best == __builtin_inff() ? best : sqrtf(best) * (1.0f - 2e-5f);
}

// ---- visiting order --------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) visit_keys_kernel(const long* __restrict__ ids,
                                                        const float* __restrict__ dis, long n, int k,
                                                        uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, int dbits) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const long c = ids[i];
    const bool ok = c >= 0 && c < k;
    const float top = (float)((1u << dbits) - 1u);
    float r = dis ? sqrtf(fmaxf(dis[i], 0.0f)) * (0.5f * top) : 0.0f;   // unit rows: r <= 2
    r = fminf(r, top);
    keys[i] = ((ok ? (uint32_t)c : (uint32_t)k) << dbits) | (uint32_t)r;
    vals[i] = (uint32_t)i;
}

// the sorted keys give the guesses in visiting order; the sorted values (row indices) leave the sort's buffer in the same pass
__global__ void __launch_bounds__(WG) key_to_hint_kernel(const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, long n,
                                                         uint32_t* __restrict__ hint_sorted,
                                                         uint32_t* __restrict__ order_out, int dbits) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i < n) {
        hint_sorted[i] = keys[i] >> dbits;
        order_out[i] = vals[i];
    }
}

__global__ void __launch_bounds__(WG) max_sqnorm_kernel(const float* __restrict__ C, int k, int d,
                                                        unsigned* __restrict__ out_bits) {
    const int c = blockIdx.x * WG + threadIdx.x;
    float s = 0.0f;
    if (c < k)
        for (int f = 0; f < d; f++) s = __builtin_fmaf(C[(size_t)c * d + f], C[(size_t)c * d + f], s);
    // non-negative floats order like their bit patterns
    atomicMax(out_bits, __float_as_uint(s));
}

// ---- per-row bound and per-tile group mask ----------------------------------------------------
// One wave per 64 visiting positions (two 32-row tiles: lanes 0-31 and 32-63).
template <int D>
__global__ void __launch_bounds__(WG) prune_mask_kernel(const float* __restrict__ X, long n,
                                                        const float* __restrict__ C, int k,
                                                        const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ hint_sorted,
                                                        const float* __restrict__ dmin, int ng,
                                                        const unsigned* __restrict__ cnmax_bits,
                                                        float* __restrict__ bd_out,
                                                        uint32_t* __restrict__ mask, int ngw,
                                                        unsigned long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long wpos = ((long)blockIdx.x * (WG / 64) + (threadIdx.x >> 6)) * 64;
    if (wpos >= n) return;
    long pos = wpos + lane;
    const bool live = pos < n;
    if (!live) pos = n - 1;
    const long r = order[pos];
    const uint32_t p = hint_sorted[pos];
    const bool has = p < (uint32_t)k;
    const f32x4_t* px = reinterpret_cast<const f32x4_t*>(X + r * D);
    const f32x4_t* pc = reinterpret_cast<const f32x4_t*>(C + (size_t)(has ? p : 0) * D);
    // every load of the two rows is issued before the first dependent fma: a row-per-lane read that
    // waits for each 16-byte piece in turn keeps its 128-byte lines alive for many microseconds, and
    // with 2048 waves doing so they fall out of L2 and are fetched again (2.4x the bytes, measured)
    f32x4_t xv[D / 4], cv[D / 4];
#pragma unroll
    for (int q = 0; q < D / 4; q++) xv[q] = px[q];
#pragma unroll
    for (int q = 0; q < D / 4; q++) cv[q] = pc[q];
    float xn = 0.0f, cn = 0.0f, ip = 0.0f;
#pragma unroll
    for (int q = 0; q < D / 4; q++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            xn = __builtin_fmaf(xv[q][e], xv[q][e], xn);
            cn = __builtin_fmaf(cv[q][e], cv[q][e], cn);
            ip = __builtin_fmaf(cv[q][e], xv[q][e], ip);
        }
    }
    // exactly the value the MFMA sweep would produce for (x, c_p)
    const float dh = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);
    if (live) bd_out[pos] = has ? dh : __builtin_inff();

    const float cnmax = __uint_as_float(*cnmax_bits);
    const float delta = (2.0f * D + 8.0f) * U32 * (xn + cnmax) * 1.01f;
    // rows without a guess need every group; positions past n need none
    const float tau = !live ? -1.0f
                            : (has ? 2.0f * sqrtf(dh + delta) * (1.0f + 4.0f * U32) + 1e-30f : __builtin_inff());

    // Rows arrive sorted by guess, so a 32-row tile holds a few runs of equal p: a group is needed by
    // the tile iff dmin[p][g] <= the largest tau of some run.  Lanes stand for groups here (64 per
    // pass, one coalesced read of the run's dmin row), not for rows.
    const long tile = wpos / 32;
    const bool second = wpos + 32 < n;
    int needed = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half == 1 && !second) break;
        const unsigned long long hm = half ? 0xffffffff00000000ull : 0x00000000ffffffffull;
        const bool mine = live && ((lane >> 5) == half);
        const bool all = (__builtin_amdgcn_ballot_w64(mine && !has) != 0);
        unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (all) {
#pragma unroll
            for (int it = 0; it < 8; it++) acc[it] = ~0ull;
        } else {
            unsigned long long todo = __builtin_amdgcn_ballot_w64(mine) & hm;
            while (todo != 0) {
                const int leader = __builtin_ctzll(todo);
                const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)p, leader);
                const bool in_run = mine && p == pl;
                todo &= ~__builtin_amdgcn_ballot_w64(in_run);
                float t = in_run ? tau : -1.0f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t = __builtin_fmaxf(t, __shfl_xor(t, off));
                const float* drow = dmin + (size_t)pl * ng;
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int g = 64 * it + lane;
                    if (64 * it < ng) acc[it] |= __builtin_amdgcn_ballot_w64(g < ng && drow[g] <= t);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            if (64 * it >= ng) break;
            unsigned long long v = acc[it];
            if (ng - 64 * it < 64) v &= (1ull << (ng - 64 * it)) - 1ull;
            if (lane == 0) {
                mask[(size_t)(tile + half) * ngw + 2 * it] = (uint32_t)v;
                if (2 * it + 1 < ngw) mask[(size_t)(tile + half) * ngw + 2 * it + 1] = (uint32_t)(v >> 32);
            }
            needed += __builtin_popcountll(v);
        }
    }
    if (lane == 0) {  // statistics only (256 slots to keep the atomics off one address)
        unsigned long long* slot = stats + 2 * (blockIdx.x & 255);
        atomicAdd(&slot[0], (unsigned long long)needed);
        atomicAdd(&slot[1], (unsigned long long)ng * (second ? 2 : 1));
    }
}

// Coarse step of the unguided search: every row names one group (hint_sorted holds GROUP ids here);
// the tile needs exactly the groups its rows name, and nobody has a running best yet.
__global__ void __launch_bounds__(WG) group_only_mask_kernel(long n, const uint32_t* __restrict__ group_sorted,
                                                             int ng, const uint32_t* __restrict__ gnbr,
                                                             float* __restrict__ bd_out,
                                                             uint32_t* __restrict__ mask, int ngw) {
    const int lane = threadIdx.x & 63;
    const long wpos = ((long)blockIdx.x * (WG / 64) + (threadIdx.x >> 6)) * 64;
    if (wpos >= n) return;
    const long pos = wpos + lane;
    const bool live = pos < n;
    const uint32_t g = live ? group_sorted[pos] : 0xffffffffu;
    if (live) bd_out[pos] = __builtin_inff();
    const long tile = wpos / 32;
    const bool second = wpos + 32 < n;
    for (int w = 0; w < ngw; w++) {
        // OR of (1 << bit) over the lanes of each half-wave whose group falls into word w
        // gnbr (optional, [ng][ngw]): the groups worth searching for a row that names group g
        uint32_t mine = 0u;
        if (live && g < (uint32_t)ng)
            mine = gnbr ? gnbr[(size_t)g * ngw + w] : ((int)(g >> 5) == w ? (1u << (g & 31)) : 0u);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) mine |= (uint32_t)__shfl_xor((int)mine, off);
        if (lane == 0) mask[(size_t)tile * ngw + w] = mine;
        if (lane == 32 && second) mask[(size_t)(tile + 1) * ngw + w] = mine;
    }
}

// means[g] = mean of the group's member rows (coarse quantiser of the unguided search)
__global__ void __launch_bounds__(WG) group_means_kernel(const float* __restrict__ C, int d,
                                                         const int32_t* __restrict__ cperm,
                                                         float* __restrict__ means) {
    const int g = blockIdx.x;
    for (int f = threadIdx.x; f < d; f += WG) {
        float s = 0.0f;
        int cnt = 0;
        for (int m = 0; m < 32; m++) {
            const int row = cperm[g * 32 + m];
            if (row >= 0) { s += C[(size_t)row * d + f]; cnt++; }
        }
        means[(size_t)g * d + f] = cnt ? s / (float)cnt : 0.0f;
    }
}

// Neighbour table of the groups (heuristic input of the guess generators): one workgroup per group g
// scores every group mean against mean g (fp32 sum of squared differences), then takes the nnb nearest
// (g itself first; ties to the lower index) and sets their bits in gnbr[g][0..ngw).
__global__ __launch_bounds__(256) void group_neighbours_kernel(const float* __restrict__ means, int ng, int d, int nnb,
                                                               uint32_t* __restrict__ gnbr) {
    __shared__ float own[128];
    __shared__ unsigned long long wave_best[4];
    __shared__ uint32_t bits[16];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int ngw = (ng + 31) / 32;
    for (int f = tid; f < d; f += 256) own[f] = means[(size_t)g * d + f];
    if (tid < 16) bits[tid] = 0u;
    __syncthreads();
    // up to two candidates per thread (ng <= 512); key = distance bits (>= 0, so they order as integers) : index
    unsigned long long key[2];
    for (int s = 0; s < 2; s++) {
        const int j = tid + 256 * s;
        key[s] = ~0ull;
        if (j < ng && j != g) {
            float acc = 0.0f;
            for (int f = 0; f < d; f++) {
                const float t = means[(size_t)j * d + f] - own[f];
                acc = __builtin_fmaf(t, t, acc);
            }
            key[s] = ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned)j;
        }
    }
    if (tid == 0) bits[g >> 5] |= 1u << (g & 31);
    for (int round = 1; round < nnb && round < ng; round++) {
        unsigned long long best = key[0] < key[1] ? key[0] : key[1];
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other < best ? other : best;
        }
        if ((tid & 63) == 0) wave_best[tid >> 6] = best;
        __syncthreads();
        best = wave_best[0];
        for (int w = 1; w < 4; w++) best = wave_best[w] < best ? wave_best[w] : best;
        __syncthreads();
        if (best == ~0ull) break;
        const unsigned j = (unsigned)best;
        if (tid == 0) bits[j >> 5] |= 1u << (j & 31);
        if (key[0] == best) key[0] = ~0ull;
        if (key[1] == best) key[1] = ~0ull;
    }
    __syncthreads();
    if (tid < ngw) gnbr[(size_t)g * ngw + tid] = bits[tid];
}

}  // namespace

// ---- entry points used by assign.hip and the C ABI -------------------------------------------
int at_prune_prepass(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                     const uint32_t* order, const uint32_t* hint_sorted, const float* dmin, int ng,
                     float* bd_out, uint32_t* mask, int ngw, int mode, hipStream_t stream) {
    if (mode == 1) {  // hint_sorted holds group ids: each tile needs just those groups (+ neighbours)
        const long waves1 = (n + 63) / 64;
        AT_LAUNCH(group_only_mask_kernel, dim3((unsigned)((waves1 + WG / 64 - 1) / (WG / 64))), dim3(WG), 0,
                           stream, (long)n, hint_sorted, ng, reinterpret_cast<const uint32_t*>(dmin), bd_out, mask,
                           ngw);
        return AT_OK;
    }
    unsigned* cnmax = static_cast<unsigned*>(at_ws(ctx, WS_REDUCE, 1024 * sizeof(double), stream));
    if (!cnmax) return AT_E_NOMEM;
    AT_HIP(hipMemsetAsync(cnmax, 0, sizeof(unsigned), stream));
    AT_LAUNCH(max_sqnorm_kernel, dim3((k + WG - 1) / WG), dim3(WG), 0, stream, c, k, d, cnmax);
    const long waves = (n + 63) / 64;
    const dim3 grid((unsigned)((waves + WG / 64 - 1) / (WG / 64)));
    const bool fresh = ctx->ws[WS_PRUNE_STATS] == nullptr;
    unsigned long long* stats = static_cast<unsigned long long*>(at_ws(ctx, WS_PRUNE_STATS, 4096, stream));
    if (!stats) return AT_E_NOMEM;
    if (fresh) AT_HIP(hipMemsetAsync(stats, 0, 4096, stream));
    if (d == 64)
        AT_LAUNCH(prune_mask_kernel<64>, grid, dim3(WG), 0, stream, x, (long)n, c, k, order, hint_sorted,
                           dmin, ng, cnmax, bd_out, mask, ngw, stats);
    else
        AT_LAUNCH(prune_mask_kernel<128>, grid, dim3(WG), 0, stream, x, (long)n, c, k, order, hint_sorted,
                           dmin, ng, cnmax, bd_out, mask, ngw, stats);
    return AT_OK;
}

extern "C" {

int at_group_min_dist_f32(at_ctx* ctx, const float* c, int k, int d, const int32_t* cperm, int ng,
                          float* dmin, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && c && cperm && dmin && k > 0 && d > 0 && ng > 0, "at_group_min_dist_f32: bad arguments");
    AT_REQUIRE((d == 64 || d == 128) && at_aligned16(c), "at_group_min_dist_f32: d must be 64 or 128");
    AT_HIP(hipSetDevice(ctx->device));
    if (ctx->dbg.dmin_kernel != 0) return   // (switch: 0 = the fp32 vector-ALU kernel below)
        at_group_min_dist_f16(ctx, c, k, d, cperm, ng, dmin, stream);
    ctx->img16_c = nullptr;  // (no fp16 image is left behind by this form)
    const dim3 grid(ng, (k + WG - 1) / WG);
    if (d == 64)
        AT_LAUNCH(group_min_dist_kernel<64>, grid, dim3(WG), 0, stream, c, k, cperm, ng, dmin);
    else
        AT_LAUNCH(group_min_dist_kernel<128>, grid, dim3(WG), 0, stream, c, k, cperm, ng, dmin);
    return AT_OK;
}

int at_prune_stats(at_ctx* ctx, int64_t* needed, int64_t* total, int reset) {
    AT_REQUIRE(ctx && needed && total, "at_prune_stats: bad arguments");
    *needed = *total = 0;
    if (!ctx->ws[WS_PRUNE_STATS]) return AT_OK;
    AT_HIP(hipSetDevice(ctx->device));
    unsigned long long host[512];
    AT_HIP(hipDeviceSynchronize());
    AT_HIP(hipMemcpy(host, ctx->ws[WS_PRUNE_STATS], sizeof host, hipMemcpyDeviceToHost));
    for (int i = 0; i < 256; i++) {
        *needed += (int64_t)host[2 * i];
        *total += (int64_t)host[2 * i + 1];
    }
    if (reset) AT_HIP(hipMemset(ctx->ws[WS_PRUNE_STATS], 0, 4096));
    return AT_OK;
}

int at_group_means_f32(at_ctx* ctx, const float* c, int k, int d, const int32_t* cperm, int ng, float* means,
                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && c && cperm && means && k > 0 && d > 0 && ng > 0, "at_group_means_f32: bad arguments");
    AT_HIP(hipSetDevice(ctx->device));
    AT_LAUNCH(group_means_kernel, dim3(ng), dim3(WG), 0, stream, c, d, cperm, means);
    return AT_OK;
}

int at_group_neighbours_f32(at_ctx* ctx, const float* means, int ng, int d, int nnb, uint32_t* gnbr, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && means && gnbr, "at_group_neighbours_f32: null pointer");
    AT_REQUIRE(ng > 0 && ng <= 512 && d > 0 && d <= 128 && nnb > 0, "at_group_neighbours_f32: needs ng <= 512, d <= 128, nnb > 0");
    AT_HIP(hipSetDevice(ctx->device));
    AT_LAUNCH(group_neighbours_kernel, dim3(ng), dim3(256), 0, stream, means, ng, d, nnb, gnbr);
    return AT_OK;
}

int at_visit_order_f32(at_ctx* ctx, const int64_t* ids, const float* dis, int64_t n, int k,
                       uint32_t* order_out, uint32_t* hint_sorted_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && ids && order_out && hint_sorted_out, "at_visit_order_f32: null pointer");
    AT_REQUIRE(n >= 0 && n < (int64_t)UINT32_MAX && k > 0 && k < (1 << 23), "at_visit_order_f32: bad sizes");
    if (n == 0) return AT_OK;
    AT_HIP(hipSetDevice(ctx->device));
    uint32_t* keys_a = static_cast<uint32_t*>(at_ws(ctx, WS_VISIT_KEYS_A, (size_t)n * 4, stream));
    uint32_t* keys_b = static_cast<uint32_t*>(at_ws(ctx, WS_VISIT_KEYS_B, (size_t)n * 4, stream));
    uint32_t* vals_a = static_cast<uint32_t*>(at_ws(ctx, WS_VISIT_VALS_A, (size_t)n * 4, stream));
    uint32_t* vals_b = static_cast<uint32_t*>(at_ws(ctx, WS_VISIT_VALS_B, (size_t)n * 4, stream));
    if (!keys_a || !keys_b || !vals_a || !vals_b) return AT_E_NOMEM;
    // key = cluster | distance in `dbits` bits (switch visit_bits, default 8: measured 2.07 / 1.86 / 1.52 / 1.46 ms per
    // Lloyd iteration at 0 / 2 / 5 / 8 bits on bench-like rows, no further gain at 10 or 12)
    unsigned cbits = 1;
    while ((1u << cbits) <= (unsigned)k) cbits++;
    int dbits = ctx->dbg.visit_bits;
    if (dbits < 0) dbits = 0;
    if (dbits > 12) dbits = 12;
    if ((int)cbits + dbits > 32) dbits = 32 - (int)cbits;
    AT_LAUNCH(visit_keys_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, stream,
                       reinterpret_cast<const long*>(ids), dis, (long)n, k, keys_a, vals_a, dbits);
    const unsigned bits = cbits + (unsigned)dbits;
    rocprim::double_buffer<uint32_t> kb(keys_a, keys_b);
    rocprim::double_buffer<uint32_t> vb(vals_a, vals_b);
    size_t tmp_bytes = 0;
    AT_HIP(rocprim::radix_sort_pairs<at_radix_config>(nullptr, tmp_bytes, kb, vb, (size_t)n, 0, bits, stream));
    void* tmp = at_ws(ctx, WS_VISIT_TMP, tmp_bytes, stream);
    if (!tmp) return AT_E_NOMEM;
    AT_HIP(rocprim::radix_sort_pairs<at_radix_config>(tmp, tmp_bytes, kb, vb, (size_t)n, 0, bits, stream));
    AT_LAUNCH(key_to_hint_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, stream, kb.current(),
                       vb.current(), (long)n, hint_sorted_out, order_out, dbits);
    return AT_OK;
}

}  // extern "C"
