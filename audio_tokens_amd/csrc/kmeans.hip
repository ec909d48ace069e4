// kmeans.hip -- the non-GEMM half of one Lloyd iteration on gfx950.
//
// Replaces, inside faiss.Kmeans.train (processors/cluster_creator.py:42-56 of danavery/audio-tokens):
//   subsample_training_set's row copy      -> at_gather_rows_f32
//   compute_centroids (accumulate)         -> at_centroid_accum_f32
//   compute_centroids (1/count scaling)    -> at_centroid_finalize_f32   (+ data-parallel combine)
//   the objective  sum_i dis[i]            -> at_sum_f32
//   the isfinite() scan of the input       -> at_any_nonfinite_f32
//
// All of these are HBM-bound passes over [n][d] fp32 rows (4d+8 B per point per iteration).
//
// compute_centroids must reproduce what FAISS's owning thread produces: the members of a cluster
// are added in ASCENDING point index with fp32 adds.  That order is made explicit here: a stable
// radix sort of (assignment, point index) pairs (rocPRIM) yields every cluster's member list in
// ascending index; one wavefront then walks one list, lanes across the feature axis, so each
// row is one coalesced 4d-byte read and the adds are sequential per (cluster, feature) exactly as
// on the CPU.  No float atomics anywhere: results are bitwise reproducible.
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "at_internal.h"

// Onesweep radix sort at every size: below a million items rocPRIM would switch to a merge sort of ~18
// small launches, which is what an iteration of a sharded (N-GPU) run would then mostly consist of;
// the keys here are 13-21 bits wide, two or three onesweep passes.
using at_radix_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

namespace {

constexpr int WG = 256;

__global__ void __launch_bounds__(WG) gather_rows_kernel(const float* __restrict__ x, int d4,
                                                         const int32_t* __restrict__ idx, long m,
                                                         float* __restrict__ out) {
    // one 16-byte chunk per thread; consecutive threads walk one row, then the next
    const long e = (long)blockIdx.x * WG + threadIdx.x;
    if (e >= m * d4) return;
    const long r = e / d4;
    const int c = (int)(e - r * d4);
    const float4* src = reinterpret_cast<const float4*>(x) + (long)idx[r] * d4 + c;
    reinterpret_cast<float4*>(out)[e] = *src;
}

__global__ void __launch_bounds__(WG) gather_rows_scalar_kernel(const float* __restrict__ x, int d,
                                                                const int32_t* __restrict__ idx,
                                                                long m, float* __restrict__ out) {
    const long e = (long)blockIdx.x * WG + threadIdx.x;
    if (e >= m * d) return;
    const long r = e / d;
    const int c = (int)(e - r * d);
    out[e] = x[(long)idx[r] * d + c];
}

__global__ void __launch_bounds__(WG) make_keys_kernel(const long* __restrict__ ids, long n, int k,
                                                       uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    long c = ids[i];
    // an id outside [0, k) (e.g. -1 from an all-NaN row) is parked in a trailing bucket that no
    // centroid reads
    keys[i] = (c >= 0 && c < k) ? (uint32_t)c : (uint32_t)k;
    vals[i] = (uint32_t)i;
}

// offsets[c] = first position p in the sorted key array with keys[p] >= c, for c in [0, k].
__global__ void __launch_bounds__(WG) segment_offsets_kernel(const uint32_t* __restrict__ keys, long n,
                                                             int k, uint32_t* __restrict__ offsets) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c > k) return;
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
    }
    offsets[c] = (uint32_t)lo;
}

// One wavefront per (cluster, 64*VEC-feature slab).  Lane owns VEC consecutive features.
template <int VEC>
__global__ void __launch_bounds__(WG) centroid_accum_kernel(const float* __restrict__ x, int d,
                                                            const uint32_t* __restrict__ order,
                                                            const uint32_t* __restrict__ offsets,
                                                            int k, int slabs, uint32_t long_list,
                                                            float* __restrict__ sums,
                                                            float* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (w >= (long)k * slabs) return;
    const int c = (int)(w / slabs);
    const int slab = (int)(w - (long)c * slabs);
    const int f0 = (slab * 64 + lane) * VEC;
    const bool live = f0 < d;  // d is a multiple of VEC
    const uint32_t beg = offsets[c], end = offsets[c + 1];
    if (end - beg > long_list) return;  // left to centroid_accum_long_kernel

    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.0f;

    // Rows are fetched RB at a time into one of two register sets: the loads of batch i+1 are in flight while
    // batch i is added (in member order: one dependent chain per feature, as the contract says), and the member
    // indices of the next 64 rows are fetched while this block of 64 is summed.  A list is one wave's sequential
    // walk, so the kernel lasts as long as its longest lists (up to 2048 members): what bounds those is how many
    // row reads the wave keeps in flight -- 16 per set (32 in flight) instead of round 2's 8: 168 -> see DESIGN us
    // at 2 M x 64.
    constexpr int RB = VEC == 4 ? 8 : 16;
    auto fetch = [&](uint32_t mine, uint32_t m, uint32_t cnt, float (&t)[RB][VEC]) {
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const uint32_t src = __builtin_amdgcn_readlane(mine, (m + u) & 63);
            const bool ok = (m + u < cnt) && live;
            if constexpr (VEC == 4) {
                float4 q = ok ? *reinterpret_cast<const float4*>(x + (size_t)src * d + f0) : make_float4(0, 0, 0, 0);
                t[u][0] = q.x; t[u][1] = q.y; t[u][2] = q.z; t[u][3] = q.w;
            } else if constexpr (VEC == 2) {
                float2 q = ok ? *reinterpret_cast<const float2*>(x + (size_t)src * d + f0) : make_float2(0, 0);
                t[u][0] = q.x; t[u][1] = q.y;
            } else {
                t[u][0] = ok ? x[(size_t)src * d + f0] : 0.0f;
            }
        }
    };
    auto add = [&](uint32_t m, uint32_t cnt, const float (&t)[RB][VEC]) {
#pragma unroll
        for (int u = 0; u < RB; u++) {
            if (m + u < cnt) {  // wave-uniform: the tail adds nothing at all
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v] += t[u][v];
            }
        }
    };
    uint32_t mine_next = (beg < end && lane < min(64u, end - beg)) ? order[beg + lane] : 0u;
    for (uint32_t base = beg; base < end; base += 64) {
        const uint32_t cnt = min(64u, end - base);
        const uint32_t mine = mine_next;
        if (base + 64 < end) mine_next = lane < min(64u, end - base - 64) ? order[base + 64 + lane] : 0u;
        float tA[RB][VEC], tB[RB][VEC];
        fetch(mine, 0, cnt, tA);
        for (uint32_t m = 0; m < cnt; m += 2 * RB) {
            if (m + RB < cnt) fetch(mine, m + RB, cnt, tB);
            add(m, cnt, tA);
            if (m + RB < cnt) {
                if (m + 2 * RB < cnt) fetch(mine, m + 2 * RB, cnt, tA);
                add(m + RB, cnt, tB);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int v = 0; v < VEC; v++) sums[(size_t)c * d + f0 + v] = acc[v];
    }
    if (slab == 0 && lane == 0) counts[c] = (float)(end - beg);
}

// Long member lists (one huge cluster, e.g. every digital-silence frame) would leave a single
// wavefront chasing HBM latency for milliseconds.  The sums of different features are independent,
// so such a cluster is cut ACROSS FEATURES: one workgroup per (long cluster, 4-feature slice).
// Waves 1-3 stream that 16-byte slice of every member row, in member order, into a
// double-buffered LDS ring (all index loads, then all row loads, then the LDS writes, so a whole
// chunk is in flight at once); four lanes of wave 0 do nothing but the dependent chain of fp32
// adds.  Same ascending-member order, hence the same bits, as the one-wave kernel.
constexpr int LONG_CHUNK = 2048;                       // members per ring buffer (32 KiB)
constexpr int LONG_LOADERS = WG - 64;                  // threads that load
constexpr int LONG_PER_THREAD = (LONG_CHUNK + LONG_LOADERS - 1) / LONG_LOADERS;

constexpr int EARLY_MAX = 16;       // long clusters whose member lists come from the ordered compaction
constexpr int EARLY_ROWS = 4096;    // rows per workgroup of the ordered compaction

// WS_LONG_PRED, in ints: [0] number of long clusters the last call saw (the next call's early set, capped at
// EARLY_MAX), [1 .. EARLY_MAX] their ids, then k generation marks ("summed early in call `gen`").  One definition for
// both call sites: round 2 raised EARLY_MAX from 8 to 16 and left the marks at offset 16, on top of the last id.
struct LongPred {
    int* pred_n;
    int* pred;
    unsigned* done;
    static constexpr size_t MARKS_AT = 1 + EARLY_MAX;
    static size_t bytes(int k) { return (MARKS_AT + (size_t)k) * sizeof(int); }
    explicit LongPred(int* pw) : pred_n(pw), pred(pw + 1), done(reinterpret_cast<unsigned*>(pw + MARKS_AT)) {}
};
static_assert(LongPred::MARKS_AT >= 1 + EARLY_MAX, "the generation marks must start behind the last predicted cluster id");

// early != 0: list `slot` of the early lists (cluster cluster_of[slot], members early_offsets[slot] ..); marks the
// cluster done[c] = gen.  early == 0: the regular pass over all clusters after the sort; records every long cluster
// in pred (the next call's early set) and skips those the early pass has already summed.
__global__ void __launch_bounds__(WG) centroid_accum_long_kernel(const float* __restrict__ x, int d,
                                                                 const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ offsets,
                                                                 uint32_t long_list,
                                                                 float* __restrict__ sums,
                                                                 float* __restrict__ counts, int k, int early,
                                                                 const int* __restrict__ cluster_of,
                                                                 const int* __restrict__ n_slots,
                                                                 unsigned* __restrict__ done, unsigned gen,
                                                                 int slot0 = 0, int slot1 = 0x7fffffff) {
    // feature-major ring: ring[buffer][feature][member], so an adder lane reads four consecutive members of
    // its feature with one 16-byte LDS read
    __shared__ __attribute__((aligned(16))) float ring[2][4][LONG_CHUNK];
    // early: slot = blockIdx.x of the early lists.  Regular pass: the workgroups stride over the late list
    // (cluster_of[0] = its length, clusters behind it), offsets indexed by cluster.
    const int n_slot = min(early ? min(*n_slots, EARLY_MAX) : cluster_of[0], slot1);   // (list mode: slots [slot0, slot1))
    const int piece = blockIdx.y;  // features 4*piece .. 4*piece+3
    for (int slot = slot0 + blockIdx.x; slot < n_slot; slot += gridDim.x) {
    const int c = early ? cluster_of[slot] : cluster_of[1 + slot];
    if (c < 0 || c >= k) continue;
    const uint32_t beg = early ? offsets[slot] : offsets[c], end = early ? offsets[slot + 1] : offsets[c + 1];
    const uint32_t len = end - beg;
    if (len <= long_list) continue;  // uniform for the workgroup
    const int tid = threadIdx.x;
    const bool adder = tid < 64;
    const uint32_t nchunks = (len + LONG_CHUNK - 1) / LONG_CHUNK;
    const float4* rows = reinterpret_cast<const float4*>(x) + piece;
    const int d4 = d >> 2;

    auto stage = [&](uint32_t ch) {  // loaders only
        const uint32_t m0 = ch * LONG_CHUNK;
        const uint32_t cnt = min((uint32_t)LONG_CHUNK, len - m0);
        const uint32_t t = tid - 64;
        uint32_t src[LONG_PER_THREAD];
        float4 v[LONG_PER_THREAD];
#pragma unroll
        for (int u = 0; u < LONG_PER_THREAD; u++) {
            const uint32_t e = t + u * LONG_LOADERS;
            src[u] = e < cnt ? order[beg + m0 + e] : 0u;
        }
#pragma unroll
        for (int u = 0; u < LONG_PER_THREAD; u++) {
            const uint32_t e = t + u * LONG_LOADERS;
            v[u] = e < cnt ? rows[(size_t)src[u] * d4] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < LONG_PER_THREAD; u++) {
            const uint32_t e = t + u * LONG_LOADERS;
            if (e < cnt) {
                ring[ch & 1][0][e] = v[u].x;
                ring[ch & 1][1][e] = v[u].y;
                ring[ch & 1][2][e] = v[u].z;
                ring[ch & 1][3][e] = v[u].w;
            }
        }
    };

    float acc = 0.0f;
    if (!adder) stage(0);
    __syncthreads();
    for (uint32_t ch = 0; ch < nchunks; ch++) {
        if (!adder) {
            if (ch + 1 < nchunks) stage(ch + 1);
        } else if (tid < 4) {
            const float* src = ring[ch & 1][tid];
            const uint32_t cnt = min((uint32_t)LONG_CHUNK, len - ch * LONG_CHUNK);
            // one dependent chain of adds; 128 members are read per batch so that the LDS latency is paid
            // once per 128 adds (batches of 16: 380 us on a 46 000-member list, of 128: 310 us)
            uint32_t m = 0;
            {
                for (; m + 128 <= cnt; m += 128) {
                    float4 t[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) t[u] = *reinterpret_cast<const float4*>(src + m + 4 * u);
#pragma unroll
                    for (int u = 0; u < 32; u++) {
                        acc += t[u].x;
                        acc += t[u].y;
                        acc += t[u].z;
                        acc += t[u].w;
                    }
                }
            }
            for (; m + 16 <= cnt; m += 16) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) t[u] = *reinterpret_cast<const float4*>(src + m + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc += t[u].x;
                    acc += t[u].y;
                    acc += t[u].z;
                    acc += t[u].w;
                }
            }
            for (; m < cnt; m++) acc += src[m];
        }
        __syncthreads();
    }
    if (tid < 4) sums[(size_t)c * d + 4 * piece + tid] = acc;
    if (tid == 0 && piece == 0) {
        counts[c] = (float)len;
        if (early && done) done[c] = gen;
    }
    __syncthreads();   // (the ring is reused by the next list)
    }  // slot
}

// ---- member lists of the (predicted) long clusters, ahead of the sort -----------------------------------------
// A list of tens of thousands of members is one dependent chain of adds, 300 us at 2 M rows: as long as everything
// else of the accumulation together, and it used to start only after the sort.  The clusters that were long in
// the previous call (a Lloyd iteration changes 2 % of the assignments) get their lists from an ordered compaction
// of ids instead -- count per 4096-row block, scan, ordered write: three small launches -- so their chains run
// beside the sort.  Whatever the prediction, a list built here is exactly the cluster's members in ascending row
// order; a cluster that is not long after all is left to the short-list kernel.
__global__ void __launch_bounds__(WG) early_count_kernel(const long* __restrict__ ids, long n, const int* __restrict__ pred,
                                                         const int* __restrict__ pred_n, int nblk,
                                                         uint32_t* __restrict__ blockcnt) {
    __shared__ uint32_t cnt[EARLY_MAX];
    const int np = min(*pred_n, EARLY_MAX);
    if (threadIdx.x < EARLY_MAX) cnt[threadIdx.x] = 0;
    __syncthreads();
    if (np > 0) {
        long want[EARLY_MAX];
#pragma unroll
        for (int m = 0; m < EARLY_MAX; m++) want[m] = m < np ? (long)pred[m] : -2L;
        const long r0 = (long)blockIdx.x * EARLY_ROWS;
        uint32_t mine[EARLY_MAX];
#pragma unroll
        for (int m = 0; m < EARLY_MAX; m++) mine[m] = 0;
        for (int i = threadIdx.x; i < EARLY_ROWS; i += WG) {
            const long r = r0 + i;
            if (r < n) {
                const long id = ids[r];
#pragma unroll
                for (int m = 0; m < EARLY_MAX; m++) mine[m] += id == want[m];
            }
        }
#pragma unroll
        for (int m = 0; m < EARLY_MAX; m++) {
            uint32_t v = mine[m];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cnt[m], v);
        }
    }
    __syncthreads();
    if (threadIdx.x < EARLY_MAX) blockcnt[(size_t)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
}

// one workgroup: blockbase[m][b] = members of pred[m] in blocks before b; eoff[m] = start of list m (lists back to back)
__global__ void __launch_bounds__(1024) early_scan_kernel(const uint32_t* __restrict__ blockcnt, int nblk,
                                                          uint32_t* __restrict__ blockbase, uint32_t* __restrict__ eoff) {
    __shared__ uint32_t part[EARLY_MAX][1024];
    __shared__ uint32_t total[EARLY_MAX];
    const int t = threadIdx.x;
    const int per = (nblk + 1023) / 1024;
    const int lo = min(nblk, t * per), hi = min(nblk, lo + per);
#pragma unroll
    for (int m = 0; m < EARLY_MAX; m++) {
        uint32_t s = 0;
        for (int b = lo; b < hi; b++) s += blockcnt[(size_t)m * nblk + b];
        part[m][t] = s;
    }
    __syncthreads();
    if (t < EARLY_MAX) {   // (serial scans side by side, over the threads that hold blocks: 64 of them at 262 144 rows)
        const int used = min(1024, (nblk + per - 1) / per);
        uint32_t run = 0;
        for (int i = 0; i < used; i++) { const uint32_t v = part[t][i]; part[t][i] = run; run += v; }
        total[t] = run;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < EARLY_MAX; m++) {
        uint32_t run = part[m][t];
        for (int b = lo; b < hi; b++) {
            blockbase[(size_t)m * nblk + b] = run;
            run += blockcnt[(size_t)m * nblk + b];
        }
    }
    if (t == 0) {
        uint32_t run = 0;
        for (int m = 0; m < EARLY_MAX; m++) { eoff[m] = run; run += total[m]; }
        eoff[EARLY_MAX] = run;
    }
}

// After the sort: every long cluster goes into the next call's early set (pred), those the early pass has not summed
// into late[] (late[0] = how many) -- so that the regular long pass is a handful of workgroups, not k x d/4 of which
// all but a few leave at once (65-95 us of dispatch at k = 8192).
__global__ void __launch_bounds__(WG) long_detect_kernel(const uint32_t* __restrict__ offsets, int k, uint32_t long_list,
                                                         const unsigned* __restrict__ done, unsigned gen,
                                                         int* __restrict__ pred, int* __restrict__ pred_n,
                                                         int* __restrict__ late) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= k) return;
    if (offsets[c + 1] - offsets[c] <= long_list) return;
    const int slot = atomicAdd(pred_n, 1);
    if (slot < EARLY_MAX) pred[slot] = c;
    if (done[c] != gen) late[1 + atomicAdd(&late[0], 1)] = c;
}

__global__ void __launch_bounds__(WG) early_write_kernel(const long* __restrict__ ids, long n, const int* __restrict__ pred,
                                                         const int* __restrict__ pred_n, int nblk,
                                                         const uint32_t* __restrict__ blockbase,
                                                         const uint32_t* __restrict__ eoff, uint32_t* __restrict__ lists,
                                                         const uint32_t* __restrict__ seg_offsets) {
    // list m starts at eoff[m] (lists back to back), or -- seg_offsets given -- at the cluster's own segment of the
    // member-list array (the bucket path: `lists` is that array)
    __shared__ uint32_t wave_cnt[WG / 64];
    __shared__ uint32_t run[EARLY_MAX];
    const int np = min(*pred_n, EARLY_MAX);
    if (np <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < np)
        run[threadIdx.x] = (seg_offsets ? seg_offsets[pred[threadIdx.x]] : eoff[threadIdx.x]) +
                           blockbase[(size_t)threadIdx.x * nblk + blockIdx.x];
    __syncthreads();
    const long r0 = (long)blockIdx.x * EARLY_ROWS;
    for (int i0 = 0; i0 < EARLY_ROWS; i0 += WG) {        // 256 consecutive rows per round, in row order
        const long r = r0 + i0 + threadIdx.x;
        const long id = r < n ? ids[r] : -1L;
        for (int m = 0; m < np; m++) {
            const bool hit = id == (long)pred[m];
            const unsigned long long b = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < WG / 64; w++) {
                before += w < wave ? wave_cnt[w] : 0u;
                all += wave_cnt[w];
            }
            if (hit) lists[run[m] + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)r;
            __syncthreads();
            if (threadIdx.x == 0) run[m] += all;
            __syncthreads();
        }
    }
}

// ---- member lists without a radix sort (the bucket path) ------------------------------------------------------
// rocPRIM's onesweep needs eight launches (histogram, two passes, their fills) whatever n is: 107 us of a 0.5 ms
// iteration at the 262 144 rows a rank holds in an 8-GPU run, 130 us at 2 M.  With k <= 16 384 clusters a row's bucket
// is known from its id alone: count per cluster (LDS histogram per 4096-row block), scan, scatter to the cluster's
// segment (slots handed out by LDS atomics: any order), then one wave per cluster restores ascending row order with
// a bitonic sort in LDS -- lists hold 30-250 rows.  Lists longer than 2048 rows never enter the scatter: their
// members come, in order, from the ordered compaction above (now driven by the exact counts of the scan instead
// of a prediction), on the side stream, so their add chains start after two small kernels.
constexpr uint32_t BK_SKIP = 0xffffffffu;

__global__ void __launch_bounds__(WG) bucket_count_kernel(const long* __restrict__ ids, long n, int k, int rows_per_block,
                                                          unsigned* __restrict__ counts) {
    extern __shared__ unsigned bk_h[];   // k + 1 bins (the last one: ids outside [0, k))
    for (int b = threadIdx.x; b <= k; b += WG) bk_h[b] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (int i = threadIdx.x; i < rows_per_block; i += WG) {
        const long r = r0 + i;
        if (r < n) {
            const long id = ids[r];
            atomicAdd(&bk_h[(id >= 0 && id < k) ? (int)id : k], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= k; b += WG)
        if (bk_h[b]) atomicAdd(&counts[b], bk_h[b]);
}

// one workgroup: offsets (exclusive scan of the k+1 counts, offsets[k+1] = n), cursor = offsets (BK_SKIP for the
// clusters the compaction serves), longs[0] = number of long clusters, longs[1..] = their ids in ascending order
// (the first EARLY_MAX are the compaction's); counts are zeroed for the next call.
__global__ void __launch_bounds__(1024) bucket_scan_kernel(unsigned* __restrict__ counts, int k, uint32_t long_list,
                                                           uint32_t* __restrict__ offsets, unsigned* __restrict__ cursor,
                                                           int* __restrict__ longs) {
    __shared__ uint32_t part[1024], lpart[1024];
    const int t = threadIdx.x;
    const int per = (k + 1 + 1023) / 1024;
    const int lo = min(k + 1, t * per), hi = min(k + 1, lo + per);
    uint32_t s = 0, nl = 0;
    for (int b = lo; b < hi; b++) {
        const uint32_t c = counts[b];
        s += c;
        nl += (b < k && c > long_list) ? 1u : 0u;
    }
    part[t] = s;
    lpart[t] = nl;
    __syncthreads();
    if (t < 64) {   // exclusive scans of the 1024 partial sums: 16 per lane, then across the wave
        uint32_t a[16], la[16], sa = 0, sl = 0;
#pragma unroll
        for (int u = 0; u < 16; u++) { a[u] = part[16 * t + u]; la[u] = lpart[16 * t + u]; sa += a[u]; sl += la[u]; }
        uint32_t xa = sa, xl = sl;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ya = __shfl_up(xa, off), yl = __shfl_up(xl, off);
            if (t >= off) { xa += ya; xl += yl; }
        }
        uint32_t ra = xa - sa, rl = xl - sl;
#pragma unroll
        for (int u = 0; u < 16; u++) { part[16 * t + u] = ra; lpart[16 * t + u] = rl; ra += a[u]; rl += la[u]; }
        if (t == 63) { offsets[k + 1] = xa; longs[0] = (int)xl; }
    }
    __syncthreads();
    uint32_t run = part[t], lrun = lpart[t];
    for (int b = lo; b < hi; b++) {
        const uint32_t c = counts[b];
        offsets[b] = run;
        const bool is_long = b < k && c > long_list;
        cursor[b] = (is_long && lrun < (uint32_t)EARLY_MAX) ? BK_SKIP : run;
        if (is_long) longs[1 + lrun++] = b;
        run += c;
        counts[b] = 0;
    }
}

__global__ void __launch_bounds__(WG) bucket_scatter_kernel(const long* __restrict__ ids, long n, int k, int rows_per_block,
                                                            unsigned* __restrict__ cursor, uint32_t* __restrict__ order) {
    extern __shared__ unsigned bk_s[];   // cnt[k+1] | base[k+1]
    unsigned* cnt = bk_s;
    unsigned* base = bk_s + (k + 1);
    for (int b = threadIdx.x; b <= k; b += WG) cnt[b] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (int i = threadIdx.x; i < rows_per_block; i += WG) {
        const long r = r0 + i;
        if (r < n) {
            const long id = ids[r];
            atomicAdd(&cnt[(id >= 0 && id < k) ? (int)id : k], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= k; b += WG) {
        const unsigned c = cnt[b];
        if (c) base[b] = cursor[b] == BK_SKIP ? BK_SKIP : atomicAdd(&cursor[b], c);
        cnt[b] = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows_per_block; i += WG) {
        const long r = r0 + i;
        if (r < n) {
            const long id = ids[r];
            const int b = (id >= 0 && id < k) ? (int)id : k;
            const unsigned bs = base[b];
            if (bs != BK_SKIP) order[bs + atomicAdd(&cnt[b], 1u)] = (uint32_t)r;
        }
    }
}

// one wave per cluster: its 2 .. 2048 members into ascending row order (bitonic sort in LDS, padded with ~0)
__global__ void __launch_bounds__(WG) member_sort_kernel(uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                                         int k, uint32_t long_list) {
    __shared__ uint32_t ms[WG / 64][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * (WG / 64) + wave;
    if (c > k) return;   // (c == k: the trailing bucket of ids outside [0, k); sorted too when it fits)
    const uint32_t beg = offsets[c], len = offsets[c + 1] - beg;
    if (len < 2 || len > long_list || len > 2048u) return;
    uint32_t P = 2;
    while (P < len) P <<= 1;
    uint32_t* s = ms[wave];
    for (uint32_t i = lane; i < P; i += 64) s[i] = i < len ? order[beg + i] : 0xffffffffu;
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __builtin_amdgcn_wave_barrier();   // (a wave's LDS operations execute in order; this keeps the compiler from moving them)
            for (uint32_t i = lane; i < P / 2; i += 64) {
                const uint32_t pos = 2 * i - (i & (stride - 1));
                const uint32_t a = s[pos], b = s[pos + stride];
                const bool up = (pos & size) == 0;
                if ((a > b) == up) { s[pos] = b; s[pos + stride] = a; }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < len; i += 64) order[beg + i] = s[i];
}

// the few long clusters beyond the compaction's EARLY_MAX: rank sort of the scattered segment by one workgroup
// (rows are distinct: rank = number of smaller rows), through a scratch copy
__global__ void __launch_bounds__(1024) long_ranksort_kernel(uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                                             const int* __restrict__ longs, uint32_t* __restrict__ scratch) {
    __shared__ uint32_t tile[1024];
    const int nl = longs[0];
    for (int slot = EARLY_MAX + blockIdx.x; slot < nl; slot += gridDim.x) {
        const int c = longs[1 + slot];
        const uint32_t beg = offsets[c], len = offsets[c + 1] - beg;
        for (uint32_t i0 = 0; i0 < len; i0 += 1024) {
            const uint32_t i = i0 + threadIdx.x;
            const uint32_t mine = i < len ? order[beg + i] : 0u;
            uint32_t rank = 0;
            for (uint32_t j0 = 0; j0 < len; j0 += 1024) {
                __syncthreads();
                tile[threadIdx.x] = j0 + threadIdx.x < len ? order[beg + j0 + threadIdx.x] : 0xffffffffu;
                __syncthreads();
                const uint32_t m = min(1024u, len - j0);
                for (uint32_t j = 0; j < m; j++) rank += tile[j] < mine;
            }
            if (i < len) scratch[beg + rank] = mine;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < len; i += 1024) order[beg + i] = scratch[beg + i];
        __syncthreads();
    }
}

// sorted_ids[p] = the cluster of position p (k for the trailing bucket of invalid ids)
// (and, when the caller wants the member order too, its copy out of the workspace in the same pass)
__global__ void __launch_bounds__(WG) segment_ids_kernel(const uint32_t* __restrict__ offsets, int k, uint32_t* __restrict__ sorted_ids,
                                                         const uint32_t* __restrict__ order, uint32_t* __restrict__ order_out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (c > k) return;
    for (uint32_t p = offsets[c] + lane; p < offsets[c + 1]; p += 64) {
        sorted_ids[p] = (uint32_t)c;
        if (order_out) order_out[p] = order[p];
    }
}

__global__ void __launch_bounds__(WG) centroid_finalize_kernel(const float* __restrict__ sums_parts,
                                                               long sums_stride,
                                                               const float* __restrict__ counts_parts,
                                                               long counts_stride, int n_parts, int k,
                                                               int d,
                                                               float* __restrict__ cent,
                                                               float* __restrict__ hassign) {
    const long e = (long)blockIdx.x * WG + threadIdx.x;
    if (e >= (long)k * d) return;
    const int c = (int)(e / d);
    float cnt = 0.0f, tot = 0.0f;
    for (int p = 0; p < n_parts; p++) {
        cnt += counts_parts[p * counts_stride + c];
        tot += sums_parts[p * sums_stride + e];
    }
    float out = 0.0f;
    if (cnt != 0.0f) {
        const float inv = 1.0f / cnt;  // IEEE division, then one multiply: faiss' "norm = 1 / hassign"
        out = tot * inv;
    }
    cent[e] = out;
    if (e == (long)c * d) hassign[c] = cnt;
}

// out[i] = ((parts[0][i] + parts[1][i]) + parts[2][i]) + ...: the rank-ordered sum of the scatter form of the exchange
__global__ void __launch_bounds__(WG) sum_parts_kernel(const float* __restrict__ parts, long stride, int n_parts, long m,
                                                       float* __restrict__ out) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= m) return;
    float tot = 0.0f;
    for (int p = 0; p < n_parts; p++) tot += parts[p * stride + i];
    out[i] = tot;
}

// ---- fixed-tree reductions ---------------------------------------------------------------------
constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// One launch: every workgroup leaves its partial sum, the last one to arrive (a counter that it also puts back to
// zero) adds the partials up in index order.  The result is a fixed function of (v, n): the grid is a function of n
// alone and every tree below is fixed, so repeated calls and all ranks agree bit for bit.  (It is NOT the partition
// round 1's two-kernel form used: the grid was capped at 256 / 1024 workgroups in round 2.  The objective it feeds is
// the one statistic the contract holds to a tolerance)
__global__ void __launch_bounds__(WG) sum_kernel(const float* __restrict__ v, long n, double* __restrict__ partial,
                                                 unsigned* __restrict__ ticket, double* __restrict__ out) {
    __shared__ double sh[WG / 64];
    __shared__ bool last;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * WG + threadIdx.x; i < n; i += (long)gridDim.x * WG) acc += (double)v[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[blockIdx.x], (sh[0] + sh[1]) + (sh[2] + sh[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;          // (uniform over the workgroup)
    __threadfence();
    const int m = (int)gridDim.x;
    acc = 0.0;
    for (int i = threadIdx.x; i < m; i += WG) acc += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        *out = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        *ticket = 0u;
    }
}

__global__ void __launch_bounds__(WG) nonfinite_kernel(const float* __restrict__ v, long n,
                                                       int32_t* __restrict__ flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * WG + threadIdx.x; i < n; i += (long)gridDim.x * WG) {
        const uint32_t bits = __float_as_uint(v[i]);
        bad |= (bits & 0x7f800000u) == 0x7f800000u;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Token histogram (spec_tokenizer.py:129-147 counts tokens with a Python
// Counter over tokens.tolist()).  Workgroup-private counts in LDS when the vocabulary fits, one
// global atomic per non-empty bin and workgroup afterwards.
constexpr int HIST_LDS_BINS = 16384;

__global__ void __launch_bounds__(WG) histogram_kernel(const long* __restrict__ ids, long n, int k,
                                                       unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int bins[];
    const bool local = k <= HIST_LDS_BINS;
    if (local) {
        for (int b = threadIdx.x; b < k; b += WG) bins[b] = 0u;
        __syncthreads();
    }
    for (long i = (long)blockIdx.x * WG + threadIdx.x; i < n; i += (long)gridDim.x * WG) {
        const long t = ids[i];
        if (t < 0 || t >= k) continue;  // rows without a token (-1) are not counted
        if (local) atomicAdd(&bins[t], 1u);
        else atomicAdd(&counts[t], 1ull);
    }
    if (local) {
        __syncthreads();
        for (int b = threadIdx.x; b < k; b += WG)
            if (bins[b]) atomicAdd(&counts[b], (unsigned long long)bins[b]);
    }
}

// at_centroid_accum_f32 by the bucket path (k <= 16 384): see the kernels above.
int accum_buckets(at_ctx* ctx, const float* x, int64_t n, int d, const int64_t* ids, int k, float* sums, float* counts,
                  uint32_t* order_out, uint32_t* sorted_ids_out, hipStream_t stream) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    const int nblk = (int)((n + EARLY_ROWS - 1) / EARLY_ROWS);
    uint32_t* order = static_cast<uint32_t*>(at_ws(ctx, WS_SORT_VALS_A, nn * 4, stream));
    uint32_t* offsets = static_cast<uint32_t*>(at_ws(ctx, WS_SEG_OFFSETS, ((size_t)k + 2) * 4, stream));
    // counts[k+1] (zero between calls: the scan clears what it has read) | cursor[k+1] | longs[k+2]
    const bool fresh = ctx->ws_bytes[WS_BUCKETS] < ((size_t)3 * k + 8) * 4;
    unsigned* bw = static_cast<unsigned*>(at_ws(ctx, WS_BUCKETS, ((size_t)3 * k + 8) * 4, stream));
    uint32_t* ew = static_cast<uint32_t*>(at_ws(ctx, WS_LONG_EARLY, ((size_t)2 * EARLY_MAX * nblk + EARLY_MAX + 1 + nn) * 4, stream));
    if (!order || !offsets || !bw || !ew) return AT_E_NOMEM;
    if (fresh || ctx->buckets_k != k) {
        AT_HIP(hipMemsetAsync(bw, 0, ((size_t)3 * k + 8) * 4, stream));
        ctx->buckets_k = k;
    }
    unsigned* bcounts = bw;
    unsigned* cursor = bw + (k + 1);
    int* longs = reinterpret_cast<int*>(bw + 2 * (k + 1));
    uint32_t* blockcnt = ew;
    uint32_t* blockbase = ew + (size_t)EARLY_MAX * nblk;
    uint32_t* scratch = blockbase + (size_t)EARLY_MAX * nblk + EARLY_MAX + 1;

    const bool al = at_aligned16(x);
    const bool long_ok = d % 4 == 0 && al;
    const uint32_t long_list = long_ok ? 2048u : UINT32_MAX;   // (without the sliced kernel every list is a short one)
    const size_t lds1 = ((size_t)k + 1) * 4, lds3 = 2 * lds1;
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&bucket_count_kernel), lds1); if (rcl_) return rcl_; }
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&bucket_scatter_kernel), lds3); if (rcl_) return rcl_; }
    const long* idl = reinterpret_cast<const long*>(ids);
    // rows per workgroup of the count / scatter passes: enough workgroups to fill the chip at small n (each pays
    // three passes over the k bins), 4096 rows at large n
    int rpb = (int)(n / 1024);
    rpb = rpb < 512 ? 512 : (rpb > 4096 ? 4096 : rpb);
    rpb = (rpb + WG - 1) / WG * WG;
    const int nbk = (int)((n + rpb - 1) / rpb);
    if (n > 0) {
        AT_LAUNCH(bucket_count_kernel, dim3(nbk), dim3(WG), lds1, stream, idl, (long)n, k, rpb, bcounts);
    }
    AT_LAUNCH(bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, bcounts, k, long_list, offsets, cursor, longs);
    const bool have_long = long_ok && n > (int64_t)long_list;
    if (have_long) {
        if (!ctx->side_stream) {
            AT_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            AT_HIP(hipEventCreateWithFlags(&ctx->side_ev[0], hipEventDisableTiming));
            AT_HIP(hipEventCreateWithFlags(&ctx->side_ev[1], hipEventDisableTiming));
        }
        // side stream: the long clusters' members by ordered compaction straight into their segments, then their sums
        hipStream_t ss = ctx->side_stream;
        AT_HIP(hipEventRecord(ctx->side_ev[0], stream));          // offsets / longs are ready
        AT_HIP(hipStreamWaitEvent(ss, ctx->side_ev[0], 0));
        AT_LAUNCH(early_count_kernel, dim3(nblk), dim3(WG), 0, ss, idl, (long)n, longs + 1, longs, nblk, blockcnt);
        AT_LAUNCH(early_scan_kernel, dim3(1), dim3(1024), 0, ss, blockcnt, nblk, blockbase, blockbase + (size_t)EARLY_MAX * nblk);
        AT_LAUNCH(early_write_kernel, dim3(nblk), dim3(WG), 0, ss, idl, (long)n, longs + 1, longs, nblk, blockbase,
                           nullptr, order, offsets);
        AT_LAUNCH(centroid_accum_long_kernel, dim3(EARLY_MAX, d / 4), dim3(WG), 0, ss, x, d, order, offsets, long_list,
                           sums, counts, k, 0, longs, nullptr, nullptr, 0u, 0, EARLY_MAX);
    }
    if (n > 0) {
        AT_LAUNCH(bucket_scatter_kernel, dim3(nbk), dim3(WG), lds3, stream, idl, (long)n, k, rpb, cursor, order);
        AT_LAUNCH(member_sort_kernel, dim3((k + 1 + WG / 64 - 1) / (WG / 64)), dim3(WG), 0, stream, order, offsets, k, 2048u);
    }
    if (have_long) {
        // more than EARLY_MAX long clusters (rare): the rest were scattered; rank-sort them, then their sums
        hipStream_t ss = ctx->side_stream;
        AT_HIP(hipEventRecord(ctx->side_ev[0], stream));          // scattered segments are ready
        AT_HIP(hipStreamWaitEvent(ss, ctx->side_ev[0], 0));
        AT_LAUNCH(long_ranksort_kernel, dim3(16), dim3(1024), 0, ss, order, offsets, longs, scratch);
        AT_LAUNCH(centroid_accum_long_kernel, dim3(16, d / 4), dim3(WG), 0, ss, x, d, order, offsets, long_list,
                           sums, counts, k, 0, longs, nullptr, nullptr, 0u, EARLY_MAX, 0x7fffffff);
        AT_HIP(hipEventRecord(ctx->side_ev[1], ss));
    }
    int vec = 1;
    if (d % 4 == 0 && d >= 256 && al) vec = 4;
    else if (d % 2 == 0 && d >= 128 && al) vec = 2;
    const int slabs = (d + 64 * vec - 1) / (64 * vec);
    const long waves = (long)k * slabs;
    const dim3 grid((unsigned)((waves + WG / 64 - 1) / (WG / 64)));
    // lists of 1025 .. 2048 rows that no local sort took (member_sort_kernel's capacity is 2048: none) -- and, when the
    // sliced kernel cannot run (d % 4 != 0), lists of any length: those need the sorted order too
    if (vec == 4)
        AT_LAUNCH(centroid_accum_kernel<4>, grid, dim3(WG), 0, stream, x, d, order, offsets, k, slabs, long_list, sums, counts);
    else if (vec == 2)
        AT_LAUNCH(centroid_accum_kernel<2>, grid, dim3(WG), 0, stream, x, d, order, offsets, k, slabs, long_list, sums, counts);
    else
        AT_LAUNCH(centroid_accum_kernel<1>, grid, dim3(WG), 0, stream, x, d, order, offsets, k, slabs, long_list, sums, counts);
    if ((order_out || sorted_ids_out) && have_long) {
        // the copies below read the long clusters' segments, which the side stream writes
        AT_HIP(hipStreamWaitEvent(stream, ctx->side_ev[1], 0));
    }
    // A caller that wants the member order gets the long clusters' segments too, and those are written on the side
    // stream: its work is waited for BEFORE the order leaves the workspace (round 1 queued the copy first and waited
    // right after it -- the same wait, one statement too late).
    const bool wants_order = (order_out || sorted_ids_out) && n > 0;
    if (have_long && wants_order) AT_HIP(hipStreamWaitEvent(stream, ctx->side_ev[1], 0));
    if (order_out && n > 0 && !sorted_ids_out)
        AT_HIP(hipMemcpyAsync(order_out, order, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    if (sorted_ids_out && n > 0) {   // (the segments [offsets[0], offsets[k+1]) cover every position: the copy rides along)
        AT_LAUNCH(segment_ids_kernel, dim3((k + 1 + WG / 64 - 1) / (WG / 64)), dim3(WG), 0, stream, offsets, k, sorted_ids_out,
                           order, order_out);
    }
    if (have_long && !wants_order) {
        if (ctx->defer_join) ctx->join_pending = 1;
        else AT_HIP(hipStreamWaitEvent(stream, ctx->side_ev[1], 0));
    }
    return AT_OK;
}

}  // namespace

extern "C" {

int at_gather_rows_f32(at_ctx* ctx, const float* x, int d, const int32_t* idx, int64_t m, float* out,
                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx, "at_gather_rows_f32: ctx is null");
    AT_REQUIRE(m >= 0 && d > 0, "at_gather_rows_f32: bad sizes");
    if (m == 0) return AT_OK;
    AT_REQUIRE(x && idx && out, "at_gather_rows_f32: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    if (d % 4 == 0 && at_aligned16(x) && at_aligned16(out)) {
        const long total = m * (d / 4);
        AT_LAUNCH(gather_rows_kernel, dim3((unsigned)((total + WG - 1) / WG)), dim3(WG), 0,
                           stream, x, d / 4, idx, (long)m, out);
    } else {
        const long total = m * d;
        AT_LAUNCH(gather_rows_scalar_kernel, dim3((unsigned)((total + WG - 1) / WG)), dim3(WG),
                           0, stream, x, d, idx, (long)m, out);
    }
    return AT_OK;
}

int at_centroid_accum_f32(at_ctx* ctx, const float* x, int64_t n, int d, const int64_t* ids, int k,
                          float* sums, float* counts, uint32_t* order_out, uint32_t* sorted_ids_out,
                          void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx, "at_centroid_accum_f32: ctx is null");
    AT_REQUIRE(n >= 0 && n < (int64_t)UINT32_MAX && d > 0 && k > 0 && k < (1 << 30),
               "at_centroid_accum_f32: bad sizes n=%lld d=%d k=%d", (long long)n, d, k);
    AT_REQUIRE(sums && counts && (n == 0 || (x && ids)), "at_centroid_accum_f32: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    // The bucket path pays off where the radix sort's eight launches are fixed cost: few rows per cluster (the
    // per-rank share of a sharded run).  At 2 M rows its scattered 4-byte stores (134 us) and the in-LDS order of 1000+
    // row lists (159 us) lose to two onesweep passes.  (It hands lists longer than 2048 rows to the feature-sliced
    // kernel: that needs d % 4 == 0.)
    if (ctx->dbg.accum_buckets != 0 && k <= 16384 && n <= 64 * (int64_t)k && d % 4 == 0 && at_aligned16(x))
        return accum_buckets(ctx, x, n, d, ids, k, sums, counts, order_out, sorted_ids_out, stream);

    const size_t nn = (size_t)(n > 0 ? n : 1);
    uint32_t* keys_a = static_cast<uint32_t*>(at_ws(ctx, WS_SORT_KEYS_A, nn * 4, stream));
    uint32_t* keys_b = static_cast<uint32_t*>(at_ws(ctx, WS_SORT_KEYS_B, nn * 4, stream));
    uint32_t* vals_a = static_cast<uint32_t*>(at_ws(ctx, WS_SORT_VALS_A, nn * 4, stream));
    uint32_t* vals_b = static_cast<uint32_t*>(at_ws(ctx, WS_SORT_VALS_B, nn * 4, stream));
    uint32_t* offsets = static_cast<uint32_t*>(at_ws(ctx, WS_SEG_OFFSETS, ((size_t)k + 2) * 4, stream));
    if (!keys_a || !keys_b || !vals_a || !vals_b || !offsets) return AT_E_NOMEM;

    // (decided before anything is queued: the early lists of the long clusters run on the side stream beside the sort)
    const bool al_x = at_aligned16(x);
    const bool early_ok = d % 4 == 0 && al_x && n > 2048;
    bool offsets_beside = false;
    if (early_ok) {
        if (!ctx->side_stream) {
            AT_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            AT_HIP(hipEventCreateWithFlags(&ctx->side_ev[0], hipEventDisableTiming));
            AT_HIP(hipEventCreateWithFlags(&ctx->side_ev[1], hipEventDisableTiming));
        }
        int* pw = static_cast<int*>(at_ws(ctx, WS_LONG_PRED, LongPred::bytes(k), stream));
        const int nblk = (int)((n + EARLY_ROWS - 1) / EARLY_ROWS);
        uint32_t* ew = static_cast<uint32_t*>(at_ws(ctx, WS_LONG_EARLY, ((size_t)2 * EARLY_MAX * nblk + EARLY_MAX + 1 + nn) * 4, stream));
        if (!pw || !ew) return AT_E_NOMEM;
        if (ctx->long_pred_k != k) {
            AT_HIP(hipMemsetAsync(pw, 0, LongPred::bytes(k), stream));
            ctx->long_pred_k = k;
            ctx->long_gen = 0;
        }
        const LongPred lp(pw);
        uint32_t* blockcnt = ew;
        uint32_t* blockbase = ew + (size_t)EARLY_MAX * nblk;
        uint32_t* eoff = blockbase + (size_t)EARLY_MAX * nblk;
        uint32_t* lists = eoff + EARLY_MAX + 1;
        const unsigned gen = ctx->long_gen + 1;   // (the regular pass below takes the same generation)
        AT_HIP(hipEventRecord(ctx->side_ev[0], stream));          // ids (and the marks) are ready
        AT_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
        hipStream_t ss = ctx->side_stream;
        if (k <= 16384) {
            // The segment offsets do not need the sort: counts per cluster (LDS histogram per row block) and their scan
            // -- the first two kernels of the bucket path -- run beside it, instead of 8193 binary searches over the
            // sorted keys behind it (40 us on the critical path of a 2 M-row iteration).
            const bool fresh = ctx->ws_bytes[WS_BUCKETS] < ((size_t)3 * k + 8) * 4;
            unsigned* bw = static_cast<unsigned*>(at_ws(ctx, WS_BUCKETS, ((size_t)3 * k + 8) * 4, stream));
            if (!bw) return AT_E_NOMEM;
            if (fresh || ctx->buckets_k != k) {
                AT_HIP(hipMemsetAsync(bw, 0, ((size_t)3 * k + 8) * 4, ss));
                ctx->buckets_k = k;
            }
            const size_t lds1 = ((size_t)k + 1) * 4;
            { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&bucket_count_kernel), lds1); if (rcl_) return rcl_; }
            int rpb = (int)(n / 1024);
            rpb = rpb < 512 ? 512 : (rpb > 4096 ? 4096 : rpb);
            rpb = (rpb + WG - 1) / WG * WG;
            AT_LAUNCH(bucket_count_kernel, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(WG), lds1, ss, reinterpret_cast<const long*>(ids),
                      (long)n, k, rpb, bw);
            AT_LAUNCH(bucket_scan_kernel, dim3(1), dim3(1024), 0, ss, bw, k, 2048u, offsets, bw + (k + 1),
                      reinterpret_cast<int*>(bw + 2 * (k + 1)));
            if (!ctx->side_ev2) AT_HIP(hipEventCreateWithFlags(&ctx->side_ev2, hipEventDisableTiming));
            AT_HIP(hipEventRecord(ctx->side_ev2, ss));
            offsets_beside = true;
        }
        AT_LAUNCH(early_count_kernel, dim3(nblk), dim3(WG), 0, ss, reinterpret_cast<const long*>(ids), (long)n, lp.pred, lp.pred_n,
                           nblk, blockcnt);
        AT_LAUNCH(early_scan_kernel, dim3(1), dim3(1024), 0, ss, blockcnt, nblk, blockbase, eoff);
        AT_LAUNCH(early_write_kernel, dim3(nblk), dim3(WG), 0, ss, reinterpret_cast<const long*>(ids), (long)n, lp.pred, lp.pred_n,
                           nblk, blockbase, eoff, lists, nullptr);
        AT_LAUNCH(centroid_accum_long_kernel, dim3(EARLY_MAX, d / 4), dim3(WG), 0, ss, x, d, lists, eoff, 2048u, sums,
                           counts, k, 1, lp.pred, lp.pred_n, lp.done, gen, 0, 0x7fffffff);
        // the regular pass rebuilds the prediction: its counter starts from zero once the early pass has read it
        AT_HIP(hipMemsetAsync(lp.pred_n, 0, 4, ss));
    }

    const uint32_t* order = vals_a;
    const uint32_t* sorted_keys = keys_a;
    if (n > 0) {
        AT_LAUNCH(make_keys_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, stream,
                           reinterpret_cast<const long*>(ids), (long)n, k, keys_a, vals_a);
        unsigned bits = 1;
        while ((1u << bits) <= (unsigned)k) bits++;  // keys take values 0..k
        rocprim::double_buffer<uint32_t> kb(keys_a, keys_b);
        rocprim::double_buffer<uint32_t> vb(vals_a, vals_b);
        size_t tmp_bytes = 0;
        AT_HIP(rocprim::radix_sort_pairs<at_radix_config>(nullptr, tmp_bytes, kb, vb, (size_t)n, 0, bits, stream));
        void* tmp = at_ws(ctx, WS_SORT_TMP, tmp_bytes, stream);
        if (!tmp) return AT_E_NOMEM;
        AT_HIP(rocprim::radix_sort_pairs<at_radix_config>(tmp, tmp_bytes, kb, vb, (size_t)n, 0, bits, stream));
        order = vb.current();
        sorted_keys = kb.current();
    }
    if (offsets_beside)
        AT_HIP(hipStreamWaitEvent(stream, ctx->side_ev2, 0));
    else
        AT_LAUNCH(segment_offsets_kernel, dim3((k + 1 + WG - 1) / WG), dim3(WG), 0, stream,
                           sorted_keys, (long)n, k, offsets);

    const bool al = at_aligned16(x);
    int vec = 1;
    if (d % 4 == 0 && d >= 256 && al) vec = 4;
    else if (d % 2 == 0 && d >= 128 && al) vec = 2;
    const int slabs = (d + 64 * vec - 1) / (64 * vec);
    const long waves = (long)k * slabs;
    const dim3 grid((unsigned)((waves + WG / 64 - 1) / (WG / 64)));
    // member lists longer than this go to the feature-sliced workgroup kernel
    const bool long_ok = d % 4 == 0 && al;
    const uint32_t long_list = long_ok ? 2048u : UINT32_MAX;
    // The long lists are one dependent add chain per feature (the contract's summation order), a few
    // workgroups busy for as long as the longest list takes: they run on a side stream beside the
    // kernel that handles all the other clusters.
    const bool have_long = long_ok && n > (int64_t)long_list;
    if (have_long) {
        AT_HIP(hipEventRecord(ctx->side_ev[0], stream));          // sorted lists and offsets are ready
        AT_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
        // pred / pred_n / done live in WS_LONG_PRED (struct LongPred)
        int* pw = static_cast<int*>(at_ws(ctx, WS_LONG_PRED, LongPred::bytes(k), stream));
        if (!pw) return AT_E_NOMEM;
        if (ctx->long_pred_k != k) {   // fresh (or another table size): no predictions, no marks
            AT_HIP(hipMemsetAsync(pw, 0, LongPred::bytes(k), stream));
            ctx->long_pred_k = k;
            ctx->long_gen = 0;
        }
        const unsigned gen = ++ctx->long_gen;
        const LongPred lp(pw);
        int* pred_n = lp.pred_n;
        int* pred = lp.pred;
        unsigned* done = lp.done;
        int* late = static_cast<int*>(at_ws(ctx, WS_LONG_LATE, ((size_t)k + 1) * 4, stream));
        if (!late) return AT_E_NOMEM;
        AT_HIP(hipMemsetAsync(late, 0, 4, ctx->side_stream));
        AT_LAUNCH(long_detect_kernel, dim3((k + WG - 1) / WG), dim3(WG), 0, ctx->side_stream, offsets, k, long_list,
                           done, gen, pred, pred_n, late);
        AT_LAUNCH(centroid_accum_long_kernel, dim3(32, d / 4), dim3(WG), 0, ctx->side_stream, x, d, order,
                           offsets, long_list, sums, counts, k, 0, late, nullptr, done, gen, 0, 0x7fffffff);
        AT_HIP(hipEventRecord(ctx->side_ev[1], ctx->side_stream));
    }
    if (vec == 4)
        AT_LAUNCH(centroid_accum_kernel<4>, grid, dim3(WG), 0, stream, x, d, order, offsets, k,
                           slabs, long_list, sums, counts);
    else if (vec == 2)
        AT_LAUNCH(centroid_accum_kernel<2>, grid, dim3(WG), 0, stream, x, d, order, offsets, k,
                           slabs, long_list, sums, counts);
    else
        AT_LAUNCH(centroid_accum_kernel<1>, grid, dim3(WG), 0, stream, x, d, order, offsets, k,
                           slabs, long_list, sums, counts);
    if (order_out && n > 0)
        AT_HIP(hipMemcpyAsync(order_out, order, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    if (sorted_ids_out && n > 0)
        AT_HIP(hipMemcpyAsync(sorted_ids_out, sorted_keys, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice,
                              stream));
    // The caller's stream must not read `sums` before the side stream is done.  With order_out == the
    // special "deferred" protocol (at_centroid_accum_join) the wait is left to the caller, who can queue
    // independent work (the next iteration's visiting order) behind the short-list kernel meanwhile.
    if (have_long) {
        if (ctx->defer_join) ctx->join_pending = 1;
        else AT_HIP(hipStreamWaitEvent(stream, ctx->side_ev[1], 0));
    }
    return AT_OK;
}

int at_centroid_finalize_f32(at_ctx* ctx, const float* sums_parts, int64_t sums_part_stride,
                             const float* counts_parts, int64_t counts_part_stride, int n_parts,
                             int k, int d, float* centroids, float* hassign, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx, "at_centroid_finalize_f32: ctx is null");
    AT_REQUIRE(n_parts >= 1 && k > 0 && d > 0, "at_centroid_finalize_f32: bad sizes");
    AT_REQUIRE(sums_parts && counts_parts && centroids && hassign, "at_centroid_finalize_f32: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    const long total = (long)k * d;
    AT_LAUNCH(centroid_finalize_kernel, dim3((unsigned)((total + WG - 1) / WG)), dim3(WG), 0,
                       stream, sums_parts, (long)sums_part_stride, counts_parts,
                       (long)counts_part_stride, n_parts, k, d, centroids, hassign);
    return AT_OK;
}

int at_sum_parts_f32(at_ctx* ctx, const float* parts, int64_t part_stride, int n_parts, int64_t m, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && n_parts >= 1 && m >= 0 && (m == 0 || (parts && out)), "at_sum_parts_f32: bad arguments");
    if (m == 0) return AT_OK;
    AT_HIP(hipSetDevice(ctx->device));
    AT_LAUNCH(sum_parts_kernel, dim3((unsigned)((m + WG - 1) / WG)), dim3(WG), 0, stream, parts, (long)part_stride,
                       n_parts, (long)m, out);
    return AT_OK;
}

int at_sum_f32(at_ctx* ctx, const float* v, int64_t n, double* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && out && n >= 0 && (n == 0 || v), "at_sum_f32: bad arguments");
    AT_HIP(hipSetDevice(ctx->device));
    double* partial = static_cast<double*>(at_ws(ctx, WS_REDUCE, RED_BLOCKS * sizeof(double), stream));
    if (!partial) return AT_E_NOMEM;
    // one workgroup per CU at most: every workgroup ends with an atomic on the one arrival counter, and a thousand of
    // those in a row cost more (15 us) than the sum itself
    int blocks = (int)((n + 4 * WG - 1) / (4 * WG));
    const int cap = n <= ((int64_t)1 << 22) ? 256 : RED_BLOCKS;   // (long vectors want the bandwidth of more workgroups)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const bool fresh = ctx->ws[WS_SUM_TICKET] == nullptr;
    unsigned* ticket = static_cast<unsigned*>(at_ws(ctx, WS_SUM_TICKET, 16, stream));
    if (!ticket) return AT_E_NOMEM;
    if (fresh) AT_HIP(hipMemsetAsync(ticket, 0, 16, stream));
    // The partial buffer and the arrival counter are the context's: two calls on different streams must not overlap
    // (the wrong "last workgroup", a mixed sum, a counter that is never put back).  A call on another stream than the
    // previous one waits for it.
    if (!ctx->sum_ev) AT_HIP(hipEventCreateWithFlags(&ctx->sum_ev, hipEventDisableTiming));
    if (ctx->sum_used && ctx->sum_stream != stream) AT_HIP(hipStreamWaitEvent(stream, ctx->sum_ev, 0));
    AT_LAUNCH(sum_kernel, dim3(blocks), dim3(WG), 0, stream, v, (long)n, partial, ticket, out);
    AT_HIP(hipEventRecord(ctx->sum_ev, stream));
    ctx->sum_stream = stream;
    ctx->sum_used = 1;
    return AT_OK;
}

int at_any_nonfinite_f32(at_ctx* ctx, const float* v, int64_t n, int32_t* flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && flag && n >= 0 && (n == 0 || v), "at_any_nonfinite_f32: bad arguments");
    AT_HIP(hipSetDevice(ctx->device));
    AT_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), stream));
    if (n == 0) return AT_OK;
    int blocks = (int)((n + WG - 1) / WG);
    if (blocks > 2048) blocks = 2048;
    AT_LAUNCH(nonfinite_kernel, dim3(blocks), dim3(WG), 0, stream, v, (long)n, flag);
    return AT_OK;
}

int at_token_histogram_i64(at_ctx* ctx, const int64_t* ids, int64_t n, int k, int64_t* counts, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && counts && k > 0 && n >= 0 && (n == 0 || ids), "at_token_histogram_i64: bad arguments");
    AT_HIP(hipSetDevice(ctx->device));
    AT_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)k, stream));
    if (n == 0) return AT_OK;
    int blocks = (int)((n + WG * 16 - 1) / (WG * 16));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const size_t lds = k <= HIST_LDS_BINS ? sizeof(unsigned int) * (size_t)k : 0;
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&histogram_kernel), lds); if (rcl_) return rcl_; }
    AT_LAUNCH(histogram_kernel, dim3(blocks), dim3(WG), lds, stream, reinterpret_cast<const long*>(ids), (long)n, k,
                       reinterpret_cast<unsigned long long*>(counts));
    return AT_OK;
}

// at_centroid_accum_defer(ctx, 1): following at_centroid_accum_f32 calls return without making `stream` wait
// for the long-list kernel on the context's side stream; at_centroid_accum_join(ctx, stream) inserts that
// wait (a no-op when nothing is pending) and must precede any use of the sums / counts.
int at_centroid_accum_defer(at_ctx* ctx, int on) {
    AT_REQUIRE(ctx, "at_centroid_accum_defer: ctx is null");
    ctx->defer_join = on ? 1 : 0;
    return AT_OK;
}

int at_centroid_accum_join(at_ctx* ctx, void* stream_) {
    AT_REQUIRE(ctx, "at_centroid_accum_join: ctx is null");
    if (!ctx->join_pending) return AT_OK;
    AT_HIP(hipSetDevice(ctx->device));
    AT_HIP(hipStreamWaitEvent((hipStream_t)stream_, ctx->side_ev[1], 0));
    ctx->join_pending = 0;
    return AT_OK;
}

}  // extern "C"
