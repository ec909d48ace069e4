// lloyd.hip -- the pieces of a Lloyd iteration that used to need the host between two sweeps.
//
// Reference call site: processors/cluster_creator.py:54,56 -> faiss.Kmeans.train -> Clustering::train_encoded
// (faiss 1.8.0 Clustering.cpp): after compute_centroids every iteration runs split_clusters (re-seed the empty
// clusters from randomly accepted donors, RandomGenerator(1234)) and records {obj, imbalance_factor, nsplit}.
//
// split_clusters is sequential and RNG-driven, so round 1 read the k counts back every iteration to decide on
// the host whether to run it: one stream synchronisation per iteration, 20 per train().  Here one workgroup
// does the whole repair on the device -- the mt19937 stream regenerated 624 draws at a time in LDS
// (mt19937_dev.h), the cyclic acceptance scan of a donor tested 256 candidates per step with a "first lane
// that accepts" reduction, which consumes exactly the draws the sequential loop consumes -- and the statistics
// of all iterations are read back once, after the last one.  With no empty cluster (the usual case) the
// kernel is one pass over the counts.
#include <climits>

#include "at_internal.h"
#include "mt19937_dev.h"

namespace {

constexpr int WG = 256;
// A full cycle over the clusters accepts a donor with probability 1 - prod(1 - p_c) >= 1 - 1/e (the p_c sum to 1),
// so a scan still empty-handed after 64 cycles (chance < 1e-27) has no donor to find: the kernel reports -1
// instead of spinning (every wave reaches this exit).
constexpr long SPLIT_CYCLES = 64;
constexpr int SPLIT_BLOCKS = 16;
constexpr int SPLIT_ROUND = SPLIT_BLOCKS * WG;   // candidates tested per step while the draws come from the context's cache
// (a repair scans 1 / mean(p) = k candidates on average and a step costs one trip to L2 and two barriers whatever its
// width: 16 blocks instead of round 2's first 4 took the 160 repairs of a cold first iteration from 2.1-4.2 ms to a quarter)

// RandomGenerator::rand_float: mt() / float(mt.max()) -- float(2^32 - 1) is 2^32
__device__ __forceinline__ float rand_float_of(uint32_t raw) { return __uint2float_rn(raw) * 2.3283064365386963e-10f; }

// stats[0] = sum over the parts (ascending) of the double at obj_parts[p * stride];
// stats[1] = faiss imbalance_factor = k * sum(h^2) / (sum h)^2 (the sums are of integers < 2^53: exact in any order)
// (one workgroup of WG threads; every thread calls it)
__device__ void lloyd_stats_block(const float* __restrict__ hassign, int k, const double* obj_parts, long obj_stride,
                                  int n_parts, double* __restrict__ stats) {
    __shared__ double s1[WG / 64], s2[WG / 64];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int c = t; c < k; c += WG) {
        const double h = (double)hassign[c];
        a += h;
        b += h * h;
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
    }
    if ((t & 63) == 0) { s1[t >> 6] = a; s2[t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
        double tot = 0.0, sq = 0.0;
        for (int w = 0; w < WG / 64; w++) { tot += s1[w]; sq += s2[w]; }
        double obj = 0.0;
        for (int p = 0; p < n_parts; p++) obj += obj_parts[(size_t)p * obj_stride];
        stats[0] = obj;
        stats[1] = sq * k / (tot * tot);
    }
    __syncthreads();
}

__global__ __launch_bounds__(WG) void lloyd_stats_kernel(const float* __restrict__ hassign, int k, const double* obj_parts,
                                                        long obj_stride, int n_parts, double* __restrict__ stats) {
    lloyd_stats_block(hassign, k, obj_parts, obj_stride, n_parts, stats);
}

// raw / raw_n: the first raw_n outputs of mt19937(1234), resident (at_mt_cached_draws); state_end: the generator's
// 624 state words after them, from which the kernel goes on by itself if a repair ever needs more.
// p_lds != 0: the acceptance probabilities of all k clusters live in LDS (k floats of dynamic shared memory).
__global__ __launch_bounds__(WG) void split_clusters_kernel(int d, int k, long n, float* hassign, float* cent,
                                                           int* __restrict__ empties, int* __restrict__ nsplit_out,
                                                           const uint32_t* __restrict__ raw, long raw_n,
                                                           const uint32_t* __restrict__ state_end, int p_lds,
                                                           const double* obj_parts, long obj_stride, int n_parts,
                                                           double* __restrict__ stats) {
    extern __shared__ float pl[];
    __shared__ at_mt::State mt;
    __shared__ float draws[at_mt::N];
    __shared__ int wave_val[4][WG / 64];
    __shared__ int n_empty;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // (optional) the iteration's statistics first: they describe the counts as the assignment left them
    if (stats) lloyd_stats_block(hassign, k, obj_parts, obj_stride, n_parts, stats);
    // the usual iteration has no empty cluster: one pass and one barrier settle that
    {
        bool any = false;
        for (int c = t; c < k; c += WG) any |= hassign[c] == 0.0f;
        if (!__syncthreads_or(any)) {
            if (t == 0) *nsplit_out = 0;
            return;
        }
    }
    // ordered list of the clusters that came out empty (a repair never empties or fills another one)
    if (t == 0) n_empty = 0;
    __syncthreads();
    for (int base = 0; base < k; base += WG) {
        const int c = base + t;
        const bool e = c < k && hassign[c] == 0.0f;
        const unsigned long long b = __ballot(e);
        if (lane == 0) wave_val[0][wave] = __popcll(b);
        __syncthreads();
        int off = n_empty;
        for (int w = 0; w < wave; w++) off += wave_val[0][w];
        if (e) empties[off + __popcll(b & ((1ull << lane) - 1ull))] = c;
        __syncthreads();
        if (t == 0) n_empty += wave_val[0][0] + wave_val[0][1] + wave_val[0][2] + wave_val[0][3];
        __syncthreads();
    }
    const int ne = n_empty;
    if (ne == 0) {
        if (t == 0) *nsplit_out = 0;
        return;
    }

    const double denom = (double)(float)(n - k);
    auto prob = [&](int c) { return (float)(((double)hassign[c] - 1.0) / denom); };   // faiss: (hassign[cj] - 1.0) / (float)(n - k)
    if (p_lds) {
        for (int c = t; c < k; c += WG) pl[c] = prob(c);
        __syncthreads();
    }
    const double up = 1.0 + 1.0 / 1024.0, down = 1.0 - 1.0 / 1024.0;
    long pos = 0;            // draws of the stream consumed so far
    int cur = 0, win = 0, wlen = 0;   // slow path: `draws` holds wlen draws of which win are consumed; cur = state copy in use
    bool own = false;
    int done = 0;
    for (int e = 0; e < ne; e++) {
        const int ci = empties[e];
        int cj0 = 0, donor = -1;
        long left = SPLIT_CYCLES * k + 1024;
        while (left > 0 && donor < 0) {
            if (pos + SPLIT_ROUND <= raw_n) {
                // SPLIT_BLOCKS blocks of 256 candidates at once, draws straight from the resident stream; candidate j of
                // the step is tested by thread j % 256 as its (j / 256)-th, so a wave's earliest acceptance is in its
                // first block with any, and the step's earliest is the smallest of the four waves'
                uint32_t draw[SPLIT_BLOCKS];
#pragma unroll
                for (int i = 0; i < SPLIT_BLOCKS; i++) draw[i] = raw[pos + i * WG + t];
                int mine = INT_MAX;
#pragma unroll
                for (int i = 0; i < SPLIT_BLOCKS; i++) {
                    const int j = i * WG + t;
                    int cj = cj0 + j;
                    cj -= (cj / k) * k;
                    const float pc = p_lds ? pl[cj] : prob(cj);
                    const bool acc = rand_float_of(draw[i]) < pc;
                    const unsigned long long b = __ballot(acc);
                    if (b && mine == INT_MAX) mine = i * WG + wave * 64 + (__ffsll((long long)b) - 1);
                }
                if (lane == 0) wave_val[0][wave] = mine;
                __syncthreads();
                int first = INT_MAX;
#pragma unroll
                for (int w = 0; w < WG / 64; w++) first = min(first, wave_val[0][w]);
                __syncthreads();
                if (first != INT_MAX) {
                    donor = (int)((cj0 + (long)first) % k);
                    pos += first + 1;
                } else {
                    pos += SPLIT_ROUND;
                    left -= SPLIT_ROUND;
                    cj0 = (int)((cj0 + (long)SPLIT_ROUND) % k);
                }
                continue;
            }
            // the tail of the resident stream, then the kernel's own generator: 624 draws at a time through LDS
            if (win == wlen) {
                if (pos < raw_n) {
                    wlen = (int)min((long)at_mt::N, raw_n - pos);
                    for (int q = t; q < wlen; q += WG) draws[q] = rand_float_of(raw[pos + q]);
                } else {
                    wlen = at_mt::N;
                    if (!own) {
                        for (int q = t; q < at_mt::N; q += WG) mt.st[0][q] = state_end[q];
                        __syncthreads();
                        own = true;
                        cur = 0;
                    }
                    const uint32_t* nw = at_mt::regenerate(mt, cur);
                    cur ^= 1;
                    for (int q = t; q < at_mt::N; q += WG) draws[q] = rand_float_of(at_mt::temper(nw[q]));
                }
                __syncthreads();
                win = 0;
            }
            const int chunk = min(wlen - win, WG);
            bool acc = false;
            if (t < chunk) {
                const int cj = (cj0 + t) % k;
                acc = draws[win + t] < (p_lds ? pl[cj] : prob(cj));
            }
            const unsigned long long b = __ballot(acc);
            if (lane == 0) wave_val[0][wave] = b ? wave * 64 + (__ffsll((long long)b) - 1) : INT_MAX;
            __syncthreads();
            const int first = min(min(wave_val[0][0], wave_val[0][1]), min(wave_val[0][2], wave_val[0][3]));
            __syncthreads();
            if (first < chunk) {
                donor = (cj0 + first) % k;
                win += first + 1;
                pos += first + 1;
            } else {
                win += chunk;
                pos += chunk;
                left -= chunk;
                cj0 = (cj0 + chunk) % k;
            }
        }
        if (donor < 0) break;
        // (a window opened in the slow path stays valid only while pos stays inside it: the fast path above is
        // entered again only when a whole round fits in the resident stream, i.e. never after the tail began)
        float* dst = cent + (size_t)ci * d;
        float* src = cent + (size_t)donor * d;
        for (int j = t; j < d; j += WG) {
            const double a = (double)src[j];
            dst[j] = (float)(a * ((j & 1) ? down : up));
            src[j] = (float)(a * ((j & 1) ? up : down));
        }
        if (t == 0) {
            const float half = hassign[donor] / 2;
            hassign[ci] = half;
            hassign[donor] -= half;
        }
        done++;
        __syncthreads();
        if (p_lds && t == 0) {
            pl[ci] = prob(ci);
            pl[donor] = prob(donor);
        }
        __syncthreads();
    }
    if (t == 0) *nsplit_out = done == ne ? ne : -1;
}

}  // namespace

extern "C" {

int at_split_clusters_f32(at_ctx* ctx, int d, int k, int64_t n, float* hassign, float* centroids, int32_t* nsplit_out,
                          void* stream_) {
    return at_lloyd_stats_split_f32(ctx, d, k, n, hassign, centroids, nsplit_out, nullptr, 0, 0, nullptr, stream_);
}

int at_lloyd_stats_split_f32(at_ctx* ctx, int d, int k, int64_t n, float* hassign, float* centroids, int32_t* nsplit_out,
                             const double* obj_parts, int64_t obj_part_stride, int n_parts, double* stats, void* stream_) {
    AT_REQUIRE(ctx && hassign && centroids && nsplit_out && d > 0 && k > 0 && n >= k, "at_split_clusters_f32: bad arguments");
    AT_REQUIRE(!stats || (obj_parts && n_parts >= 1 && (reinterpret_cast<uintptr_t>(obj_parts) & 7u) == 0 &&
                          (reinterpret_cast<uintptr_t>(stats) & 7u) == 0),
               "at_lloyd_stats_split_f32: obj_parts / stats must be given together, 8-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AT_HIP(hipSetDevice(ctx->device));
    int* empties = static_cast<int*>(at_ws(ctx, WS_SPLIT_LIST, (size_t)k * sizeof(int), stream));
    if (!empties) return AT_E_NOMEM;
    const uint32_t *raw = nullptr, *state_end = nullptr;
    int64_t raw_n = 0;
    int rc = at_mt_cached_draws(ctx, 1234u, stream, &raw, &raw_n, &state_end);
    if (rc) return rc;
    const int p_lds = k <= 16384;
    const size_t lds = p_lds ? (size_t)k * sizeof(float) : 0;
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&split_clusters_kernel), lds); if (rcl_) return rcl_; }
    AT_LAUNCH(split_clusters_kernel, dim3(1), dim3(WG), lds, stream, d, k, (long)n, hassign, centroids, empties,
                       nsplit_out, raw, (long)raw_n, state_end, p_lds, obj_parts, (long)obj_part_stride, n_parts, stats);
    return AT_OK;
}

int at_lloyd_stats_f64(at_ctx* ctx, const float* hassign, int k, const double* obj_parts, int64_t obj_part_stride,
                       int n_parts, double* stats, void* stream_) {
    AT_REQUIRE(ctx && hassign && obj_parts && stats && k > 0 && n_parts >= 1, "at_lloyd_stats_f64: bad arguments");
    AT_REQUIRE((reinterpret_cast<uintptr_t>(obj_parts) & 7u) == 0 && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0,
               "at_lloyd_stats_f64: obj_parts / stats must be 8-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AT_HIP(hipSetDevice(ctx->device));
    AT_LAUNCH(lloyd_stats_kernel, dim3(1), dim3(WG), 0, stream, hassign, k, obj_parts, (long)obj_part_stride,
                       n_parts, stats);
    return AT_OK;
}

}  // extern "C"
