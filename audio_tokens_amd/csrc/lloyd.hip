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

__global__ __launch_bounds__(WG) void split_clusters_kernel(int d, int k, long n, float* hassign, float* cent,
                                                           int* __restrict__ empties, int* __restrict__ nsplit_out) {
    __shared__ at_mt::State mt;
    __shared__ float draws[at_mt::N];
    __shared__ int wave_val[WG / 64];
    __shared__ int n_empty;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // ordered list of the clusters that came out empty (a repair never empties or fills another one)
    if (t == 0) n_empty = 0;
    __syncthreads();
    for (int base = 0; base < k; base += WG) {
        const int c = base + t;
        const bool e = c < k && hassign[c] == 0.0f;
        const unsigned long long b = __ballot(e);
        if (lane == 0) wave_val[wave] = __popcll(b);
        __syncthreads();
        int off = n_empty;
        for (int w = 0; w < wave; w++) off += wave_val[w];
        if (e) empties[off + __popcll(b & ((1ull << lane) - 1ull))] = c;
        __syncthreads();
        if (t == 0) n_empty += wave_val[0] + wave_val[1] + wave_val[2] + wave_val[3];
        __syncthreads();
    }
    const int ne = n_empty;
    if (ne == 0) {
        if (t == 0) *nsplit_out = 0;
        return;
    }

    at_mt::seed(mt, 1234u);
    int cur = 0, pos = at_mt::N;
    const double denom = (double)(float)(n - k);
    const double up = 1.0 + 1.0 / 1024.0, down = 1.0 - 1.0 / 1024.0;
    int done = 0;
    for (int e = 0; e < ne; e++) {
        const int ci = empties[e];
        int cj0 = 0, donor = -1;
        long left = SPLIT_CYCLES * k + 1024;
        while (left > 0) {
            if (pos == at_mt::N) {
                const uint32_t* nw = at_mt::regenerate(mt, cur);
                cur ^= 1;
                // RandomGenerator::rand_float: mt() / float(mt.max()) -- float(2^32 - 1) is 2^32
                for (int q = t; q < at_mt::N; q += WG) draws[q] = __uint2float_rn(at_mt::temper(nw[q])) * 2.3283064365386963e-10f;
                __syncthreads();
                pos = 0;
            }
            const int chunk = min(at_mt::N - pos, WG);
            bool acc = false;
            if (t < chunk) {
                const int cj = (cj0 + t) % k;
                const float p = (float)(((double)hassign[cj] - 1.0) / denom);
                acc = draws[pos + t] < p;
            }
            const unsigned long long b = __ballot(acc);
            if (lane == 0) wave_val[wave] = b ? wave * 64 + (__ffsll((long long)b) - 1) : INT_MAX;
            __syncthreads();
            const int first = min(min(wave_val[0], wave_val[1]), min(wave_val[2], wave_val[3]));
            __syncthreads();
            if (first < chunk) {
                donor = (cj0 + first) % k;
                pos += first + 1;
                break;
            }
            pos += chunk;
            left -= chunk;
            cj0 = (cj0 + chunk) % k;
        }
        if (donor < 0) break;
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
    }
    if (t == 0) *nsplit_out = done == ne ? ne : -1;
}

// stats[0] = sum over the parts (ascending) of the double at obj_parts[p * stride];
// stats[1] = faiss imbalance_factor = k * sum(h^2) / (sum h)^2 (the sums are of integers < 2^53: exact in any order)
__global__ __launch_bounds__(WG) void lloyd_stats_kernel(const float* __restrict__ hassign, int k, const double* obj_parts,
                                                        long obj_stride, int n_parts, double* __restrict__ stats) {
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
}

}  // namespace

extern "C" {

int at_split_clusters_f32(at_ctx* ctx, int d, int k, int64_t n, float* hassign, float* centroids, int32_t* nsplit_out,
                          void* stream_) {
    AT_REQUIRE(ctx && hassign && centroids && nsplit_out && d > 0 && k > 0 && n >= k, "at_split_clusters_f32: bad arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AT_HIP(hipSetDevice(ctx->device));
    int* empties = static_cast<int*>(at_ws(ctx, WS_SPLIT_LIST, (size_t)k * sizeof(int), stream));
    if (!empties) return AT_E_NOMEM;
    hipLaunchKernelGGL(split_clusters_kernel, dim3(1), dim3(WG), 0, stream, d, k, (long)n, hassign, centroids, empties,
                       nsplit_out);
    AT_LAUNCH_CHECK();
    return AT_OK;
}

int at_lloyd_stats_f64(at_ctx* ctx, const float* hassign, int k, const double* obj_parts, int64_t obj_part_stride,
                       int n_parts, double* stats, void* stream_) {
    AT_REQUIRE(ctx && hassign && obj_parts && stats && k > 0 && n_parts >= 1, "at_lloyd_stats_f64: bad arguments");
    AT_REQUIRE((reinterpret_cast<uintptr_t>(obj_parts) & 7u) == 0 && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0,
               "at_lloyd_stats_f64: obj_parts / stats must be 8-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AT_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(lloyd_stats_kernel, dim3(1), dim3(WG), 0, stream, hassign, k, obj_parts, (long)obj_part_stride,
                       n_parts, stats);
    AT_LAUNCH_CHECK();
    return AT_OK;
}

}  // extern "C"
