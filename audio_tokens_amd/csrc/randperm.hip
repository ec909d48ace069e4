// randperm.hip -- the first m entries of faiss' rand_perm(n, seed) computed on the device.
//
// Reference call site: processors/cluster_creator.py:54,56 -> faiss.Kmeans.train ->
// Clustering::train_encoded -> subsample_training_set: perm = rand_perm(nx, seed), rows perm[0 : k*256]
// are kept in that order.  rand_perm (faiss 1.8.0 utils/random.cpp) is a Fisher-Yates shuffle driven by
// std::mt19937: for i in [0, n-1): swap(perm[i], perm[i + mt() % (n - i)]).
//
// The host form (at_rand_perm_prefix_mt19937) is a chain of 2 M dependent cache misses -- 60 to 240 ms
// on one core against a 180 ms pipeline pass.  Here the same bits come out of four short kernels:
//
//   1. mt19937_kernel: ONE workgroup owns the 624-word state in LDS (two copies).  A regeneration has three
//      phases of <= 227 independent words each (word k needs the NEW word k-227 from the phase before, and
//      old words k, k+1), so 624 tempered outputs cost three barriers.  2.1 M draws: ~1.5 ms.
//   2. the draws are turned into swap partners o_i = i + raw_i % (n - i), and (o_i, i) pairs are radix
//      sorted by o (stable, so the steps that touch one position stay in ascending order).
//   3. who-touched-what: step i reads position o_i and leaves there the value V_i that sat at position i
//      just before step i.  Hence the value step i receives is V_prev of the previous step that touched
//      o_i, or o_i itself when it is the first; and V_i = V_last of the last step j < i with o_j == i, or i.
//      Positions < m are final after their own step, so prefix[i] is exactly the value step i received
//      (V_i for a self swap).  `last` chains are followed by pointer chasing (expected length < 1.2).
//
// Same result as the sequential shuffle, bit for bit (tests/test_gpu_ops.py::test_rand_perm_device_*).
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "at_internal.h"
#include "mt19937_dev.h"

namespace {

using perm_radix_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

// raw[q] = the q-th output of std::mt19937(seed), q in [0, count).
// state_out (624 words, may be null): the generator's state after the last whole block of 624 outputs, i.e. what
// a later kernel regenerates from to continue the stream (count must then be a multiple of 624).
__global__ __launch_bounds__(256) void mt19937_kernel(uint32_t seed, int64_t count, uint32_t* __restrict__ raw,
                                                      uint32_t* __restrict__ state_out) {
    __shared__ at_mt::State s;
    at_mt::seed(s, seed);
    int cur = 0;
    for (int64_t base = 0; base < count; base += at_mt::N) {
        const uint32_t* nw = at_mt::regenerate(s, cur);
        for (int k = threadIdx.x; k < at_mt::N; k += 256)
            if (base + k < count) raw[base + k] = at_mt::temper(nw[k]);
        cur ^= 1;   // (the next regeneration writes the other copy: no barrier needed before it reads this one)
    }
    if (state_out)
        for (int k = threadIdx.x; k < at_mt::N; k += 256) state_out[k] = s.st[cur][k];
}

__device__ __forceinline__ uint32_t partner(const uint32_t* raw, int64_t i, int64_t n) {
    return (uint32_t)(i + (int64_t)(raw[i] % (uint32_t)(n - i)));
}

// keys[i] = position step i swaps with (n for a self swap: sorted behind everything, ignored), vals[i] = i
__global__ void perm_keys_kernel(const uint32_t* __restrict__ raw, int64_t steps, int64_t n, uint32_t* __restrict__ keys,
                                 uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= steps) return;
    const uint32_t o = partner(raw, i, n);
    keys[i] = (o == (uint32_t)i) ? (uint32_t)n : o;
    vals[i] = (uint32_t)i;
}

// sorted (position, step): prev[step] = the step that touched the same position just before (-1: none),
// last[position] = the last step that touched it (positions < m only; pre-set to -1)
__global__ void perm_links_kernel(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs, int64_t steps, int64_t n,
                                  int64_t m, int32_t* __restrict__ prev, int32_t* __restrict__ last) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= steps) return;
    const uint32_t key = ks[q], i = vs[q];
    if (key >= (uint32_t)n) { prev[i] = -1; return; }
    prev[i] = (q > 0 && ks[q - 1] == key) ? (int32_t)vs[q - 1] : -1;
    if ((int64_t)key < m && (q + 1 == steps || ks[q + 1] != key)) last[key] = (int32_t)i;
}

__device__ __forceinline__ int32_t value_before_own_step(const int32_t* __restrict__ last, int32_t j) {
    for (int32_t l = last[j]; l >= 0; l = last[j]) j = l;
    return j;
}

__global__ void perm_resolve_kernel(const uint32_t* __restrict__ raw, const int32_t* __restrict__ prev,
                                    const int32_t* __restrict__ last, int64_t steps, int64_t n, int64_t m,
                                    int32_t* __restrict__ prefix) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int32_t out;
    if (i >= steps) {
        out = value_before_own_step(last, (int32_t)i);   // position n-1 has no step of its own
    } else {
        const uint32_t o = partner(raw, i, n);
        if (o == (uint32_t)i) out = value_before_own_step(last, (int32_t)i);
        else { const int32_t p = prev[i]; out = p >= 0 ? value_before_own_step(last, p) : (int32_t)o; }
    }
    prefix[i] = out;
}

}  // namespace

// The first AT_MT_CACHE_DRAWS outputs of mt19937(seed), computed once per context and kept: faiss seeds BOTH the
// subsample permutation (rand_perm(n, 1234)) and every split_clusters call (RandomGenerator(1234)) with the same
// constant, so every train() and every repair replays the same stream from its start.  *state_end = the state to
// continue from.  Consumers on other streams are ordered behind the generation by an event.
int at_mt_cached_draws(at_ctx* ctx, uint32_t seed, hipStream_t stream, const uint32_t** raw, int64_t* raw_n,
                       const uint32_t** state_end) {
    constexpr int64_t n_draws = AT_MT_CACHE_DRAWS;
    uint32_t* buf = static_cast<uint32_t*>(at_ws(ctx, WS_MT_RAW, (size_t)(n_draws + at_mt::N) * 4, stream));
    if (!buf) return AT_E_NOMEM;
    if (!ctx->mt_ready) AT_HIP(hipEventCreateWithFlags(&ctx->mt_ready, hipEventDisableTiming));
    if (!ctx->mt_have || ctx->mt_seed != seed) {
        if (ctx->mt_have) AT_HIP(hipDeviceSynchronize());   // another seed's consumers may still be reading
        AT_LAUNCH(mt19937_kernel, dim3(1), dim3(256), 0, stream, seed, n_draws, buf, buf + n_draws);
        AT_HIP(hipEventRecord(ctx->mt_ready, stream));
        ctx->mt_have = 1;
        ctx->mt_seed = seed;
    } else {
        AT_HIP(hipStreamWaitEvent(stream, ctx->mt_ready, 0));
    }
    *raw = buf;
    *raw_n = n_draws;
    *state_end = buf + n_draws;
    return AT_OK;
}

extern "C" int at_rand_perm_prefix_device(at_ctx* ctx, int64_t n, int64_t seed, int64_t m, int32_t* prefix, void* stream_) {
    AT_REQUIRE(ctx != nullptr, "at_rand_perm_prefix_device: ctx is null");
    AT_REQUIRE(n >= 0 && n < INT32_MAX && m >= 0 && m <= n, "at_rand_perm_prefix_device: bad sizes n=%lld m=%lld",
               (long long)n, (long long)m);
    AT_REQUIRE(prefix != nullptr || m == 0, "at_rand_perm_prefix_device: prefix is null");
    if (m == 0) return AT_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t steps = m < n - 1 ? m : n - 1;
    const size_t sb = (size_t)(steps > 0 ? steps : 1) * 4;
    // the draws: the context's resident mt19937(1234) stream when it is that seed and long enough, else this call's own
    const uint32_t* raw = nullptr;
    uint32_t* raw_own = nullptr;
    if ((uint32_t)seed == 1234u && steps <= AT_MT_CACHE_DRAWS) {
        int64_t have = 0;
        const uint32_t* st_end = nullptr;
        int rcd = at_mt_cached_draws(ctx, 1234u, stream, &raw, &have, &st_end);
        if (rcd) return rcd;
    } else {
        raw_own = static_cast<uint32_t*>(at_ws(ctx, WS_PERM_RAW, sb, stream));
        if (!raw_own) return AT_E_NOMEM;
        raw = raw_own;
    }
    uint32_t* ka = static_cast<uint32_t*>(at_ws(ctx, WS_PERM_KEYS_A, sb, stream));
    uint32_t* kb = static_cast<uint32_t*>(at_ws(ctx, WS_PERM_KEYS_B, sb, stream));
    uint32_t* va = static_cast<uint32_t*>(at_ws(ctx, WS_PERM_VALS_A, sb, stream));
    uint32_t* vb = static_cast<uint32_t*>(at_ws(ctx, WS_PERM_VALS_B, sb, stream));
    int32_t* prev = static_cast<int32_t*>(at_ws(ctx, WS_PERM_PREV, sb, stream));
    int32_t* last = static_cast<int32_t*>(at_ws(ctx, WS_PERM_LAST, (size_t)m * 4, stream));
    if (!ka || !kb || !va || !vb || !prev || !last) return AT_E_NOMEM;
    AT_HIP(hipMemsetAsync(last, 0xFF, (size_t)m * 4, stream));
    const unsigned tb = 256;
    if (steps > 0) {
        if (raw_own) {
            AT_LAUNCH(mt19937_kernel, dim3(1), dim3(256), 0, stream, (uint32_t)seed, steps, raw_own, nullptr);
        }
        const unsigned gs = (unsigned)((steps + tb - 1) / tb);
        AT_LAUNCH(perm_keys_kernel, dim3(gs), dim3(tb), 0, stream, raw, steps, n, ka, va);
        int bits = 1;
        while (bits < 32 && (uint64_t(1) << bits) <= (uint64_t)n) bits++;
        rocprim::double_buffer<uint32_t> kbuf(ka, kb);
        rocprim::double_buffer<uint32_t> vbuf(va, vb);
        size_t tmp_bytes = 0;
        AT_HIP(rocprim::radix_sort_pairs<perm_radix_config>(nullptr, tmp_bytes, kbuf, vbuf, (size_t)steps, 0, bits, stream));
        void* tmp = at_ws(ctx, WS_PERM_TMP, tmp_bytes, stream);
        if (!tmp) return AT_E_NOMEM;
        AT_HIP(rocprim::radix_sort_pairs<perm_radix_config>(tmp, tmp_bytes, kbuf, vbuf, (size_t)steps, 0, bits, stream));
        AT_LAUNCH(perm_links_kernel, dim3(gs), dim3(tb), 0, stream, kbuf.current(), vbuf.current(), steps, n, m, prev, last);
    }
    AT_LAUNCH(perm_resolve_kernel, dim3((unsigned)((m + tb - 1) / tb)), dim3(tb), 0, stream, raw, prev, last, steps, n, m, prefix);
    return AT_OK;
}
