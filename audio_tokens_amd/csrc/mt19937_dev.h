// mt19937_dev.h -- std::mt19937 on one workgroup (>= 227 threads), state in LDS.
//
// A regeneration of the 624-word state has three phases of <= 227 independent words each: word k
// needs the NEW word k-227 (computed one phase earlier) and the old words k and k+1.  With two copies
// of the state (read one, write the other) a regeneration costs three barriers.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace at_mt {

constexpr int N = 624, M = 397, D = N - M;   // D = 227

struct State {
    uint32_t st[2][N];
};

__device__ __forceinline__ uint32_t twist(uint32_t a, uint32_t b, uint32_t far) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// Called by every thread of the workgroup; the seeded state is copy 0.  Ends with a barrier.
__device__ __forceinline__ void seed(State& s, uint32_t value) {
    if (threadIdx.x == 0) {
        uint32_t x = value;
        s.st[0][0] = x;
        for (int i = 1; i < N; i++) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            s.st[0][i] = x;
        }
    }
    __syncthreads();
}

// Called by every thread: the next 624 untempered words are written to copy cur^1, which is returned
// (valid after the trailing barrier).  The caller flips `cur` afterwards.
__device__ __forceinline__ const uint32_t* regenerate(State& s, int cur) {
    const int t = threadIdx.x;
    const uint32_t* o = s.st[cur];
    uint32_t* nw = s.st[cur ^ 1];
    if (t < D) nw[t] = twist(o[t], o[t + 1], o[t + M]);                                   // k in [0, 227)
    __syncthreads();
    if (t < D) { const int k = t + D; nw[k] = twist(o[k], o[k + 1], nw[k - D]); }          // [227, 454)
    __syncthreads();
    if (t < N - 2 * D) {                                                                    // [454, 624)
        const int k = t + 2 * D;
        nw[k] = twist(o[k], (k == N - 1) ? nw[0] : o[k + 1], nw[k - D]);
    }
    __syncthreads();
    return nw;
}

}  // namespace at_mt
