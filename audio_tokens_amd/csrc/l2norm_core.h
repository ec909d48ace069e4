// l2norm_core.h -- numpy's pairwise fp32 sum of squares with contraction switched off locally, so
// that it gives numpy's bits whatever -ffp-contract the including file uses.  (HIP's __fmul_rn /
// __fadd_rn are plain operators the compiler may still fuse, and __fsqrt_rn is the approximate
// native sqrt; sqrtf and '/' are the correctly rounded forms under hipcc's defaults.)
// Reference semantics: np.linalg.norm(x, axis=1) at processors/cluster_creator.py:64-66 and
// processors/spec_tokenizer.py:106-109 of danavery/audio-tokens (numpy loops_utils pairwise sum).
#pragma once
#include <hip/hip_runtime.h>

namespace l2n {

// a[0], a[stride], ... a[(n-1)*stride]
__device__ inline float pairwise_sumsq(const float* a, int n, int stride) {
#pragma clang fp contract(off)
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) {
            const float sq = a[i * stride] * a[i * stride];
            res = res + sq;
        }
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = a[j * stride] * a[j * stride];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float sq = a[(i + j) * stride] * a[(i + j) * stride];
                r[j] = r[j] + sq;
            }
        }
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) {
            const float sq = a[i * stride] * a[i * stride];
            res = res + sq;
        }
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sumsq(a, n2, stride) + pairwise_sumsq(a + (size_t)n2 * stride, n - n2, stride);
}

// The same value for 8 <= n <= 128 computed by 8 adjacent lanes (j = lane & 7 owns numpy's partial sum
// r[j]; the combining tree is numpy's, and fp32 addition is commutative, so every lane ends with numpy's
// bits).  All 8 lanes must call it together.
__device__ inline float pairwise_sumsq_8lanes(const float* a, int n, int j) {
#pragma clang fp contract(off)
    float r = a[j] * a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        const float sq = a[i + j] * a[i + j];
        r = r + sq;
    }
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    for (; i < n; i++) {
        const float sq = a[i] * a[i];
        r = r + sq;
    }
    return r;
}

// ||row|| + 1e-10 as numpy computes it for a float32 row
__device__ inline float row_denominator(const float* a, int n, int stride) {
    return __builtin_sqrtf(pairwise_sumsq(a, n, stride)) + 1e-10f;
}

// x / den, IEEE-rounded
__device__ inline float divide(float x, float den) { return x / den; }

}  // namespace l2n
