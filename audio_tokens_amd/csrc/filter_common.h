// filter_common.h -- layout of the fp16 centroid image and the small helpers of the fp16-split filter (filter.hip).
// Not installed.
#pragma once
#include "at_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace atf {

constexpr unsigned NONE = 0xffffffffu;
constexpr float RANGE_SQ = 1073741824.0f;  // 2^30: |v| < 2^15 for every component

// Image of group g (filter.hip, prep_centroids_f16_kernel): hi fragments [s = feature/16][lane][8 halves], 1 KiB each;
// then 128 B of |c|^2 of the 32 slots and 128 B of their centroid indices (padded to 1 KiB); then the lo fragments.
__host__ __device__ constexpr size_t group_bytes(int d) { return (size_t)32 * d * 4 + 1024; }
__host__ __device__ constexpr size_t misc_off(int d) { return (size_t)(d / 16) * 1024; }   // |c|^2, +128: indices
__host__ __device__ constexpr size_t lo_off(int d) { return (size_t)(d / 16) * 1024 + 1024; }

__device__ __forceinline__ float min16(const float (&p)[16]) {
    const float a = __builtin_fminf(__builtin_fminf(p[0], p[1]), __builtin_fminf(p[2], p[3]));
    const float b = __builtin_fminf(__builtin_fminf(p[4], p[5]), __builtin_fminf(p[6], p[7]));
    const float c = __builtin_fminf(__builtin_fminf(p[8], p[9]), __builtin_fminf(p[10], p[11]));
    const float d = __builtin_fminf(__builtin_fminf(p[12], p[13]), __builtin_fminf(p[14], p[15]));
    return __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(c, d));
}

// insert (P, idx) into a sorted triple (v1 <= v2 <= v3) with the slots of the first two
__device__ __forceinline__ void insert3(float P, unsigned idx, float& v1, float& v2, float& v3, unsigned& j1, unsigned& j2) {
    const bool lt1 = P < v1, lt2 = P < v2;
    const float n3 = __builtin_amdgcn_fmed3f(v2, v3, __builtin_fmaxf(v1, P));
    j2 = lt1 ? j1 : (lt2 ? idx : j2);
    v2 = __builtin_amdgcn_fmed3f(v1, v2, P);
    j1 = lt1 ? idx : j1;
    v1 = __builtin_fminf(v1, P);
    v3 = n3;
}

// What the fused pre-pass needs (FUSED instantiation): the guesses in visiting order, the raw
// centroids, the centroid-to-group bounds, and where to leave the exact guess distances / statistics.
struct FusedPrepass {
    const uint32_t* hint_sorted;
    const float* C;
    const float* dmin;
    float* bd_out;
    float* dist_out;   // optional: the guess distance where the winner is the guess, DIST_TODO elsewhere
    unsigned long long* stats;   // (unused by the sweep since round 2: see blk_stats)
    int k;
    // guess generator over rows in their own order (frames of clips): nearest group mean first, then the
    // groups its neighbour table names -- both inside one launch
    const unsigned char* means_img;
    const uint32_t* gnbr;
    int ngm;
};
constexpr unsigned DIST_TODO = 0x7fc0deadu;  // a NaN no computed distance can be
constexpr unsigned DIST_LISTED = 0x7fc0beefu; // another one: the row is listed, the redo writes its distance
constexpr unsigned AMB_SUBLISTS = 64;      // the rows a sweep lists for the redo are appended to 64 sub-lists

}  // namespace atf
