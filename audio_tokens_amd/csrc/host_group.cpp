// host_group.cpp -- spatial grouping of the centroid table for the pruned Lloyd sweep (heuristic:
// it decides how much work the exact pruned search can skip, never what it returns).  Kept in its
// own translation unit because it is compiled with -ffast-math (vectorised reductions); the
// bit-exact host pieces (mt19937 permutation, split_clusters, filterbank) live in host.cpp.
#include <algorithm>
#include <cmath>
#include <vector>

#include "at_internal.h"

extern "C" {

// Balanced spatial grouping of the centroid table (performance aid of the pruned Lloyd sweep; it
// never affects results): recursively cut the rows at the median of their projection onto the
// part's principal axis (a few power iterations) until a part holds at most `leaf` rows.
// perm_out receives ceil(k/leaf)*leaf entries: the rows group after group, each group padded to
// `leaf` entries with -1.
int at_group_rows_kd_host(const float* rows, int k, int d, int leaf, int32_t* perm_out) {
    AT_REQUIRE(rows && perm_out && k > 0 && d > 0 && leaf > 0, "at_group_rows_kd_host: bad arguments");
    const int ngroups = (k + leaf - 1) / leaf;
    std::vector<int32_t> idx((size_t)k);
    for (int i = 0; i < k; i++) idx[i] = i;
    struct Part { int lo, hi, g0, g1; };
    std::vector<Part> stack{{0, k, 0, ngroups}};
    std::vector<int> bounds((size_t)ngroups + 1, 0);
    bounds[ngroups] = k;
    std::vector<float> meanf((size_t)d), axisf((size_t)d), nextf((size_t)d), proj((size_t)k), cen;
    while (!stack.empty()) {
        const Part p = stack.back();
        stack.pop_back();
        if (p.g1 - p.g0 <= 1) {
            bounds[p.g0] = p.lo;
            continue;
        }
        const int m = p.hi - p.lo;
        // centred copy of the part (contiguous, float: the loops below vectorise)
        cen.resize((size_t)m * d);
        std::fill(meanf.begin(), meanf.end(), 0.0f);
        for (int i = 0; i < m; i++) {
            const float* r = rows + (size_t)idx[p.lo + i] * d;
            for (int f = 0; f < d; f++) meanf[f] += r[f];
        }
        for (int f = 0; f < d; f++) meanf[f] /= (float)m;
        for (int i = 0; i < m; i++) {
            const float* r = rows + (size_t)idx[p.lo + i] * d;
            float* c = cen.data() + (size_t)i * d;
            for (int f = 0; f < d; f++) c[f] = r[f] - meanf[f];
        }
        // principal axis by power iteration (deterministic start)
        for (int f = 0; f < d; f++) axisf[f] = 1.0f + 0.01f * (float)f;
        for (int it = 0; it < 4; it++) {
            std::fill(nextf.begin(), nextf.end(), 0.0f);
            for (int i = 0; i < m; i++) {
                const float* c = cen.data() + (size_t)i * d;
                float t = 0.0f;
                for (int f = 0; f < d; f++) t += c[f] * axisf[f];
                for (int f = 0; f < d; f++) nextf[f] += t * c[f];
            }
            float nrm = 0.0f;
            for (int f = 0; f < d; f++) nrm += nextf[f] * nextf[f];
            if (!(nrm > 0.0f)) break;
            nrm = 1.0f / std::sqrt(nrm);
            for (int f = 0; f < d; f++) axisf[f] = nextf[f] * nrm;
        }
        for (int i = 0; i < m; i++) {
            const float* c = cen.data() + (size_t)i * d;
            float t = 0.0f;
            for (int f = 0; f < d; f++) t += c[f] * axisf[f];
            proj[idx[p.lo + i]] = t;
        }
        const int gmid = p.g0 + (p.g1 - p.g0) / 2;
        int cut = p.lo + (gmid - p.g0) * leaf;  // left side = whole groups
        if (cut > p.hi) cut = p.hi;
        std::nth_element(idx.begin() + p.lo, idx.begin() + cut, idx.begin() + p.hi, [&](int32_t a, int32_t b) {
            return proj[a] < proj[b] || (proj[a] == proj[b] && a < b);
        });
        stack.push_back({p.lo, cut, p.g0, gmid});
        stack.push_back({cut, p.hi, gmid, p.g1});
    }
    for (int g = 0; g < ngroups; g++) {
        const int lo = bounds[g], hi = bounds[g + 1];
        for (int e = 0; e < leaf; e++) perm_out[(size_t)g * leaf + e] = lo + e < hi ? idx[lo + e] : -1;
    }
    return AT_OK;
}


}  // extern "C"
