// filter.hip -- fp16-split filter for the exact pruned sweep (used by at_assign_pruned_f32).
//
// Accelerates faiss.IndexFlatL2.search(x, 1) at processors/spec_tokenizer.py:77 and inside
// faiss.Kmeans.train (processors/cluster_creator.py:54-56 of danavery/audio-tokens) without
// changing an output bit.
//
// The fp32 MFMA runs at 1/16 of the fp16 MFMA rate on gfx950.  Every fp32 value v is split as
// v = hi + lo + r with hi = fp16(v), lo = fp16(v - hi); then
//     x.c  ~=  hi_x.hi_c + hi_x.lo_c + lo_x.hi_c          (three v_mfma_f32_32x32x16_f16, fp32 accumulate)
// with an error that is bounded a priori (below).  The sweep keeps, per row, the smallest and the
// second smallest approximate distance P(j) = |c_j|^2 - 2 ip16(x, c_j) over the candidates the group
// masks admit.  With eps bounding |P(j) + |x|^2 - T(j)| (T = true squared distance) and delta bounding
// |D(j) - T(j)| for the distance D the fp32 contract computes (prune.hip), the contract's arg-min
// j* satisfies D(j*) <= D(j) for all j, hence P(j*) <= P(j) + 2 delta + 2 eps =: P(j) + tau.
// So when the runner-up is more than tau above the best, the best IS j* -- whatever the tie rule --
// and only its distance remains to be evaluated with the contract's fmaf chain (at_exact_dist_rows).
// Rows that fail the test (about 2-3 % on log-mel frames), rows or centroids outside the fp16 range
// and rows with non-finite values are listed and redone by the fp32 sweep.
//
// Error budget (s = |x|, t = max |c|, H = s^2 + t^2 >= 2 s t, u = 2^-24, q = sqrt(d) 2^-25,
// m = 3 d / 16 MFMAs); errors of the inner product count twice in P = |c|^2 - 2 ip:
//   representation   |x.c - (three kept terms)| <= 3*2^-22 s t + 2 q (s + t)
//                    (fp16 rounding 2^-11 relative, 2^-25 absolute below the normal range)
//   accumulation     each MFMA adds 16 exact products to an fp32 accumulator; charged 17 roundings
//                    of one ulp (2u) of the running magnitude <= 1.01 s t each:  34 m u 1.01 s t
//   |c|^2, |x|^2     fmaf chains: d u 1.01 each, relative
//   forming P        one fma: u (t^2 + 2 s t) <= 2 u H
//   eps <= H (34 m u 1.01 + 3*2^-22 + d u 1.01 + 2 u + 2 q) + 4 q      (s t <= H/2, s + t <= 1 + H/2)
//   tau  = 2 (2d + 8) u 1.01 H + 2 eps
// tests/test_gpu_ops.py::test_filter_error_bound measures the actual error against float64 (it is
// some fifty times smaller).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "filter_common.h"

namespace {
using namespace atf;

constexpr int WG = 256;

// Image of group g, hot part first: hi fragments [s = feature/16][lane][8 halves] (lane = 32 * ((f % 16)
// / 8) + slot, exactly the A operand of v_mfma_f32_32x32x16_f16, 1 KiB per fragment), then 256 bytes of
// |c|^2 of the 32 slots (+inf for padding) and their centroid indices (padded to 1 KiB), then the lo
// fragments in the same form.  The sweep streams the first d/16 KiB + 256 B of every group it
// visits and touches the lo fragments of a few.
__global__ void __launch_bounds__(WG) prep_centroids_f16_kernel(const float* __restrict__ c, int k, int d,
                                                                const int32_t* __restrict__ cperm,
                                                                unsigned char* __restrict__ img,
                                                                unsigned* __restrict__ max_bits) {
    const int g = blockIdx.x;
    unsigned char* out = img + (size_t)g * group_bytes(d);
    _Float16* frag = reinterpret_cast<_Float16*>(out);
    for (int e = threadIdx.x; e < 32 * d; e += WG) {
        const int i = e / d, f = e - i * d;
        const int row = cperm[g * 32 + i];
        const float v = (row >= 0 && row < k) ? c[(size_t)row * d + f] : 0.0f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int s = f >> 4, lane = 32 * ((f & 15) >> 3) + i, el = f & 7;
        frag[((size_t)s * 64 + lane) * 8 + el] = hi;
        reinterpret_cast<_Float16*>(out + lo_off(d))[((size_t)s * 64 + lane) * 8 + el] = lo;
    }
    float* cn = reinterpret_cast<float*>(out + misc_off(d));
    unsigned* idx = reinterpret_cast<unsigned*>(out + misc_off(d) + 128);
    for (int i = threadIdx.x; i < 32; i += WG) {
        const int row = cperm[g * 32 + i];
        float nrm = __builtin_inff();
        if (row >= 0 && row < k) {
            nrm = 0.0f;
            for (int f = 0; f < d; f++) {
                const float v = c[(size_t)row * d + f];
                nrm = __builtin_fmaf(v, v, nrm);
            }
        }
        cn[i] = nrm;
        idx[i] = (row >= 0 && row < k) ? (unsigned)row : NONE;
        // max |c|^2 over the table (every row is in exactly one group): non-negative floats (and +inf, NaN
        // above them) order like their bit patterns
        if (max_bits && row >= 0 && row < k) atomicMax(max_bits, __float_as_uint(nrm) & 0x7fffffffu);
    }
}

// One wavefront per workgroup, the structure of assign_mfma_pruned_reg_kernel (assign.hip): the wave
// walks the groups its 32*NB rows (NB = 4 tiles of 32, handled in pairs) need; the hi fragments of the next group are
// fetched into registers while the current one multiplies.  The needed groups are compacted once
// into a small LDS list (group | tile bits), so the walk costs a handful of scalar instructions per
// group.  misc[0] = max |c|^2 bits, misc[1] = ambiguous-row counter.
//
// The hi*hi product alone is within rho of the three-term value: a tile in which no row's hi*hi
// distance comes within rho of that row's runner-up cannot change anybody's (best, runner-up)
// pair, so its two lo products (8 of the 12 MFMAs) are skipped and the lo fragments of the group
// are fetched only when some tile asks for them.
template <int D, int NB, bool GUESS, bool FUSED, int WPS = 2>
__global__ void __launch_bounds__(64, WPS)
assign_f16filter_kernel(const float* __restrict__ X, long n, const unsigned char* __restrict__ img, int ng,
                        const uint32_t* __restrict__ order, const float* __restrict__ bd_in,
                        const uint32_t* __restrict__ mask, int ngw, unsigned* __restrict__ misc, float tau_a,
                        float tau_b, float rho_a, float rho_b, int screen, int collect, long* __restrict__ ids,
                        uint32_t* __restrict__ amb_list, uint32_t* __restrict__ amb_aux,
                        float* __restrict__ approx_out, FusedPrepass fp, uint4* __restrict__ blk_stats, unsigned amb_cap) {
    static_assert(NB == 1 || NB == 2 || NB == 4, "tiles are processed in pairs (NB = 1: a pair whose second tile does not exist)");
    // blk_stats (may be null = statistics off): one record per workgroup {groups needed, groups of the dense sweep,
    // tiles multiplied hi*hi, tiles refined}, summed by filter_stats_reduce_kernel.  Round 1 added these with
    // atomics on 16 + 256 hot words from all 32 768 workgroups of a Lloyd sweep: 40 us of a 600 us kernel.
    unsigned st_needed = 0, st_total = 0;
    constexpr int NS = D / 16;
    constexpr size_t GB = group_bytes(D);
    __shared__ unsigned short glist[512];
    __shared__ uint32_t maskl[NB][16];   // (fused coarse mode) the tiles' group masks
    // per-row values that only the epilogue needs sit out the walk in LDS (the walk is short of registers)
    constexpr bool STASH = NB <= 2;      // (four tiles already fill the CU's LDS with the lo parts of the rows)
    __shared__ float stash_tau[STASH ? NB : 1][64], stash_gbd[STASH ? NB : 1][64];
    __shared__ unsigned stash_row[STASH ? NB : 1][64], stash_hint[STASH ? NB : 1][64];
    const unsigned char* img_cur = img;  // the image the walk reads: the centroids', or first the group means'
    bool means_walk = false;             // (guess generator) the walk under way is the one over the group means
    __shared__ half8 xl_lds[NB][D / 16][64];  // lo parts of the rows: only the rare refined tiles read them

    const int lane = threadIdx.x;
    const int j = lane & 31;
    const int h = lane >> 5;
    const long pos0 = (long)blockIdx.x * (32 * NB);
    const long ntile32 = (n + 31) / 32;
    const float cnmax = __uint_as_float(misc[0]);
    const bool c_bad = !(cnmax < RANGE_SQ);

    half8 xh[NB][NS];
    // per row (lane-local; the two half-waves hold disjoint centroids of the same row): the three
    // smallest approximate distances seen and the slots of the two smallest
    float b1[NB], b2[NB], b3[NB], tau[NB], rho[NB];
    unsigned i1[NB], i2[NB];
    bool bad[NB];
    unsigned rowid[NB];
    float mtau[FUSED ? NB : 1];      // fused pre-pass: the row's Elkan radius (2R) / its guess / its distance
    float gbd[(FUSED || GUESS) ? NB : 1];   // (guess generator: |x|^2, to turn P into an approximate distance)
    uint32_t hintp[FUSED ? NB : 1];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        long pos = pos0 + 32 * b + j;
        if (pos >= n) pos = n - 1;
        const unsigned r = order ? order[pos] : (unsigned)pos;
        rowid[b] = r;
        const f32x4* p = reinterpret_cast<const f32x4*>(X + (size_t)r * D);
        float part = 0.0f;
        f32x4 xu[NS], xv[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            xu[s] = p[4 * s + 2 * h];
            xv[s] = p[4 * s + 2 * h + 1];
        }
        float bd = __builtin_inff();
        float nrm;
        if constexpr (FUSED) {
            // The pre-pass in place (prune.hip prune_mask_kernel): the exact distance to the guess is the
            // contract's three ascending fmaf chains.  The row's features sit in 8-wide chunks on lanes
            // j (even chunks) and j+32 (odd chunks), so the chain state hops between the two lanes:
            // both compute every step on their own chunk, the owner's result is the one that is kept.
            const uint32_t g = fp.hint_sorted[pos];
            const bool has = g < (uint32_t)fp.k;
            hintp[b] = has ? g : NONE;
            const f32x4* pc = reinterpret_cast<const f32x4*>(fp.C + (size_t)(has ? g : 0u) * D);
            f32x4 cu[NS], cv[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                cu[s] = pc[4 * s + 2 * h];
                cv[s] = pc[4 * s + 2 * h + 1];
            }
            float xn = 0.0f, cn = 0.0f, ip = 0.0f;
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
                for (int owner = 0; owner < 2; owner++) {
                    float a = xn, c2 = cn, d2 = ip;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        a = __builtin_fmaf(xu[s][e], xu[s][e], a);
                        c2 = __builtin_fmaf(cu[s][e], cu[s][e], c2);
                        d2 = __builtin_fmaf(cu[s][e], xu[s][e], d2);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        a = __builtin_fmaf(xv[s][e], xv[s][e], a);
                        c2 = __builtin_fmaf(cv[s][e], cv[s][e], c2);
                        d2 = __builtin_fmaf(cv[s][e], xv[s][e], d2);
                    }
                    const float oa = __shfl_xor(a, 32), oc = __shfl_xor(c2, 32), od = __shfl_xor(d2, 32);
                    const bool mine = h == owner;
                    xn = mine ? a : oa;
                    cn = mine ? c2 : oc;
                    ip = mine ? d2 : od;
                }
            }
            nrm = xn;
            // exactly the value the fp32 sweep would produce for (x, c_p)
            const float dh = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);
            bd = has ? dh : __builtin_inff();
            gbd[b] = bd;
            if (h == 0 && pos0 + 32 * b + j < n) fp.bd_out[pos] = bd;
            const float delta = (2.0f * D + 8.0f) * 5.9604645e-8f * (xn + cnmax) * 1.01f;
            // rows without a guess need every group; positions past n need none
            mtau[b] = pos0 + 32 * b + j >= n ? -1.0f
                                             : (has ? 2.0f * sqrtf(dh + delta) * (1.0f + 4.0f * 5.9604645e-8f) + 1e-30f
                                                    : __builtin_inff());
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const f32x4 u = xu[s], v = xv[s];
            half8 xlo;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                part = __builtin_fmaf(u[e], u[e], part);
                const _Float16 hu = (_Float16)u[e];
                xh[b][s][e] = hu;
                xlo[e] = (_Float16)(u[e] - (float)hu);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                part = __builtin_fmaf(v[e], v[e], part);
                const _Float16 hv = (_Float16)v[e];
                xh[b][s][4 + e] = hv;
                xlo[4 + e] = (_Float16)(v[e] - (float)hv);
            }
            xl_lds[b][s][lane] = xlo;
        }
        if constexpr (!FUSED) {
            nrm = part + __shfl_xor(part, 32);
            bd = bd_in ? bd_in[pos] : __builtin_inff();
            if constexpr (GUESS) gbd[b] = nrm;
        }
        tau[b] = __builtin_fmaf(tau_a, nrm * 1.001f + cnmax, tau_b);
        rho[b] = screen ? __builtin_fmaf(rho_a, nrm * 1.001f + cnmax, rho_b) : __builtin_inff();
        bad[b] = c_bad || !(nrm < RANGE_SQ);
        // candidates above the cap can neither be the arg-min nor within tau of it: the guess itself
        // (always admitted by the masks) has P <= bd - |x|^2 + eps
        const float cap = bd < __builtin_inff() ? (bd - nrm) + 3.0f * tau[b] : __builtin_inff();
        b1[b] = cap;
        b2[b] = cap;
        b3[b] = cap;
        i1[b] = NONE;
        i2[b] = NONE;
    }

    // needed groups of this wave, compacted: entry = group | (tile bits << 9)
    int cnt = 0;
    if constexpr (FUSED) {
        // per tile (its 32 rows are on lanes 0..31 and again on 32..63): a group is needed iff
        // dmin[p][g] <= the largest radius of some run of equal guesses p (prune_mask_kernel's test,
        // lanes standing for groups, one coalesced read of the run's dmin row per 64 groups)
        unsigned long long need[NB][8];
        int needed_total = 0;
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int it = 0; it < 8; it++) need[b][it] = 0ull;
            const bool live = mtau[b] >= 0.0f;            // position < n
            const bool nohint = live && hintp[b] == NONE;
            if (__builtin_amdgcn_ballot_w64(nohint) != 0) {
#pragma unroll
                for (int it = 0; it < 8; it++) need[b][it] = ~0ull;
            } else {
                unsigned long long todo = __builtin_amdgcn_ballot_w64(live) & 0xffffffffull;
                while (todo != 0) {
                    const int leader = __builtin_ctzll(todo);
                    const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)hintp[b], leader);
                    const bool in_run = live && hintp[b] == pl;
                    todo &= ~__builtin_amdgcn_ballot_w64(in_run);
                    float t = in_run ? mtau[b] : -1.0f;
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) t = __builtin_fmaxf(t, __shfl_xor(t, off));
                    const float* drow = fp.dmin + (size_t)pl * ng;
#pragma unroll
                    for (int it = 0; it < 8; it++) {
                        const int g = 64 * it + lane;
                        if (64 * it < ng) need[b][it] |= __builtin_amdgcn_ballot_w64(g < ng && drow[g] <= t);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 8; it++) {
                if (64 * it >= ng) break;
                if (ng - 64 * it < 64) need[b][it] &= (1ull << (ng - 64 * it)) - 1ull;
                if (pos0 + 32 * b < n) needed_total += __builtin_popcountll(need[b][it]);
                else need[b][it] = 0ull;
            }
        }
        {
            int tiles_here = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) tiles_here += pos0 + 32 * b < n;
            st_needed = (unsigned)needed_total;
            st_total = (unsigned)(ng * tiles_here);
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            if (64 * it >= ng) break;
            const int g = 64 * it + lane;
            unsigned f = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) f |= (unsigned)((need[b][it] >> lane) & 1ull) << b;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(f != 0);
            if (f != 0) glist[cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(g | (f << 9));
            cnt += __builtin_popcountll(bal);
        }
    } else {
        bool two_level = false;
        if constexpr (GUESS) two_level = fp.means_img != nullptr;
        if (two_level) {  // first walk: every group of the means image, every tile that exists
            for (int base = 0; base < fp.ngm; base += 64) {
                const int g = base + lane;
                unsigned f = 0;
                if (g < fp.ngm) {
#pragma unroll
                    for (int b = 0; b < NB; b++) f |= (unsigned)(pos0 + 32 * b < n) << b;
                }
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(f != 0);
                if (f != 0) glist[cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(g | (f << 9));
                cnt += __builtin_popcountll(bal);
            }
            img_cur = fp.means_img;
            means_walk = true;
        } else {
            const long tile0 = pos0 / 32;
            for (int base = 0; base < ng; base += 64) {
                const int g = base + lane;
                unsigned f = 0;
                if (g < ng) {
#pragma unroll
                    for (int b = 0; b < NB; b++)
                        if (tile0 + b < ntile32) f |= ((mask[(size_t)(tile0 + b) * ngw + (g >> 5)] >> (g & 31)) & 1u) << b;
                }
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(f != 0);
                if (f != 0) glist[cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(g | (f << 9));
                cnt += __builtin_popcountll(bal);
            }
        }
    }
    auto entry = [&](int i) { return __builtin_amdgcn_readfirstlane((int)glist[i]); };

    auto load_group = [&](int g, half8 (&ah)[NS], f32x4 (&cn)[4]) {
        const unsigned char* base = img_cur + (size_t)g * GB;
        const half8* fr = reinterpret_cast<const half8*>(base);
#pragma unroll
        for (int s = 0; s < NS; s++) ah[s] = fr[s * 64 + lane];
        const float* cnp = reinterpret_cast<const float*>(base + misc_off(D));
#pragma unroll
        for (int q = 0; q < 4; q++) cn[q] = *reinterpret_cast<const f32x4*>(cnp + 8 * q + 4 * h);
    };
    auto screen_min = [&](const f32x16& a, const f32x4 (&cnv)[4]) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; r++) p[r] = __builtin_fmaf(-2.0f, a[r], cnv[r >> 2][r & 3]);
        return min16(p);
    };
    auto refine = [&](int b, int g, f32x16 a, const half8 (&ah)[NS], const half8 (&al)[NS], const f32x4 (&cnv)[4]) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], xh[b][s], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xl_lds[b][s][lane], a, 0, 0, 0);
        }
        float P[16];
#pragma unroll
        for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cnv[r >> 2][r & 3]);
        const float m = min16(P);
        if (__builtin_amdgcn_ballot_w64(m < b3[b]) != 0) {
            float v1 = b1[b], v2 = b2[b], v3 = b3[b];
            unsigned j1 = i1[b], j2 = i2[b];
            const unsigned base = (unsigned)g * 32u + 4u * h;
#pragma unroll
            for (int r = 0; r < 16; r++) insert3(P[r], base + (unsigned)((r & 3) + 8 * (r >> 2)), v1, v2, v3, j1, j2);
            b1[b] = v1;
            b2[b] = v2;
            b3[b] = v3;
            i1[b] = j1;
            i2[b] = j2;
        }
    };
    auto best_only = [&](int b, int g, const f32x16& a, const f32x4 (&cnv)[4]) {
        float P[16];
#pragma unroll
        for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cnv[r >> 2][r & 3]);
        if (__builtin_amdgcn_ballot_w64(min16(P) < b1[b]) != 0) {
            float v1 = b1[b];
            unsigned lr = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                lr = P[r] < v1 ? (unsigned)r : lr;
                v1 = __builtin_fminf(v1, P[r]);
            }
            if (v1 < b1[b]) i1[b] = (unsigned)g * 32u + 4u * h + (lr & 3u) + 8u * (lr >> 2);
            b1[b] = v1;
        }
    };
    unsigned n_hh = 0, n_ref = 0;  // statistics: tiles multiplied (hi*hi) / refined with the lo products
    auto compute_group = [&](int e, const half8 (&ah)[NS], const f32x4 (&cnv)[4]) {
        const int g = e & 511;
        half8 al[NS];
        bool have_al = false;
#pragma unroll
        for (int pr = 0; pr < (NB + 1) / 2; pr++) {  // tiles in pairs: two independent accumulator chains, interleaved
            const int t0 = 2 * pr;
            const bool second = 2 * pr + 1 < NB;             // (NB = 1: the pair's second tile does not exist)
            const int t1 = second ? 2 * pr + 1 : t0;
            const bool need0 = (e >> (9 + t0)) & 1, need1 = second && ((e >> (9 + 2 * pr + 1)) & 1);  // wave-uniform
            if (!(need0 || need1)) continue;
            n_hh += (unsigned)need0 + (unsigned)need1;
            f32x16 a0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            f32x16 a1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            bool pass0 = false, pass1 = false;
            if constexpr (GUESS) {
                // guess generator (coarse passes): nobody has a running best worth screening against, so
                // every tile gets all three products (a guess made from hi*hi alone is measurably worse and
                // costs the exact pass more than it saves here), but only the best candidate is kept
                // (the walk over the group MEANS only picks which groups to search: hi*hi is all it needs)
                if (!means_walk) {
                    if (!have_al) {
                        const half8* fr = reinterpret_cast<const half8*>(img_cur + (size_t)g * GB + lo_off(D));
#pragma unroll
                        for (int s = 0; s < NS; s++) al[s] = fr[s * 64 + lane];
                        have_al = true;
                    }
                    n_ref += (unsigned)need0 + (unsigned)need1;
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        if (need0) {
                            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], xh[t0][s], a0, 0, 0, 0);
                            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xl_lds[t0][s][lane], a0, 0, 0, 0);
                        }
                        if (need1) {
                            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], xh[t1][s], a1, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xl_lds[t1][s][lane], a1, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    if (need0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[t0][s], a0, 0, 0, 0);
                    if (need1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[t1][s], a1, 0, 0, 0);
                }
                if (need0) best_only(t0, g, a0, cnv);
                if (need1) best_only(t1, g, a1, cnv);
                continue;
            }
            if (need0 && need1) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[t0][s], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[t1][s], a1, 0, 0, 0);
                }
                const float m0 = screen_min(a0, cnv), m1 = screen_min(a1, cnv);
                pass0 = __builtin_amdgcn_ballot_w64(m0 < b3[t0] + rho[t0]) != 0;
                pass1 = __builtin_amdgcn_ballot_w64(m1 < b3[t1] + rho[t1]) != 0;
            } else if (need0) {
#pragma unroll
                for (int s = 0; s < NS; s++) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[t0][s], a0, 0, 0, 0);
                pass0 = __builtin_amdgcn_ballot_w64(screen_min(a0, cnv) < b3[t0] + rho[t0]) != 0;
            } else {
#pragma unroll
                for (int s = 0; s < NS; s++) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[t1][s], a1, 0, 0, 0);
                pass1 = __builtin_amdgcn_ballot_w64(screen_min(a1, cnv) < b3[t1] + rho[t1]) != 0;
            }
            if (!(pass0 || pass1)) continue;
            if (!have_al) {
                const half8* fr = reinterpret_cast<const half8*>(img_cur + (size_t)g * GB + lo_off(D));
#pragma unroll
                for (int s = 0; s < NS; s++) al[s] = fr[s * 64 + lane];
                have_al = true;
            }
            n_ref += (unsigned)pass0 + (unsigned)pass1;
            if (pass0) refine(t0, g, a0, ah, al, cnv);
            if (pass1) refine(t1, g, a1, ah, al, cnv);
        }
    };

    if constexpr (STASH) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            stash_tau[b][lane] = bad[b] ? -1.0f : tau[b];          // (a negative threshold stands for "not sane")
            stash_row[b][lane] = rowid[b];
            if constexpr (FUSED || GUESS) stash_gbd[b][lane] = gbd[b];
            if constexpr (FUSED) stash_hint[b][lane] = hintp[b];
        }
    }
    auto run_sweep = [&]() {
        // Two register sets alternate; the list entries are read from LDS two groups before they are
        // needed (as a vector register, turned into a scalar only when used), so neither the LDS round trip
        // nor the fragment loads it addresses sit in front of a group's MFMAs.
        if constexpr (WPS >= 3 || D == 128) {
            // three waves per SIMD hide the fragment loads instead of a second register set (d = 64); at
            // d = 128 a second set of 8 KiB fragments does not fit the register file beside the rows
            half8 ah[NS];
            f32x4 cn[4];
            for (int i = 0; i < cnt; i++) {
                const int e = entry(i);
                load_group(e & 511, ah, cn);
                compute_group(e, ah, cn);
            }
            return;
        }
        half8 ahA[NS], ahB[NS];
        f32x4 cnA[4], cnB[4];
        int e0 = -1, e1 = -1;
        if (cnt > 0) {
            e0 = entry(0);
            load_group(e0 & 511, ahA, cnA);
        }
        if (cnt > 1) e1 = entry(1);
        unsigned raw2 = cnt > 2 ? (unsigned)glist[2] : 0u;
        for (int i = 0; i < cnt; i += 2) {
            if (e1 >= 0) load_group(e1 & 511, ahB, cnB);
            const unsigned raw3 = i + 3 < cnt ? (unsigned)glist[i + 3] : 0u;
            compute_group(e0, ahA, cnA);
            if (e1 < 0) break;
            const int e0n = i + 2 < cnt ? __builtin_amdgcn_readfirstlane((int)raw2) : -1;
            if (e0n >= 0) load_group(e0n & 511, ahA, cnA);
            const unsigned raw4 = i + 4 < cnt ? (unsigned)glist[i + 4] : 0u;
            compute_group(e1, ahB, cnB);
            e0 = e0n;
            e1 = i + 3 < cnt ? __builtin_amdgcn_readfirstlane((int)raw3) : -1;
            raw2 = raw4;
        }
    };
    run_sweep();
    if constexpr (GUESS) {
        if (fp.means_img != nullptr) {
            // second walk: per tile the groups the neighbour table names for its rows' nearest means
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const float o1 = __shfl_xor(b1[b], 32);
                const unsigned oi = (unsigned)__shfl_xor((int)i1[b], 32);
                const unsigned slot = o1 < b1[b] ? oi : i1[b];
                const unsigned gm = slot == NONE ? NONE
                    : reinterpret_cast<const unsigned*>(img_cur + (size_t)(slot >> 5) * GB + misc_off(D) + 128)[slot & 31];
                const bool live = pos0 + 32 * b + j < n && gm != NONE && gm < (unsigned)ng;
                for (int w = 0; w < ngw; w++) {
                    uint32_t v = live ? fp.gnbr[(size_t)gm * ngw + w] : 0u;
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) v |= (uint32_t)__shfl_xor((int)v, off);
                    if (lane == 0) maskl[b][w] = v;
                }
                b1[b] = b2[b] = b3[b] = __builtin_inff();
                i1[b] = i2[b] = NONE;
            }
            img_cur = img;
            means_walk = false;
            cnt = 0;
            for (int base = 0; base < ng; base += 64) {
                const int g = base + lane;
                unsigned f = 0;
                if (g < ng) {
#pragma unroll
                    for (int b = 0; b < NB; b++) f |= ((maskl[b][g >> 5] >> (g & 31)) & 1u) << b;
                }
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(f != 0);
                if (f != 0) glist[cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(g | (f << 9));
                cnt += __builtin_popcountll(bal);
            }
            run_sweep();
        }
    }

    if (lane == 0 && blk_stats) blk_stats[blockIdx.x] = make_uint4(st_needed, st_total, n_hh, n_ref);
    auto slot_id = [&](unsigned slot) {
        return slot == NONE ? NONE
                            : reinterpret_cast<const unsigned*>(img_cur + (size_t)(slot >> 5) * GB + misc_off(D) + 128)[slot & 31];
    };
#pragma unroll
    for (int b = 0; b < NB; b++) {
        // merge the other half-wave's triple into this one
        const float o1 = __shfl_xor(b1[b], 32), o2 = __shfl_xor(b2[b], 32), o3 = __shfl_xor(b3[b], 32);
        const unsigned oj1 = (unsigned)__shfl_xor((int)i1[b], 32), oj2 = (unsigned)__shfl_xor((int)i2[b], 32);
        float n1 = b1[b], n2 = b2[b], n3 = b3[b];
        unsigned nj1 = i1[b], nj2 = i2[b];
        insert3(o1, oj1, n1, n2, n3, nj1, nj2);
        insert3(o2, oj2, n1, n2, n3, nj1, nj2);
        insert3(o3, NONE, n1, n2, n3, nj1, nj2);
        const long pos = pos0 + 32 * b + j;
        const bool mine = h == 0 && pos < n;
        float tau_b, gbd_b = 0.0f;
        unsigned row_b, hint_b = NONE;
        if constexpr (STASH) {
            tau_b = stash_tau[b][lane];
            row_b = stash_row[b][lane];
            if constexpr (FUSED || GUESS) gbd_b = stash_gbd[b][lane];
            if constexpr (FUSED) hint_b = stash_hint[b][lane];
        } else {
            tau_b = bad[b] ? -1.0f : tau[b];
            row_b = rowid[b];
            if constexpr (FUSED || GUESS) gbd_b = gbd[b];
            if constexpr (FUSED) hint_b = hintp[b];
        }
        const bool sane = nj1 != NONE && tau_b >= 0.0f && n1 > -__builtin_inff();
        const bool unique = sane && (n2 - n1) > tau_b;
        // exactly two candidates within reach: the redo only has to score those two
        const bool pair = sane && nj2 != NONE && (n3 - n1) > tau_b;
        if (mine) {
            const unsigned id = slot_id(nj1);
            ids[row_b] = id == NONE ? -1L : (long)id;
            if constexpr (FUSED) {
                // the guess distance where the winner is the guess; DIST_TODO where the distance pass has to evaluate it;
                // DIST_LISTED where the redo will (it always writes the distance of a listed row) -- the two passes then
                // touch disjoint rows and run as one launch (at_filter_finish)
                if (fp.dist_out)
                    fp.dist_out[row_b] = (collect && !unique) ? __uint_as_float(DIST_LISTED)
                                         : (id != NONE && id == hint_b) ? gbd_b : __uint_as_float(DIST_TODO);
            }
            if constexpr (GUESS) {  // a guess comes with an approximate distance (it only orders the next visit)
                if (fp.dist_out) fp.dist_out[row_b] = id != NONE ? __builtin_fmaxf(n1 + gbd_b, 0.0f) : __builtin_inff();
            }
            if (approx_out) {  // test hook: approximate distance of the winner and the gap to the runner-up
                approx_out[2 * (size_t)row_b] = n1;
                approx_out[2 * (size_t)row_b + 1] = n2 - n1;
            }
        }
        if (collect) {
            const unsigned long long flagged = __builtin_amdgcn_ballot_w64(mine && !unique);
            if (flagged != 0) {
                // 64 sub-lists (workgroup b appends to list b % 64, each with its own counter misc[64 + s] and amb_cap
                // slots): one counter for all 23 000 appending workgroups of a Lloyd sweep was a 30 us queue
                const unsigned sub = blockIdx.x & (AMB_SUBLISTS - 1);
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&misc[64 + sub], (unsigned)__builtin_popcountll(flagged));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + sub * amb_cap;
                if (mine && !unique) {
                    const unsigned rank = (unsigned)__builtin_popcountll(flagged & ((1ull << lane) - 1ull));
                    amb_list[base + rank] = (uint32_t)pos;
                    amb_aux[base + rank] = pair ? slot_id(nj2) : NONE;
                }
            }
        }
    }
}

// dist[i] = the contract's distance between row i and centroid ids[i] (same fmaf chains as the fp32
// sweep: |x|^2, |c|^2 and the inner product in ascending feature order, (xn + cn) - 2 ip, clamped).
template <int D>
__global__ void __launch_bounds__(WG) exact_dist_rows_kernel(const float* __restrict__ X, long n,
                                                             const float* __restrict__ C, int k,
                                                             const long* __restrict__ ids,
                                                             float* __restrict__ dist) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const long p = ids[i];
    if (p < 0 || p >= k) {
        dist[i] = __builtin_inff();
        return;
    }
    const f32x4* px = reinterpret_cast<const f32x4*>(X + i * D);
    const f32x4* pc = reinterpret_cast<const f32x4*>(C + p * D);
    f32x4 xv[D / 4], cv[D / 4];  // all loads in flight before the first dependent fma
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
    dist[i] = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);
}

// Same, in visiting order and reusing the pre-pass: where the winner is the guess, its distance is
// the bd the pre-pass already evaluated with the very same chain.
template <int D>
__global__ void __launch_bounds__(WG) exact_dist_visit_kernel(const float* __restrict__ X, long n,
                                                              const float* __restrict__ C, int k,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ hint_sorted,
                                                              const float* __restrict__ bd,
                                                              const long* __restrict__ ids,
                                                              float* __restrict__ dist) {
    const long pos = (long)blockIdx.x * WG + threadIdx.x;
    if (pos >= n) return;
    const long i = order[pos];
    const long p = ids[i];
    if (p < 0 || p >= k) {
        dist[i] = __builtin_inff();
        return;
    }
    if ((uint32_t)p == hint_sorted[pos]) {
        dist[i] = bd[pos];
        return;
    }
    const f32x4* px = reinterpret_cast<const f32x4*>(X + i * D);
    const f32x4* pc = reinterpret_cast<const f32x4*>(C + p * D);
    f32x4 xv[D / 4], cv[D / 4];  // all loads in flight before the first dependent fma
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
    dist[i] = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);
}

// After a fused sweep: rows whose winner is not their guess still need the contract's distance.
template <int D>
__device__ __forceinline__ void exact_dist_todo_body(long i, const float* __restrict__ X, long n, const float* __restrict__ C, int k,
                                                     const long* __restrict__ ids, float* __restrict__ dist) {
    if (i >= n || __float_as_uint(dist[i]) != DIST_TODO) return;
    const long p = ids[i];
    if (p < 0 || p >= k) {
        dist[i] = __builtin_inff();
        return;
    }
    const f32x4* px = reinterpret_cast<const f32x4*>(X + i * D);
    const f32x4* pc = reinterpret_cast<const f32x4*>(C + p * D);
    f32x4 xv[D / 4], cv[D / 4];
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
    dist[i] = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);
}
template <int D>
__global__ void __launch_bounds__(WG) exact_dist_todo_kernel(const float* __restrict__ X, long n,
                                                             const float* __restrict__ C, int k,
                                                             const long* __restrict__ ids,
                                                             float* __restrict__ dist) {
    exact_dist_todo_body<D>((long)blockIdx.x * WG + threadIdx.x, X, n, C, k, ids, dist);
}

// order_amb[i] = order[pos_i], hint_amb[i] = the filter's winner for that row (a very good guess)
__global__ void __launch_bounds__(WG) gather_ambiguous_kernel(const uint32_t* __restrict__ pos_sorted, long m,
                                                              long m_valid, const uint32_t* __restrict__ order,
                                                              const long* __restrict__ ids,
                                                              uint32_t* __restrict__ order_amb,
                                                              uint32_t* __restrict__ hint_amb) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= m) return;
    const uint32_t pos = pos_sorted[i < m_valid ? i : m_valid - 1];  // padding repeats the last row
    const uint32_t row = order[pos];
    const long id = ids[row];
    order_amb[i] = row;
    hint_amb[i] = id >= 0 ? (uint32_t)id : NONE;
}

// sums the per-workgroup records of a sweep into the statistics words (stats[0..1]: 64-bit, misc[4..5])
__global__ void __launch_bounds__(1024) filter_stats_reduce_kernel(const uint4* __restrict__ rec, unsigned n_rec,
                                                                   unsigned long long* __restrict__ stats,
                                                                   unsigned* __restrict__ misc) {
    __shared__ unsigned long long part[4][16];
    unsigned long long a = 0, b = 0, c = 0, d = 0;
    for (unsigned i = threadIdx.x; i < n_rec; i += 1024) {
        const uint4 r = rec[i];
        a += r.x; b += r.y; c += r.z; d += r.w;
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); d += __shfl_down(d, off);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        part[0][w] = a; part[1][w] = b; part[2][w] = c; part[3][w] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t[4] = {0, 0, 0, 0};
        for (int q = 0; q < 4; q++)
            for (int w = 0; w < 16; w++) t[q] += part[q][w];
        if (stats) { atomicAdd(&stats[0], t[0]); atomicAdd(&stats[1], t[1]); }
        atomicAdd(&misc[4], (unsigned)t[2]);
        atomicAdd(&misc[5], (unsigned)t[3]);
    }
}

// The listed rows live in AMB_SUBLISTS sub-lists of amb_cap slots (counters at misc[64 ..]).  lds_prefix[s] = entries
// before sub-list s, lds_prefix[64] = all of them; entry e of the concatenation sits at slot amb_slot(e).
__device__ __forceinline__ unsigned amb_prefix(const unsigned* __restrict__ misc, unsigned* lds_prefix) {
    if (threadIdx.x < 64) {
        const unsigned c = misc[64 + threadIdx.x];
        unsigned x = c;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off);
            if ((int)threadIdx.x >= off) x += y;
        }
        lds_prefix[threadIdx.x] = x - c;
        if (threadIdx.x == 63) lds_prefix[64] = x;
    }
    __syncthreads();
    return lds_prefix[64];
}
__device__ __forceinline__ unsigned amb_slot(const unsigned* lds_prefix, unsigned amb_cap, unsigned e) {
    unsigned lo = 0;
#pragma unroll
    for (unsigned step = 32; step > 0; step >>= 1)
        if (lds_prefix[lo + step] <= e) lo += step;
    return lo * amb_cap + (e - lds_prefix[lo]);
}

// sub-lists -> one contiguous list (the synchronous form of the exact call, whose long lists are sorted and swept)
__global__ void __launch_bounds__(WG) amb_compact_kernel(const unsigned* __restrict__ misc, unsigned amb_cap,
                                                         const uint32_t* __restrict__ list, const uint32_t* __restrict__ aux,
                                                         uint32_t* __restrict__ list_out, uint32_t* __restrict__ aux_out) {
    __shared__ unsigned pre[65];
    const unsigned total = amb_prefix(misc, pre);
    for (unsigned e = blockIdx.x * WG + threadIdx.x; e < total; e += gridDim.x * WG) {
        const unsigned sl = amb_slot(pre, amb_cap, e);
        list_out[e] = list[sl];
        aux_out[e] = aux[sl];
    }
}

// Redo of ONE listed row per workgroup, on the vector ALU: the row's own Elkan test (the per-row
// criterion of prune_mask_kernel, with the filter's winner as the guess) picks the groups, every
// thread evaluates the contract's distance (the same three fmaf chains as everywhere else) for a few
// of their centroids, and the workgroup reduces lexicographically by (distance, index).  A few
// microseconds per row whatever the neighbours in its tile need -- the MFMA sweep on a sparse list
// of rows spends its time walking unions of groups instead.
template <int D>
__device__ __forceinline__ void exact_rows_body(unsigned vblock, unsigned vgrid, const float* __restrict__ X,
                                                const float* __restrict__ C, int k, const uint32_t* __restrict__ list,
                                                const uint32_t* __restrict__ order, const int32_t* __restrict__ cperm,
                                                const float* __restrict__ dmin, int ng, const unsigned* __restrict__ misc,
                                                const uint32_t* __restrict__ aux, long* __restrict__ ids, float* __restrict__ dist,
                                                const unsigned* __restrict__ count_dev, unsigned amb_cap) {
    // (vblock / vgrid: this workgroup's index among the workgroups that do this job, and how many there are)
    __shared__ int needed[512];
    __shared__ int n_needed;
    __shared__ float red_d[WG / 64];
    __shared__ unsigned red_i[WG / 64];
    __shared__ unsigned pre[65];
    const int tid = threadIdx.x;
    // count_dev given (the asynchronous exact call): the entries are the concatenation of the sweep's sub-lists, whose
    // lengths are read here, on the device, and the grid strides over them (no host round trip; workgroups beyond the
    // list leave at once).  Otherwise: a contiguous list of gridDim.x entries.
    const bool sub = count_dev != nullptr;
    const unsigned count = sub ? amb_prefix(misc, pre) : vgrid;
    // Entries for which the filter left exactly two candidates need two chains, not a workgroup: one
    // thread each, before the workgroup-per-row loop (which then passes over them).
    if (aux) {
        for (unsigned entry_e = vblock * WG + tid; entry_e < count; entry_e += vgrid * WG) {
            const unsigned entry = sub ? amb_slot(pre, amb_cap, entry_e) : entry_e;
            const uint32_t second = aux[entry];
            if (second >= (uint32_t)k) continue;
            const long row = order[list[entry]];
            const long p = ids[row];
            if (!(p >= 0 && p < k)) continue;
            const f32x4* px = reinterpret_cast<const f32x4*>(X + row * D);
            const f32x4* p1 = reinterpret_cast<const f32x4*>(C + p * D);
            const f32x4* p2 = reinterpret_cast<const f32x4*>(C + (long)second * D);
            float xn = 0.0f, cn1 = 0.0f, ip1 = 0.0f, cn2 = 0.0f, ip2 = 0.0f;
#pragma unroll 4
            for (int q = 0; q < D / 4; q++) {
                const f32x4 xq = px[q], c1 = p1[q], c2 = p2[q];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    xn = __builtin_fmaf(xq[e], xq[e], xn);
                    cn1 = __builtin_fmaf(c1[e], c1[e], cn1);
                    ip1 = __builtin_fmaf(c1[e], xq[e], ip1);
                    cn2 = __builtin_fmaf(c2[e], c2[e], cn2);
                    ip2 = __builtin_fmaf(c2[e], xq[e], ip2);
                }
            }
            const float d1 = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip1, xn + cn1), 0.0f);
            const float d2 = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip2, xn + cn2), 0.0f);
            const bool take2 = d2 < d1 || (d2 == d1 && (long)second < p);   // lowest index on a tie
            ids[row] = take2 ? (long)second : p;
            if (dist) dist[row] = take2 ? d2 : d1;
        }
    }
    for (unsigned entry_e = vblock; entry_e < count; entry_e += vgrid) {
    const unsigned entry = sub ? amb_slot(pre, amb_cap, entry_e) : entry_e;
    const long row = order[list[entry]];
    const uint32_t second = aux ? aux[entry] : NONE;
    const long p = ids[row];
    const bool has = p >= 0 && p < k;
    // (a two-candidate entry was settled above; its ids[row] may already hold the other candidate, and
    // its aux word still says so: block-uniform)
    if (second < (uint32_t)k && has) continue;
    f32x4 xv[D / 4];
    const f32x4* px = reinterpret_cast<const f32x4*>(X + row * D);
#pragma unroll
    for (int q = 0; q < D / 4; q++) xv[q] = px[q];
    float xn = 0.0f;
#pragma unroll
    for (int q = 0; q < D / 4; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) xn = __builtin_fmaf(xv[q][e], xv[q][e], xn);
    auto contract_dist = [&](long cid) {
        const f32x4* pc = reinterpret_cast<const f32x4*>(C + cid * D);
        float cn = 0.0f, ip = 0.0f;
#pragma unroll
        for (int q = 0; q < D / 4; q++) {
            const f32x4 cu = pc[q];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                cn = __builtin_fmaf(cu[e], cu[e], cn);
                ip = __builtin_fmaf(cu[e], xv[q][e], ip);
            }
        }
        return __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);
    };
    float tau = __builtin_inff();
    if (has) {
        const float dh = contract_dist(p);
        const float cnmax = __uint_as_float(misc[0]);
        const float delta = (2.0f * D + 8.0f) * 5.9604645e-8f * (xn + cnmax) * 1.01f;
        tau = 2.0f * sqrtf(dh + delta) * (1.0f + 4.0f * 5.9604645e-8f) + 1e-30f;
        if (!(tau == tau)) tau = __builtin_inff();  // non-finite row: look everywhere
    }
    if (tid == 0) n_needed = 0;
    __syncthreads();
    for (int g = tid; g < ng; g += WG) {
        const bool need = !has || dmin[(size_t)p * ng + g] <= tau;
        if (need) needed[atomicAdd(&n_needed, 1)] = g;
    }
    __syncthreads();
    const int items = n_needed * 32;
    float bd = __builtin_inff();
    unsigned bi = NONE;
    for (int w = tid; w < items; w += WG) {
        const int cid = cperm[needed[w >> 5] * 32 + (w & 31)];
        if (cid < 0 || cid >= k) continue;
        const float dd = contract_dist(cid);
        if (dd < bd || (dd == bd && (unsigned)cid < bi)) {
            bd = dd;
            bi = (unsigned)cid;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off);
        const unsigned oi = (unsigned)__shfl_xor((int)bi, off);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    if ((tid & 63) == 0) {
        red_d[tid >> 6] = bd;
        red_i[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < WG / 64; w++)
            if (red_d[w] < bd || (red_d[w] == bd && red_i[w] < bi)) {
                bd = red_d[w];
                bi = red_i[w];
            }
        ids[row] = bi == NONE ? -1L : (long)bi;
        if (dist) dist[row] = bd;
    }
    __syncthreads();  // the shared scratch is reused by the next entry
    }  // entry
}
template <int D>
__global__ void __launch_bounds__(WG) exact_rows_kernel(const float* __restrict__ X, const float* __restrict__ C, int k,
                                                        const uint32_t* __restrict__ list,
                                                        const uint32_t* __restrict__ order,
                                                        const int32_t* __restrict__ cperm,
                                                        const float* __restrict__ dmin, int ng,
                                                        const unsigned* __restrict__ misc,
                                                        const uint32_t* __restrict__ aux, long* __restrict__ ids,
                                                        float* __restrict__ dist, const unsigned* __restrict__ count_dev,
                                                        unsigned amb_cap) {
    exact_rows_body<D>(blockIdx.x, gridDim.x, X, C, k, list, order, cperm, dmin, ng, misc, aux, ids, dist, count_dev, amb_cap);
}
// Both jobs behind a fused sweep in ONE launch: the first `todo_blocks` workgroups evaluate the distances the sweep
// left as DIST_TODO (a linear pass over dist), the others redo the listed rows.  The sweep marks listed rows
// DIST_LISTED, so the two jobs write disjoint rows.
template <int D>
__global__ void __launch_bounds__(WG) exact_finish_kernel(unsigned todo_blocks, long n, const float* __restrict__ X,
                                                          const float* __restrict__ C, int k, const uint32_t* __restrict__ list,
                                                          const uint32_t* __restrict__ order, const int32_t* __restrict__ cperm,
                                                          const float* __restrict__ dmin, int ng, const unsigned* __restrict__ misc,
                                                          const uint32_t* __restrict__ aux, long* __restrict__ ids,
                                                          float* __restrict__ dist, const unsigned* __restrict__ count_dev,
                                                          unsigned amb_cap) {
    if (blockIdx.x < todo_blocks)
        exact_dist_todo_body<D>((long)blockIdx.x * WG + threadIdx.x, X, n, C, k, ids, dist);
    else
        exact_rows_body<D>(blockIdx.x - todo_blocks, gridDim.x - todo_blocks, X, C, k, list, order, cperm, dmin, ng, misc, aux, ids,
                           dist, count_dev, amb_cap);
}

// dmin[p][g] = a lower bound of min over the members m of group g of |c_p - c_m| (prune.hip uses it in
// Elkan's test), from the hi*hi product of the sweep's fp16 split alone: the hi*hi value P is within rho of the
// three-term value (filter_rho) and that within eps of the truth (header), so P + |c_p|^2 - (eps + rho) <= the true
// squared distance.  (Round 1 evaluated all three products here: 3x the MFMAs and twice the fragment bytes for a
// bound that is 0.4 % tighter on unit rows; the table is rebuilt every Lloyd iteration on every rank.)
// One wave = 64 centroids p (two 32-row tiles as the MFMA B operand) x GPW groups, the next group's fragments
// fetched while the current one multiplies.
constexpr int DMIN_GPW = 8;

template <int D, bool THREE>
__global__ void __launch_bounds__(64, THREE ? 2 : 4)
group_min_dist_f16_kernel(const float* __restrict__ C, int k, const unsigned char* __restrict__ img, int ng,
                          const unsigned* __restrict__ misc, float eps_a, float eps_b, float* __restrict__ dmin) {
    constexpr int NS = D / 16;
    constexpr size_t GB = group_bytes(D);
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int p0 = blockIdx.x * 64;
    const int g0 = blockIdx.y * DMIN_GPW;
    const float cnmax = __uint_as_float(misc[0]);
    const bool c_bad = !(cnmax < RANGE_SQ);

    half8 xh[2][NS], xl[2][THREE ? NS : 1];
    float cnp[2], eps[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        int p = p0 + 32 * b + j;
        if (p >= k) p = k - 1;
        const f32x4* row = reinterpret_cast<const f32x4*>(C + (size_t)p * D);
        float part = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const f32x4 u = row[4 * s + 2 * h], v = row[4 * s + 2 * h + 1];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                part = __builtin_fmaf(u[e], u[e], part);
                const _Float16 hu = (_Float16)u[e];
                xh[b][s][e] = hu;
                part = __builtin_fmaf(v[e], v[e], part);
                const _Float16 hv = (_Float16)v[e];
                xh[b][s][4 + e] = hv;
                if constexpr (THREE) {
                    xl[b][s][e] = (_Float16)(u[e] - (float)hu);
                    xl[b][s][4 + e] = (_Float16)(v[e] - (float)hv);
                }
            }
        }
        cnp[b] = part + __shfl_xor(part, 32);
        eps[b] = __builtin_fmaf(eps_a, cnp[b] * 1.001f + cnmax, eps_b);
    }
    auto load = [&](int g, half8 (&ah)[NS], f32x4 (&cn)[4]) {
        const unsigned char* base = img + (size_t)g * GB;
        const half8* fh = reinterpret_cast<const half8*>(base);
#pragma unroll
        for (int s = 0; s < NS; s++) ah[s] = fh[s * 64 + lane];
        const float* cnq = reinterpret_cast<const float*>(base + misc_off(D));
#pragma unroll
        for (int q = 0; q < 4; q++) cn[q] = *reinterpret_cast<const f32x4*>(cnq + 8 * q + 4 * h);
    };
    auto compute = [&](int g, const half8 (&ah)[NS], const f32x4 (&cn)[4]) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            f32x16 a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (THREE) {   // (switch dmin_kernel = 2: all three products, the tighter eps)
                const half8* fl = reinterpret_cast<const half8*>(img + (size_t)g * GB + lo_off(D));
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[s * 64 + lane], xh[b][s], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xl[b][s], a, 0, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < NS; s++) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[b][s], a, 0, 0, 0);
            float P[16];
#pragma unroll
            for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cn[r >> 2][r & 3]);
            float m = min16(P);
            m = __builtin_fminf(m, __shfl_xor(m, 32));
            const int p = p0 + 32 * b + j;
            if (h == 0 && p < k) {
                // true squared distance >= P + |c_p|^2 - eps; padding slots carry |c|^2 = +inf
                float lb = m == __builtin_inff() ? m : __builtin_sqrtf(__builtin_fmaxf((m + cnp[b]) - eps[b], 0.0f)) * (1.0f - 1e-6f);
                if (c_bad || !(lb == lb)) lb = 0.0f;   // outside the fp16 range / non-finite: no pruning
                dmin[(size_t)p * ng + g] = lb;
            }
        }
    };
    const int g1 = g0 + DMIN_GPW < ng ? g0 + DMIN_GPW : ng;
    half8 ahA[NS], ahB[NS];
    f32x4 cnA[4], cnB[4];
    if (g0 < g1) load(g0, ahA, cnA);
    for (int g = g0; g < g1; g += 2) {
        if (g + 1 < g1) load(g + 1, ahB, cnB);
        compute(g, ahA, cnA);
        if (g + 1 >= g1) break;
        if (g + 2 < g1) load(g + 2, ahA, cnA);
        compute(g + 1, ahB, cnB);
    }
}

void filter_tau(int d, float* tau_a, float* tau_b) {
    const double u = std::ldexp(1.0, -24);
    const double q = std::sqrt((double)d) * std::ldexp(1.0, -25);
    const double m = 3.0 * d / 16.0;
    const double eps_h = 34.0 * m * u * 1.01 + 3.0 * std::ldexp(1.0, -22) + d * u * 1.01 + 2.0 * u + 2.0 * q;
    const double delta_h = (2.0 * d + 8.0) * u * 1.01;
    // rounded up generously when narrowed to float
    *tau_a = (float)((2.0 * delta_h + 2.0 * eps_h) * 1.0001);
    *tau_b = (float)(8.0 * q * 1.0001);
}

// |P(three terms) - P(hi*hi only)| <= rho_a (|x|^2 + max|c|^2) + rho_b: the two lo products are at most
// 2^-10 * 1.001 s t + q (s + t) (fp16 rounding of the hi parts), accumulating them takes 2 (d/16) MFMAs
// (17 roundings of 2u of the running magnitude each), each of the two P values is rounded once by the
// fma that forms it (u (t^2 + 2 s t) <= 2 u H); inner-product terms doubled for the factor -2, with
// s t <= H / 2 and s + t <= 1 + H / 2.
void filter_rho(int d, float* rho_a, float* rho_b) {
    const double u = std::ldexp(1.0, -24);
    const double q = std::sqrt((double)d) * std::ldexp(1.0, -25);
    const double m = 2.0 * d / 16.0;
    *rho_a = (float)((std::ldexp(1.0, -10) * 1.001 + 34.0 * m * u * 1.01 + 4.0 * u + q) * 1.0001);
    *rho_b = (float)(2.0 * q * 1.0001);
}

}  // namespace

size_t at_filter_group_bytes(int d) { return group_bytes(d); }

int at_amb_compact(at_ctx* ctx, const unsigned* misc, unsigned amb_cap, const uint32_t* list, const uint32_t* aux,
                   uint32_t* list_out, uint32_t* aux_out, hipStream_t stream) {
    (void)ctx;
    AT_LAUNCH(amb_compact_kernel, dim3(256), dim3(WG), 0, stream, misc, amb_cap, list, aux, list_out, aux_out);
    return AT_OK;
}

// Stage 1: image + sweep.  misc (device, 2 words) receives max|c|^2 and the number of listed rows;
// amb_list receives their visiting positions (unordered).
int at_filter_sweep(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const uint32_t* order,
                    const int32_t* cperm, int ng, const float* bd, const uint32_t* mask, int ngw, int collect,
                    int64_t* ids, unsigned* misc, uint32_t* amb_list, uint32_t* amb_aux, float* approx_out,
                    const uint32_t* fuse_hint_sorted, const float* fuse_dmin, float* fuse_bd_out, float* fuse_dist_out,
                    unsigned amb_cap, hipStream_t stream) {
    unsigned char* img = static_cast<unsigned char*>(at_ws(ctx, WS_CENT_IMG16, group_bytes(d) * (size_t)ng, stream));
    if (!img) return AT_E_NOMEM;
    // fused pre-pass (exact calls only): the sweep computes the guess distances and its own group masks
    const bool fused = fuse_hint_sorted && fuse_dmin && fuse_bd_out && collect;
    FusedPrepass fp{fuse_hint_sorted, c, fuse_dmin, fuse_bd_out, fuse_dist_out, nullptr, k};  // (guess generators use dist_out only)
    if (fused) {
        const bool fresh = ctx->ws[WS_PRUNE_STATS] == nullptr;
        fp.stats = static_cast<unsigned long long*>(at_ws(ctx, WS_PRUNE_STATS, 4096, stream));
        if (!fp.stats) return AT_E_NOMEM;
        if (fresh) AT_HIP(hipMemsetAsync(fp.stats, 0, 4096, stream));
    }
    // The image and max|c|^2 of exactly these centroids may be current already (at_group_min_dist_f32 built
    // them, the caller vouches that the centroids have not changed since: prepass_done bit 1).
    const bool reuse = ctx->img16_trusted && ctx->img16_c == c && ctx->img16_cperm == cperm && ctx->img16_k == k &&
                       ctx->img16_d == d && ctx->img16_ng == ng && misc == ctx->img16_misc;
    ctx->img16_trusted = 0;
    const int misc_was_clean = ctx->img16_misc_clean;
    ctx->img16_misc_clean = 0;
    (void)misc_was_clean;
    if (reuse) {
        // statistics and sub-list lengths behind max|c|^2: at_group_min_dist_f16 cleared them with the same memset that
        // cleared max|c|^2 (an odd-sized memset from misc + 1 costs two fill launches)
        if (!misc_was_clean) AT_HIP(hipMemsetAsync(misc + 1, 0, 127 * sizeof(unsigned), stream));
    } else {
        ctx->img16_c = nullptr;
        AT_HIP(hipMemsetAsync(misc, 0, 128 * sizeof(unsigned), stream));  // max|c|^2, statistics, sub-list lengths
        AT_LAUNCH(prep_centroids_f16_kernel, dim3(ng), dim3(WG), 0, stream, c, k, d, cperm, img, misc);
    }
    float ta = 0.0f, tb = 0.0f, ra = 0.0f, rb = 0.0f;
    filter_tau(d, &ta, &tb);
    filter_rho(d, &ra, &rb);
    const int screen = ctx->dbg.filter_screen != 0;   // (switch: 0 = always evaluate all three products)
    // Rows per wave = 32 * NB.  Lloyd-sized exact sweeps (rows sorted by guess, <= 8 M of them) run two
    // tiles per wave at three waves per SIMD (168 registers, no second fragment set: occupancy hides the
    // loads; 3 % faster than four tiles at two waves); the long tokenise sweeps and the guess generators
    // keep four tiles per wave.  AT_FILTER_NB=1|2|4 forces the choice (A/B aid).
    const int nbv = ctx->dbg.filter_nb;
    // Short sweeps (a shard of an N-GPU run: 262 144 rows are 4096 two-tile waves for 3072 slots, i.e. two rounds where
    // 1.33 would do) run ONE tile per wave at four waves per SIMD: 8192 half-size waves on 4096 slots.
    const bool one_tile = fused && d == 64 && !ctx->dbg.filter_wps2 && (nbv ? nbv == 1 : n <= (int64_t)1 << 19);
    const bool wps3 = !one_tile && fused && d == 64 && (nbv ? nbv == 2 : n <= (int64_t)8 << 20) && !ctx->dbg.filter_wps2;
    const int NB = one_tile ? 1 : (wps3 || nbv == 2) ? 2 : 4;
    const dim3 grid((unsigned)((n + 32 * NB - 1) / (32 * NB)));
    AT_REQUIRE(d == 64 || d == 128, "at_filter_sweep: d must be 64 or 128");
    // statistics (switch filter_stats; off by default): one record per workgroup, summed behind the sweep
    const unsigned n_wg = d == 128 ? (unsigned)((n + 63) / 64) : grid.x;
    uint4* blk_stats = nullptr;
    if (ctx->dbg.filter_stats) {
        blk_stats = static_cast<uint4*>(at_ws(ctx, WS_FILTER_BLKSTATS, (size_t)n_wg * sizeof(uint4), stream));
        if (!blk_stats) return AT_E_NOMEM;
    }
    // exact calls under the switch filter_timing (bench.py): two timing events around the kernel, read by
    // at_filter_resolve_pending / the synchronous form once the call's statistics have arrived.  The product leaves
    // the switch off: no event is recorded and none is read.
    at_filter_slot& tslot = ctx->fring[(ctx->filter_slot >= 0 && ctx->filter_slot <= AT_FILTER_RING) ? ctx->filter_slot : AT_FILTER_RING];
    const bool timed = collect && ctx->dbg.filter_timing != 0;
    tslot.timed = 0;
    if (timed) {
        for (int i = 0; i < 2; i++)
            if (!tslot.ev[i])  // no system-scope release at the event: it would charge an L2 write-back to the kernel
                AT_HIP(hipEventCreateWithFlags(&tslot.ev[i], hipEventDisableSystemFence));
        AT_HIP(hipEventRecord(tslot.ev[0], stream));
    }
#define AT_FILTER_LAUNCH(DD, NBB, GG, FF, GRID)                                                                      \
    AT_LAUNCH((assign_f16filter_kernel<DD, NBB, GG, FF>), GRID, dim3(64), 0, stream, x, (long)n, img, ng, order, \
                       bd, mask, ngw, misc, ta, tb, ra, rb, screen, collect, reinterpret_cast<long*>(ids), amb_list,      \
                       amb_aux, approx_out, fp, blk_stats, amb_cap)
    // exact calls (collect) and guess generators are separate instantiations: the guess path's code
    // would otherwise cost the exact sweep registers it does not have
    const dim3 grid64((unsigned)((n + 63) / 64));
    if (d == 128) {  // two tiles per wave: the fragment sets of d = 128 leave no registers for four
        if (fused) AT_FILTER_LAUNCH(128, 2, false, true, grid64);
        else if (collect) AT_FILTER_LAUNCH(128, 2, false, false, grid64);
        else AT_FILTER_LAUNCH(128, 2, true, false, grid64);
    } else if (NB == 4) {
        if (fused) AT_FILTER_LAUNCH(64, 4, false, true, grid);
        else if (collect) AT_FILTER_LAUNCH(64, 4, false, false, grid);
        else AT_FILTER_LAUNCH(64, 4, true, false, grid);
    } else if (one_tile) {
        AT_LAUNCH((assign_f16filter_kernel<64, 1, false, true, 4>), grid, dim3(64), 0, stream, x, (long)n, img, ng,
                           order, bd, mask, ngw, misc, ta, tb, ra, rb, screen, collect, reinterpret_cast<long*>(ids),
                           amb_list, amb_aux, approx_out, fp, blk_stats, amb_cap);
    } else {
        if (wps3) {
            AT_LAUNCH((assign_f16filter_kernel<64, 2, false, true, 3>), grid, dim3(64), 0, stream, x, (long)n, img, ng,
                               order, bd, mask, ngw, misc, ta, tb, ra, rb, screen, collect, reinterpret_cast<long*>(ids),
                               amb_list, amb_aux, approx_out, fp, blk_stats, amb_cap);
        } else if (fused) AT_FILTER_LAUNCH(64, 2, false, true, grid);
        else if (collect) AT_FILTER_LAUNCH(64, 2, false, false, grid);
        else AT_FILTER_LAUNCH(64, 2, true, false, grid);
    }
#undef AT_FILTER_LAUNCH
    if (timed) {
        AT_HIP(hipEventRecord(tslot.ev[1], stream));
        tslot.timed = 1;
    }
    if (blk_stats) {
        AT_LAUNCH(filter_stats_reduce_kernel, dim3(1), dim3(1024), 0, stream, blk_stats, n_wg, fused ? fp.stats : nullptr, misc);
    }
    return AT_OK;
}

int at_exact_dist_rows(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const int64_t* ids,
                       float* dist, const uint32_t* order, const uint32_t* hint_sorted, const float* bd,
                       hipStream_t stream) {
    (void)ctx;
    const dim3 grid((unsigned)((n + WG - 1) / WG));
    if (order && hint_sorted && bd) {  // guesses with their pre-pass distances available
        if (d == 64)
            AT_LAUNCH(exact_dist_visit_kernel<64>, grid, dim3(WG), 0, stream, x, (long)n, c, k, order,
                               hint_sorted, bd, reinterpret_cast<const long*>(ids), dist);
        else
            AT_LAUNCH(exact_dist_visit_kernel<128>, grid, dim3(WG), 0, stream, x, (long)n, c, k, order,
                               hint_sorted, bd, reinterpret_cast<const long*>(ids), dist);
        return AT_OK;
    }
    if (d == 64)
        AT_LAUNCH(exact_dist_rows_kernel<64>, grid, dim3(WG), 0, stream, x, (long)n, c, k,
                           reinterpret_cast<const long*>(ids), dist);
    else
        AT_LAUNCH(exact_dist_rows_kernel<128>, grid, dim3(WG), 0, stream, x, (long)n, c, k,
                           reinterpret_cast<const long*>(ids), dist);
    return AT_OK;
}

// Sorts the m_valid listed positions (visiting order keeps the tiles of the redo pass coherent) and
// expands them to m >= m_valid entries (the fp32 sweep wants at least 20 rows).
int at_filter_gather_ambiguous(at_ctx* ctx, uint32_t* amb_list, uint32_t* amb_sorted, int64_t m_valid, int64_t m,
                               const uint32_t* order, const int64_t* ids, uint32_t* order_amb, uint32_t* hint_amb,
                               hipStream_t stream) {
    size_t tmp_bytes = 0;
    AT_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, amb_list, amb_sorted, (size_t)m_valid, 0, 32, stream));
    void* tmp = at_ws(ctx, WS_SORT_TMP, tmp_bytes, stream);
    if (!tmp) return AT_E_NOMEM;
    AT_HIP(rocprim::radix_sort_keys(tmp, tmp_bytes, amb_list, amb_sorted, (size_t)m_valid, 0, 32, stream));
    AT_LAUNCH(gather_ambiguous_kernel, dim3((unsigned)((m + WG - 1) / WG)), dim3(WG), 0, stream, amb_sorted,
                       (long)m, (long)m_valid, order, reinterpret_cast<const long*>(ids), order_amb, hint_amb);
    return AT_OK;
}


// Stage 2 for short lists: one workgroup per listed row (see exact_rows_kernel).
int at_filter_redo_rows(at_ctx* ctx, const float* x, int d, const float* c, int k, const uint32_t* list, int64_t m,
                        const uint32_t* order, const int32_t* cperm, const float* dmin, int ng, const unsigned* misc,
                        const uint32_t* aux, int64_t* ids, float* dist, const unsigned* count_dev, unsigned amb_cap,
                        hipStream_t stream) {
    (void)ctx;
    if (m <= 0) return AT_OK;   // m = list length, or (with count_dev) the number of workgroups to launch
    AT_REQUIRE(ng <= 512, "at_filter_redo_rows: ng > 512");
    if (d == 64)
        AT_LAUNCH(exact_rows_kernel<64>, dim3((unsigned)m), dim3(WG), 0, stream, x, c, k, list, order, cperm, dmin,
                           ng, misc, aux, reinterpret_cast<long*>(ids), dist, count_dev, amb_cap);
    else
        AT_LAUNCH(exact_rows_kernel<128>, dim3((unsigned)m), dim3(WG), 0, stream, x, c, k, list, order, cperm,
                           dmin, ng, misc, aux, reinterpret_cast<long*>(ids), dist, count_dev, amb_cap);
    return AT_OK;
}

// The distance pass and the redo of the listed rows behind a fused sweep, one launch (asynchronous exact calls).
int at_filter_finish(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const uint32_t* list, int64_t redo_wgs,
                     const uint32_t* order, const int32_t* cperm, const float* dmin, int ng, const unsigned* misc,
                     const uint32_t* aux, int64_t* ids, float* dist, const unsigned* count_dev, unsigned amb_cap,
                     hipStream_t stream) {
    (void)ctx;
    AT_REQUIRE(ng <= 512 && dist && redo_wgs > 0, "at_filter_finish: bad arguments");
    const unsigned todo_blocks = (unsigned)((n + WG - 1) / WG);
    const dim3 grid(todo_blocks + (unsigned)redo_wgs);
    if (d == 64)
        AT_LAUNCH(exact_finish_kernel<64>, grid, dim3(WG), 0, stream, todo_blocks, (long)n, x, c, k, list, order, cperm, dmin, ng,
                  misc, aux, reinterpret_cast<long*>(ids), dist, count_dev, amb_cap);
    else
        AT_LAUNCH(exact_finish_kernel<128>, grid, dim3(WG), 0, stream, todo_blocks, (long)n, x, c, k, list, order, cperm, dmin, ng,
                  misc, aux, reinterpret_cast<long*>(ids), dist, count_dev, amb_cap);
    return AT_OK;
}

int at_group_min_dist_f16(at_ctx* ctx, const float* c, int k, int d, const int32_t* cperm, int ng, float* dmin,
                          hipStream_t stream) {
    unsigned char* img = static_cast<unsigned char*>(at_ws(ctx, WS_CENT_IMG16, group_bytes(d) * (size_t)ng, stream));
    unsigned* misc = static_cast<unsigned*>(at_ws(ctx, WS_FILTER_MISC, 1024, stream));  // word 0 = max|c|^2, as the sweep wants it
    if (!img || !misc) return AT_E_NOMEM;
    AT_HIP(hipMemsetAsync(misc, 0, 128 * sizeof(unsigned), stream));   // max|c|^2 and, for the sweep that follows, its counters
    AT_LAUNCH(prep_centroids_f16_kernel, dim3(ng), dim3(WG), 0, stream, c, k, d, cperm, img, misc);
    ctx->img16_c = c; ctx->img16_cperm = cperm; ctx->img16_k = k; ctx->img16_d = d; ctx->img16_ng = ng;
    ctx->img16_misc = misc;
    ctx->img16_misc_clean = 1;
    // eps of the filter's budget (header of this file) without the contract's delta: tau = 2 delta + 2 eps
    float ta = 0.0f, tb = 0.0f;
    filter_tau(d, &ta, &tb);
    const double u = std::ldexp(1.0, -24);
    const bool three = ctx->dbg.dmin_kernel == 2;
    float ra = 0.0f, rb = 0.0f;
    if (!three) filter_rho(d, &ra, &rb);     // the kernel keeps the hi*hi product only
    const float eps_a = (float)(0.5 * ((double)ta - 2.0 * (2.0 * d + 8.0) * u * 1.01) * 1.001) + ra;
    const float eps_b = 0.5f * tb * 1.001f + rb;
    const dim3 grid((unsigned)((k + 63) / 64), (unsigned)((ng + DMIN_GPW - 1) / DMIN_GPW));
    if (d == 64 && three)
        AT_LAUNCH((group_min_dist_f16_kernel<64, true>), grid, dim3(64), 0, stream, c, k, img, ng, misc, eps_a, eps_b, dmin);
    else if (d == 64)
        AT_LAUNCH((group_min_dist_f16_kernel<64, false>), grid, dim3(64), 0, stream, c, k, img, ng, misc, eps_a, eps_b, dmin);
    else if (three)
        AT_LAUNCH((group_min_dist_f16_kernel<128, true>), grid, dim3(64), 0, stream, c, k, img, ng, misc, eps_a, eps_b, dmin);
    else
        AT_LAUNCH((group_min_dist_f16_kernel<128, false>), grid, dim3(64), 0, stream, c, k, img, ng, misc, eps_a, eps_b, dmin);
    return AT_OK;
}

// Completes dist after a fused sweep that was given dist_out (see exact_dist_todo_kernel).
int at_exact_dist_todo(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const int64_t* ids,
                       float* dist, hipStream_t stream) {
    (void)ctx;
    const dim3 grid((unsigned)((n + WG - 1) / WG));
    if (d == 64)
        AT_LAUNCH(exact_dist_todo_kernel<64>, grid, dim3(WG), 0, stream, x, (long)n, c, k,
                           reinterpret_cast<const long*>(ids), dist);
    else
        AT_LAUNCH(exact_dist_todo_kernel<128>, grid, dim3(WG), 0, stream, x, (long)n, c, k,
                           reinterpret_cast<const long*>(ids), dist);
    return AT_OK;
}

namespace {
__global__ void __launch_bounds__(WG) iota_pad_kernel(int32_t* __restrict__ perm, int n_live, int n_total) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i < n_total) perm[i] = i < n_live ? i : -1;
}
}  // namespace

// Guess generator for rows whose own order is coherent (consecutive frames of clips): per 32-row tile the
// nearest of the ng group means decides, through the neighbour table gnbr [ng][ceil(ng/32)], which groups
// of centroids are searched -- one launch, rows read once (assign.hip: at_assign_coarse_f32).
int at_filter_coarse(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const int32_t* cperm, int ng,
                     const float* means, const uint32_t* gnbr, int64_t* ids, float* dist, hipStream_t stream) {
    const int ngm = (ng + 31) / 32;              // groups of the means image
    const int ngw = (ng + 31) / 32;
    unsigned char* img = static_cast<unsigned char*>(at_ws(ctx, WS_CENT_IMG16, group_bytes(d) * (size_t)ng, stream));
    unsigned char* img_m = static_cast<unsigned char*>(at_ws(ctx, WS_CENT_IMG16B, group_bytes(d) * (size_t)ngm, stream));
    int32_t* perm_m = static_cast<int32_t*>(at_ws(ctx, WS_IOTA, sizeof(int32_t) * (size_t)ngm * 32, stream));
    unsigned* misc = static_cast<unsigned*>(at_ws(ctx, WS_FILTER_MISC, 1024, stream));
    if (!img || !img_m || !perm_m || !misc) return AT_E_NOMEM;
    ctx->img16_c = nullptr;
    ctx->img16_misc_clean = 0;
    AT_LAUNCH(iota_pad_kernel, dim3((ngm * 32 + WG - 1) / WG), dim3(WG), 0, stream, perm_m, ng, ngm * 32);
    AT_HIP(hipMemsetAsync(misc, 0, 64 * sizeof(unsigned), stream));
    AT_LAUNCH(prep_centroids_f16_kernel, dim3(ngm), dim3(WG), 0, stream, means, ng, d, perm_m, img_m,
                       static_cast<unsigned*>(nullptr));
    AT_LAUNCH(prep_centroids_f16_kernel, dim3(ng), dim3(WG), 0, stream, c, k, d, cperm, img, misc);
    float ta = 0.0f, tb = 0.0f, ra = 0.0f, rb = 0.0f;
    filter_tau(d, &ta, &tb);
    filter_rho(d, &ra, &rb);
    FusedPrepass fp{};
    fp.dist_out = dist;
    fp.k = k;
    fp.means_img = img_m;
    fp.gnbr = gnbr;
    fp.ngm = ngm;
    const float* no_bd = nullptr;
    const uint32_t* no_mask = nullptr;
    const uint32_t* no_order = nullptr;
    uint32_t* no_list = nullptr;
    float* no_approx = nullptr;
    if (d == 128)
        AT_LAUNCH((assign_f16filter_kernel<128, 2, true, false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, x,
                           (long)n, img, ng, no_order, no_bd, no_mask, ngw, misc, ta, tb, ra, rb, 1, 0,
                           reinterpret_cast<long*>(ids), no_list, no_list, no_approx, fp, static_cast<uint4*>(nullptr), 0u);
    else
        AT_LAUNCH((assign_f16filter_kernel<64, 4, true, false>), dim3((unsigned)((n + 127) / 128)), dim3(64), 0, stream, x,
                           (long)n, img, ng, no_order, no_bd, no_mask, ngw, misc, ta, tb, ra, rb, 1, 0,
                           reinterpret_cast<long*>(ids), no_list, no_list, no_approx, fp, static_cast<uint4*>(nullptr), 0u);
    return AT_OK;
}
