// filter_wg.hip -- stage 1 of the exact sweep (filter.hip) rebuilt around ONE fragment stream per workgroup.
//
// Accelerates faiss.IndexFlatL2.search(x, 1) at processors/spec_tokenizer.py:77 and inside faiss.Kmeans.train
// (processors/cluster_creator.py:54-56 of danavery/audio-tokens); same contract, same bits as filter.hip.
//
// Round 2's kernel gave every wave its own walk: each wave fetched the fp16 fragments of every group it visited
// from L2 into registers (9.7 GB of L2->L1 traffic per Lloyd sweep, 17x the algorithmic bytes), waited for them
// (57 % of wave residency parked) and refined ambiguous tiles in the middle of the walk.  Here a workgroup of four
// waves owns 4*NB consecutive 32-row tiles of the visiting order (NB = 1, 2):
//
//   * the needed groups of all its tiles are merged into ONE list (group | tile bits);
//   * a group's hi fragments + |c|^2 (d/16 KiB + 256 B) are staged in LDS ONCE per workgroup by LDS-DMA
//     (global_load_lds_dwordx4, every wave issues one 1 KiB piece), in a ring of three stages: the stage being
//     multiplied, and two in flight behind counted s_waitcnt vmcnt -- the L2 latency is never waited for;
//   * tiles are dealt to the waves round robin (wave w owns tiles w and w + 4): the tiles that need a group are
//     mostly neighbours in the visiting order, so a visit's work spreads over the waves;
//   * the walk multiplies hi*hi only (4 MFMAs per tile and group) and screens the result against a threshold that
//     is fixed per row for the whole walk (the cap the guess gives: no candidate above it can enter the row's
//     (best, runner-up, third) triple); tiles that pass are only MARKED (a bit per tile and group in LDS);
//   * the marked pairs are refined after the walk, each wave for its own tiles: hi + lo fragments straight from L2
//     (they are few: ~5 % of the pairs), all three products, the triple updated exactly as in filter.hip.
//
// Why the fixed threshold is enough: filter.hip screened against b3 + rho with b3 = the row's third-best value so
// far, starting from cap = P(guess) + 3 tau.  b3 never exceeds cap, so a tile whose hi*hi values all stay above
// cap + rho also stays above every b3 + rho the old walk would have used: the set marked here is a superset of the
// set refined there, and a pair that is refined needlessly only offers candidates that lose.  The triple that comes
// out is the same; rows are settled / listed by the same test.
#include "filter_common.h"

namespace {
using namespace atf;

constexpr int WGT = 256;     // four waves
constexpr int RING = 3;

template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    // every earlier LDS read of this wave has returned (the stage they read may be overwritten after the barrier),
    // all but the N youngest vector-memory operations (the LDS-DMA pieces of the stage after this one) have landed
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(const unsigned char* gsrc_lane, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const unsigned char* gsrc_lane, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// LDS carve (one dynamic array: a second __shared__ object beside an LDS-DMA target makes hipcc drain the DMA
// before every ds_read).  Offsets in bytes.
template <int D, int NB>
struct Carve {
    static constexpr int NS = D / 16;
    static constexpr int NT = 4 * NB;                                  // tiles of the workgroup
    static constexpr size_t STAGE = (size_t)NS * 1024 + 256;           // hi fragments, |c|^2 + indices
    static constexpr size_t ring = 0;
    static constexpr size_t xl = ring + RING * STAGE;                  // lo parts of the rows [NT][NS][64] half8
    static constexpr size_t need = xl + (size_t)NT * NS * 1024;        // need masks [NT][8] u64
    static constexpr size_t glist = need + (size_t)NT * 64;            // merged list, 512 u32
    static constexpr size_t refb = glist + 2048;                       // marked pairs [NT][16] u32 (bit per group)
    static constexpr size_t stash = refb + (size_t)NT * 64;            // per tile [4][32] words: tau, row, gbd, hint
    static constexpr size_t total = stash + (size_t)NT * 512;
};

template <int D, int NB>
__global__ void __launch_bounds__(WGT, NB == 1 ? 4 : 3)
assign_f16filter_wg_kernel(const float* __restrict__ X, long n, const unsigned char* __restrict__ img, int ng,
                           const uint32_t* __restrict__ order, unsigned* __restrict__ misc, float tau_a, float tau_b,
                           float rho_a, float rho_b, int screen, long* __restrict__ ids, uint32_t* __restrict__ amb_list,
                           uint32_t* __restrict__ amb_aux, float* __restrict__ approx_out, FusedPrepass fp,
                           uint4* __restrict__ blk_stats, unsigned amb_cap) {
    using L = Carve<D, NB>;
    constexpr int NS = L::NS;
    constexpr int NT = L::NT;
    constexpr size_t GB = group_bytes(D);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31;
    const int h = lane >> 5;
    const long pos0 = (long)blockIdx.x * (32 * NT);
    if (pos0 >= n) return;                      // (uniform over the workgroup: before any barrier)
    const float cnmax = __uint_as_float(misc[0]);
    const bool c_bad = !(cnmax < RANGE_SQ);

    half8 xh[NB][NS];
    float thr[NB], cap0[NB];
    unsigned st_needed = 0, st_total = 0;

    // ---- prologue: this wave's tiles (w, w + 4) exactly as filter.hip's fused pre-pass does them ---------------
    unsigned long long need[NB][8];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int T = w + 4 * b;
        long pos = pos0 + 32 * T + j;
        const bool live = pos < n;
        if (!live) pos = n - 1;
        const unsigned r = order ? order[pos] : (unsigned)pos;
        const f32x4* p = reinterpret_cast<const f32x4*>(X + (size_t)r * D);
        f32x4 xu[NS], xv[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            xu[s] = p[4 * s + 2 * h];
            xv[s] = p[4 * s + 2 * h + 1];
        }
        // the contract's distance to the guess: three ascending fmaf chains whose state hops between the two lanes
        // that hold the even and odd 8-feature chunks of the row (filter.hip)
        const uint32_t g = fp.hint_sorted[pos];
        const bool has = g < (uint32_t)fp.k;
        const uint32_t hint = has ? g : NONE;
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
        const float nrm = xn;
        const float dh = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, xn + cn), 0.0f);   // what the fp32 sweep computes for (x, c_p)
        const float bd = has ? dh : __builtin_inff();
        if (h == 0 && live) fp.bd_out[pos] = bd;
        const float delta = (2.0f * D + 8.0f) * 5.9604645e-8f * (xn + cnmax) * 1.01f;
        // Elkan radius (2R); rows without a guess need every group, positions past n none
        const float mtau = !live ? -1.0f
                                 : (has ? 2.0f * sqrtf(dh + delta) * (1.0f + 4.0f * 5.9604645e-8f) + 1e-30f : __builtin_inff());
        // fp16 split of the row: hi parts stay in registers (the MFMA B operand), lo parts wait in LDS
        half8* xl_lds = reinterpret_cast<half8*>(lds + L::xl) + (size_t)T * NS * 64;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const f32x4 u = xu[s], v = xv[s];
            half8 xlo;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const _Float16 hu = (_Float16)u[e];
                xh[b][s][e] = hu;
                xlo[e] = (_Float16)(u[e] - (float)hu);
                const _Float16 hv = (_Float16)v[e];
                xh[b][s][4 + e] = hv;
                xlo[4 + e] = (_Float16)(v[e] - (float)hv);
            }
            xl_lds[s * 64 + lane] = xlo;
        }
        const float tau = __builtin_fmaf(tau_a, nrm * 1.001f + cnmax, tau_b);
        const float rho = screen ? __builtin_fmaf(rho_a, nrm * 1.001f + cnmax, rho_b) : __builtin_inff();
        const bool bad = c_bad || !(nrm < RANGE_SQ);
        // candidates above the cap can neither be the arg-min nor within tau of it: the guess itself (always
        // admitted by the masks) has P <= bd - |x|^2 + eps
        cap0[b] = bd < __builtin_inff() ? (bd - nrm) + 3.0f * tau : __builtin_inff();
        thr[b] = cap0[b] + rho;
        // what only the epilogue needs sits out the walk in LDS
        if (h == 0) {
            float* st = reinterpret_cast<float*>(lds + L::stash) + (size_t)T * 128;
            st[j] = bad ? -1.0f : tau;                          // (a negative threshold stands for "not sane")
            reinterpret_cast<unsigned*>(st)[32 + j] = r;
            st[64 + j] = bd;
            reinterpret_cast<unsigned*>(st)[96 + j] = hint;
        }

        // group masks of the tile (prune.hip prune_mask_kernel's test): a group is needed iff dmin[p][g] <= the
        // largest radius of some run of equal guesses p; lanes stand for groups, one coalesced read of the run's
        // dmin row per 64 groups
#pragma unroll
        for (int it = 0; it < 8; it++) need[b][it] = 0ull;
        const bool tile_live = pos0 + 32 * T < n;
        const bool nohint = live && hint == NONE;
        if (__builtin_amdgcn_ballot_w64(nohint) != 0) {
#pragma unroll
            for (int it = 0; it < 8; it++) need[b][it] = ~0ull;
        } else {
            unsigned long long todo = __builtin_amdgcn_ballot_w64(live) & 0xffffffffull;
            while (todo != 0) {
                const int leader = __builtin_ctzll(todo);
                const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)hint, leader);
                const bool in_run = live && hint == pl;
                todo &= ~__builtin_amdgcn_ballot_w64(in_run);
                float t = in_run ? mtau : -1.0f;
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) t = __builtin_fmaxf(t, __shfl_xor(t, off));
                const float* drow = fp.dmin + (size_t)pl * ng;
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int gg = 64 * it + lane;
                    if (64 * it < ng) need[b][it] |= __builtin_amdgcn_ballot_w64(gg < ng && drow[gg] <= t);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            if (64 * it >= ng || !tile_live) need[b][it] = 0ull;
            else if (ng - 64 * it < 64) need[b][it] &= (1ull << (ng - 64 * it)) - 1ull;
            st_needed += (unsigned)__builtin_popcountll(need[b][it]);
        }
        st_total += tile_live ? (unsigned)ng : 0u;
        if (lane == 0) {
            unsigned long long* nm = reinterpret_cast<unsigned long long*>(lds + L::need) + (size_t)T * 8;
#pragma unroll
            for (int it = 0; it < 8; it++) nm[it] = need[b][it];
        }
        if (lane < 16) reinterpret_cast<uint32_t*>(lds + L::refb)[T * 16 + lane] = 0u;
    }
    __syncthreads();

    // ---- one list for the workgroup: entry = group | (tile bits << 9) ------------------------------------------
    uint32_t* glist = reinterpret_cast<uint32_t*>(lds + L::glist);
    int cnt = 0;
    {
        const unsigned long long* nm = reinterpret_cast<const unsigned long long*>(lds + L::need);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            if (64 * it >= ng) break;
            unsigned f = 0;
#pragma unroll
            for (int T = 0; T < NT; T++) f |= (unsigned)((nm[T * 8 + it] >> lane) & 1ull) << T;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(f != 0);
            if (f != 0 && w == 0)
                glist[cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = (uint32_t)(64 * it + lane) | (f << 9);
            cnt += __builtin_popcountll(bal);
        }
    }
    __syncthreads();

    // ---- the walk ---------------------------------------------------------------------------------------------
    unsigned char* ring = lds + L::ring;
    const unsigned lane_off = (unsigned)w * 1024u + (unsigned)lane * 16u;
    auto stage = [&](int slot_i, unsigned e) {   // LDS-DMA of a visit's group into ring slot slot_i
        const unsigned char* base = img + (size_t)(e & 511u) * GB;
        unsigned char* slot = ring + (size_t)slot_i * L::STAGE;
        if constexpr (NS == 4) {
            dma16(base + lane_off, slot + (size_t)w * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < NS / 4; q++) dma16(base + lane_off + q * 4096u, slot + (size_t)(4 * q + w) * 1024);
        }
        if (w == 0) dma4(base + misc_off(D) + lane * 4, slot + (size_t)NS * 1024);
    };
    uint32_t ve_cur = cnt > 0 ? glist[0] : 0u, ve_nxt = cnt > 1 ? glist[1] : 0u;
    if (cnt > 0) stage(0, (unsigned)__builtin_amdgcn_readfirstlane((int)ve_cur));
    if (cnt > 1) stage(1, (unsigned)__builtin_amdgcn_readfirstlane((int)ve_nxt));
    int cur_slot = 0;                      // slot of visit i; visit i + 2 goes into the slot visit i - 1 has left
    unsigned n_hh = 0;
    constexpr int PIECES = NS / 4;          // LDS-DMA instructions per wave and stage (wave 0: one more)
    for (int i = 0; i < cnt; i++) {
        const uint32_t ve_st = i + 2 < cnt ? glist[i + 2] : 0u;
        if (i + 1 < cnt) {
            if (w == 0) wait_vm_and_barrier<PIECES + 1>();
            else wait_vm_and_barrier<PIECES>();
        } else {
            wait_vm_and_barrier<0>();
        }
        const int prev_slot = cur_slot == 0 ? RING - 1 : cur_slot - 1;
        if (i + 2 < cnt) stage(prev_slot, (unsigned)__builtin_amdgcn_readfirstlane((int)ve_st));
        const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)ve_cur);
        ve_cur = ve_nxt;
        ve_nxt = ve_st;
        const unsigned char* slot = ring + (size_t)cur_slot * L::STAGE;
        cur_slot = cur_slot == RING - 1 ? 0 : cur_slot + 1;
        const unsigned mine = ((e >> (9 + w)) & 1u) | (NB == 2 ? (((e >> (13 + w)) & 1u) << 1) : 0u);
        if (mine == 0) continue;
        half8 ah[NS];
        f32x4 cnv[4];
#pragma unroll
        for (int s = 0; s < NS; s++) ah[s] = reinterpret_cast<const half8*>(slot)[s * 64 + lane];
        const float* cnp = reinterpret_cast<const float*>(slot + (size_t)NS * 1024);
#pragma unroll
        for (int q = 0; q < 4; q++) cnv[q] = *reinterpret_cast<const f32x4*>(cnp + 8 * q + 4 * h);
        const unsigned g = e & 511u;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (!((mine >> b) & 1u)) continue;     // wave-uniform
            f32x16 a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NS; s++) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[b][s], a, 0, 0, 0);
            float P[16];
#pragma unroll
            for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cnv[r >> 2][r & 3]);
            const bool pass = __builtin_amdgcn_ballot_w64(min16(P) < thr[b]) != 0;
            n_hh++;
            if (pass && lane == 0) {
                uint32_t* rb = reinterpret_cast<uint32_t*>(lds + L::refb) + (w + 4 * b) * 16;
                rb[g >> 5] |= 1u << (g & 31);
            }
        }
    }

    // ---- the marked pairs: all three products, the triple as filter.hip keeps it -------------------------------
    float b1[NB], b2[NB], b3[NB];
    unsigned i1[NB], i2[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        b1[b] = b2[b] = b3[b] = cap0[b];
        i1[b] = i2[b] = NONE;
    }
    unsigned n_ref = 0;
    {
        // (the marks are this wave's own stores: LDS operations of one wave execute in order)
        const uint32_t* rb0 = reinterpret_cast<const uint32_t*>(lds + L::refb) + w * 16;
        const uint32_t* rb1 = reinterpret_cast<const uint32_t*>(lds + L::refb) + (w + 4) * 16;
        auto load_pair = [&](unsigned g, half8 (&ah)[NS], half8 (&al)[NS], f32x4 (&cnv)[4]) {
            const unsigned char* base = img + (size_t)g * GB;
            const half8* fh = reinterpret_cast<const half8*>(base);
            const half8* fl = reinterpret_cast<const half8*>(base + lo_off(D));
#pragma unroll
            for (int s = 0; s < NS; s++) {
                ah[s] = fh[s * 64 + lane];
                al[s] = fl[s * 64 + lane];
            }
            const float* cnp = reinterpret_cast<const float*>(base + misc_off(D));
#pragma unroll
            for (int q = 0; q < 4; q++) cnv[q] = *reinterpret_cast<const f32x4*>(cnp + 8 * q + 4 * h);
        };
        auto refine_pair = [&](unsigned g, unsigned which, const half8 (&ah)[NS], const half8 (&al)[NS], const f32x4 (&cnv)[4]) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (!((which >> b) & 1u)) continue;
                const half8* xl_lds = reinterpret_cast<const half8*>(lds + L::xl) + (size_t)(w + 4 * b) * NS * 64;
                f32x16 a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < NS; s++) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[b][s], a, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], xh[b][s], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xl_lds[s * 64 + lane], a, 0, 0, 0);
                }
                float P[16];
#pragma unroll
                for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cnv[r >> 2][r & 3]);
                n_ref++;
                if (__builtin_amdgcn_ballot_w64(min16(P) < b3[b]) != 0) {
                    float v1 = b1[b], v2 = b2[b], v3 = b3[b];
                    unsigned j1 = i1[b], j2 = i2[b];
                    const unsigned base = g * 32u + 4u * h;
#pragma unroll
                    for (int r = 0; r < 16; r++) insert3(P[r], base + (unsigned)((r & 3) + 8 * (r >> 2)), v1, v2, v3, j1, j2);
                    b1[b] = v1; b2[b] = v2; b3[b] = v3;
                    i1[b] = j1; i2[b] = j2;
                }
            }
        };
        // walk the marks in group order, the next pair's fragments requested while this one multiplies
        half8 ahA[NS], alA[NS], ahB[NS], alB[NS];
        f32x4 cnA[4], cnB[4];
        int wi = 0;
        uint32_t m0 = rb0[0], m1 = NB == 2 ? rb1[0] : 0u;
        auto next_mark = [&](unsigned& g, unsigned& which) -> bool {   // wave-uniform
            for (;;) {
                const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(m0 | m1));
                if (u != 0) {
                    const int bit = __builtin_ctz(u);
                    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m0);
                    const uint32_t f1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m1);
                    g = (unsigned)(wi * 32 + bit);
                    which = ((f0 >> bit) & 1u) | (((f1 >> bit) & 1u) << 1);
                    m0 &= ~(1u << bit);
                    m1 &= ~(1u << bit);
                    return true;
                }
                if (++wi >= 16 || wi * 32 >= ng) return false;
                m0 = rb0[wi];
                m1 = NB == 2 ? rb1[wi] : 0u;
            }
        };
        unsigned gA = 0, wA = 0, gB = 0, wB = 0;
        bool haveA = next_mark(gA, wA);
        if (haveA) load_pair(gA, ahA, alA, cnA);
        while (haveA) {
            const bool haveB = next_mark(gB, wB);
            if (haveB) load_pair(gB, ahB, alB, cnB);
            refine_pair(gA, wA, ahA, alA, cnA);
            if (!haveB) break;
            haveA = next_mark(gA, wA);
            if (haveA) load_pair(gA, ahA, alA, cnA);
            refine_pair(gB, wB, ahB, alB, cnB);
        }
    }

    if (lane == 0 && blk_stats) blk_stats[(size_t)blockIdx.x * 4 + w] = make_uint4(st_needed, st_total, n_hh, n_ref);

    // ---- epilogue: merge the half-waves, settle or list (filter.hip) -------------------------------------------
    auto slot_id = [&](unsigned slot) {
        return slot == NONE ? NONE
                            : reinterpret_cast<const unsigned*>(img + (size_t)(slot >> 5) * GB + misc_off(D) + 128)[slot & 31];
    };
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int T = w + 4 * b;
        const float o1 = __shfl_xor(b1[b], 32), o2 = __shfl_xor(b2[b], 32), o3 = __shfl_xor(b3[b], 32);
        const unsigned oj1 = (unsigned)__shfl_xor((int)i1[b], 32), oj2 = (unsigned)__shfl_xor((int)i2[b], 32);
        float n1 = b1[b], n2 = b2[b], n3 = b3[b];
        unsigned nj1 = i1[b], nj2 = i2[b];
        insert3(o1, oj1, n1, n2, n3, nj1, nj2);
        insert3(o2, oj2, n1, n2, n3, nj1, nj2);
        insert3(o3, NONE, n1, n2, n3, nj1, nj2);
        const long pos = pos0 + 32 * T + j;
        const bool mine = h == 0 && pos < n;
        const float* st = reinterpret_cast<const float*>(lds + L::stash) + (size_t)T * 128;
        const float tau_b2 = st[j];
        const unsigned row_b = reinterpret_cast<const unsigned*>(st)[32 + j];
        const float gbd_b = st[64 + j];
        const unsigned hint_b = reinterpret_cast<const unsigned*>(st)[96 + j];
        const bool sane = nj1 != NONE && tau_b2 >= 0.0f && n1 > -__builtin_inff();
        const bool unique = sane && (n2 - n1) > tau_b2;
        const bool pair = sane && nj2 != NONE && (n3 - n1) > tau_b2;   // exactly two candidates within reach
        if (mine) {
            const unsigned id = slot_id(nj1);
            ids[row_b] = id == NONE ? -1L : (long)id;
            if (fp.dist_out) fp.dist_out[row_b] = (id != NONE && id == hint_b) ? gbd_b : __uint_as_float(DIST_TODO);
            if (approx_out) {   // test hook: approximate distance of the winner and the gap to the runner-up
                approx_out[2 * (size_t)row_b] = n1;
                approx_out[2 * (size_t)row_b + 1] = n2 - n1;
            }
        }
        const unsigned long long flagged = __builtin_amdgcn_ballot_w64(mine && !unique);
        if (flagged != 0) {
            const unsigned sub = (unsigned)(blockIdx.x * 4 + w) & (AMB_SUBLISTS - 1);
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

}  // namespace

int at_filter_sweep_wg(at_ctx* ctx, const float* x, int64_t n, int d, const unsigned char* img, int ng,
                       const uint32_t* order, unsigned* misc, float tau_a, float tau_b, float rho_a, float rho_b,
                       int screen, int64_t* ids, uint32_t* amb_list, uint32_t* amb_aux, float* approx_out,
                       const atf::FusedPrepass& fp, uint4* blk_stats, unsigned amb_cap, int tiles_per_wave,
                       hipStream_t stream) {
    AT_REQUIRE(d == 64 && ng <= 512 && (tiles_per_wave == 1 || tiles_per_wave == 2), "at_filter_sweep_wg: unsupported shape");
    long* idl = reinterpret_cast<long*>(ids);
    if (tiles_per_wave == 2) {
        constexpr size_t lds = Carve<64, 2>::total;
        const int rc = at_raise_lds(ctx, reinterpret_cast<const void*>(&assign_f16filter_wg_kernel<64, 2>), lds);
        if (rc) return rc;
        AT_LAUNCH((assign_f16filter_wg_kernel<64, 2>), dim3((unsigned)((n + 255) / 256)), dim3(WGT), lds, stream, x, (long)n, img,
                  ng, order, misc, tau_a, tau_b, rho_a, rho_b, screen, idl, amb_list, amb_aux, approx_out, fp, blk_stats, amb_cap);
    } else {
        constexpr size_t lds = Carve<64, 1>::total;
        const int rc = at_raise_lds(ctx, reinterpret_cast<const void*>(&assign_f16filter_wg_kernel<64, 1>), lds);
        if (rc) return rc;
        AT_LAUNCH((assign_f16filter_wg_kernel<64, 1>), dim3((unsigned)((n + 127) / 128)), dim3(WGT), lds, stream, x, (long)n, img,
                  ng, order, misc, tau_a, tau_b, rho_a, rho_b, screen, idl, amb_list, amb_aux, approx_out, fp, blk_stats, amb_cap);
    }
    return AT_OK;
}
