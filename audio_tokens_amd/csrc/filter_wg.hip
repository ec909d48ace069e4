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
//     (global_load_lds_dwordx4, every wave issues one 1 KiB piece), in a ring of RING stages: the stage being
//     multiplied and RING - 2 in flight behind a counted s_waitcnt vmcnt.  The depth is what the L2 -> LDS path needs:
//     it delivers ~70 GB/s per CU only with ~70 KiB in flight per CU (MI355X_MICROARCH.md, gather into LDS); with a
//     ring of three (26 KiB in flight per CU) the first version of this kernel ran at 1.3 ms where the one-wave
//     kernel took 0.95.  The lo parts of the rows therefore do NOT sit in LDS (32 KiB per workgroup) as in
//     filter.hip: the few tiles that have marked pairs re-read their rows (they are still in L2 / the Infinity
//     Cache) and split them again;
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
    // three 256-row workgroups (NB = 2) or four 128-row workgroups (NB = 1) per CU: ~46 / ~31 KiB each;
    // three batches of visits in the ring
    static constexpr int RING = NB == 2 ? 9 : 6;
    static constexpr size_t ring = 0;
    static constexpr size_t need = ring + RING * STAGE;                // need masks [NT][8] u64
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
    constexpr int RING = L::RING;
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
#ifdef AT_WG_STAMPS   // diagnostic build: where a wave's cycles go (replaces the statistics record; results unchanged)
    unsigned long long tst[6];
    unsigned long long t_wait = 0, t_comp = 0;
#define AT_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); tst[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
    AT_STAMP(0);
#else
#define AT_STAMP(i) do {} while (0)
#endif

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
        // fp16 split of the row: the hi parts stay in registers (the MFMA B operand); the lo parts are formed again
        // by the refinement, for the tiles that need them
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                xh[b][s][e] = (_Float16)xu[s][e];
                xh[b][s][4 + e] = (_Float16)xv[s][e];
            }
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
    AT_STAMP(1);
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
    // Visits go in batches of BATCH with ONE barrier per batch: the four waves' work differs from visit to visit (a
    // visit has ~2.7 of the 8 tiles active) and a barrier per visit made every visit cost its slowest wave -- 1.3 ms
    // for the sweep against 0.95 for round 2's kernel.  The ring holds three batches: the one being multiplied and
    // two in flight (requested when the batch two before them has been left by every wave).
    constexpr int BATCH = RING / 3;
    constexpr int PIECES = NS / 4;          // LDS-DMA instructions per wave and stage (wave 0: one more)
    const int nbatch = (cnt + BATCH - 1) / BATCH;
    auto issue_batch = [&](int k) {
#pragma unroll
        for (int v = 0; v < BATCH; v++) {
            const int i = k * BATCH + v;
            if (i < cnt) stage((k % 3) * BATCH + v, (unsigned)__builtin_amdgcn_readfirstlane((int)glist[i]));
        }
    };
    AT_STAMP(2);
    if (nbatch > 0) issue_batch(0);
    if (nbatch > 1) issue_batch(1);
    unsigned n_hh = 0;
    for (int k = 0; k < nbatch; k++) {
        uint32_t ve[BATCH];
#pragma unroll
        for (int v = 0; v < BATCH; v++) ve[v] = k * BATCH + v < cnt ? glist[k * BATCH + v] : 0u;
        // batch k has landed when all but the pieces of batch k + 1 have -- if that one is complete (only the last
        // batch can be short: in front of it everything is waited for)
#ifdef AT_WG_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        if (k + 2 < nbatch) {
            if (w == 0) wait_vm_and_barrier<BATCH * (PIECES + 1)>();
            else wait_vm_and_barrier<BATCH * PIECES>();
        } else {
            wait_vm_and_barrier<0>();
        }
#ifdef AT_WG_STAMPS
        const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
        t_wait += tw1 - tw0;
#endif
        if (k + 2 < nbatch) issue_batch(k + 2);     // into the slots batch k - 1 has left
#ifdef AT_WG_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tw2 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int v = 0; v < BATCH; v++) {
            if (k * BATCH + v >= cnt) break;
            const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)ve[v]);
            const unsigned mine = ((e >> (9 + w)) & 1u) | (NB == 2 ? (((e >> (13 + w)) & 1u) << 1) : 0u);
            if (mine == 0) continue;
            const unsigned char* slot = ring + (size_t)((k % 3) * BATCH + v) * L::STAGE;
            half8 ah[NS];
            f32x4 cnv[4];
#pragma unroll
            for (int s = 0; s < NS; s++) ah[s] = reinterpret_cast<const half8*>(slot)[s * 64 + lane];
            const float* cnp = reinterpret_cast<const float*>(slot + (size_t)NS * 1024);
#pragma unroll
            for (int q = 0; q < 4; q++) cnv[q] = *reinterpret_cast<const f32x4*>(cnp + 8 * q + 4 * h);
            const unsigned g = e & 511u;
            auto hihi = [&](int b) {
                f32x16 a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < NS; s++) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[b][s], a, 0, 0, 0);
                return a;
            };
            auto screen_tile = [&](int b, const f32x16& a) {
                float P[16];
#pragma unroll
                for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cnv[r >> 2][r & 3]);
                const bool pass = __builtin_amdgcn_ballot_w64(min16(P) < thr[b]) != 0;
                n_hh++;
                if (pass && lane == 0) {
                    uint32_t* rb = reinterpret_cast<uint32_t*>(lds + L::refb) + (w + 4 * b) * 16;
                    rb[g >> 5] |= 1u << (g & 31);
                }
            };
            if constexpr (NB == 2) {
                if (mine == 3u) {   // both tiles: the second tile's MFMAs run while the first tile is screened
                    const f32x16 a0 = hihi(0);
                    const f32x16 a1 = hihi(1);
                    screen_tile(0, a0);
                    screen_tile(1, a1);
                } else if (mine == 1u) {
                    screen_tile(0, hihi(0));
                } else {
                    screen_tile(1, hihi(1));
                }
            } else {
                screen_tile(0, hihi(0));
            }
        }
#ifdef AT_WG_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        t_comp += __builtin_amdgcn_s_memtime() - tw2;
#endif
    }
    AT_STAMP(3);

    // ---- the marked pairs: all three products, the triple as filter.hip keeps it -------------------------------
    // Tile by tile (a tile's lo parts are formed when its first mark turns up and live in registers until its last):
    // ONE fragment set; the next pair's fragments are requested as soon as this pair's MFMAs have been issued (they
    // have read their operands by then) and arrive while the MFMAs run and the triple is updated.
    float b1[NB], b2[NB], b3[NB];
    unsigned i1[NB], i2[NB];
    unsigned n_ref = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        b1[b] = b2[b] = b3[b] = cap0[b];
        i1[b] = i2[b] = NONE;
        const int T = w + 4 * b;
        const uint32_t* rb = reinterpret_cast<const uint32_t*>(lds + L::refb) + T * 16;   // (this wave's own stores)
        int wi = 0;
        uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb[0]);
        auto next_mark = [&](unsigned& g) -> bool {   // wave-uniform
            for (;;) {
                if (m != 0) {
                    const int bit = __builtin_ctz(m);
                    g = (unsigned)(wi * 32 + bit);
                    m &= m - 1u;
                    return true;
                }
                if (++wi >= 16 || wi * 32 >= ng) return false;
                m = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb[wi]);
            }
        };
        unsigned gC = 0, gN = 0;
        bool have = next_mark(gC);
        if (!have) continue;
        auto load_frags = [&](unsigned g, half8 (&ah)[NS], half8 (&al)[NS]) {
            const unsigned char* base = img + (size_t)g * GB;
            const half8* fh = reinterpret_cast<const half8*>(base);
            const half8* fl = reinterpret_cast<const half8*>(base + lo_off(D));
#pragma unroll
            for (int s = 0; s < NS; s++) {
                ah[s] = fh[s * 64 + lane];
                al[s] = fl[s * 64 + lane];
            }
        };
        auto load_cn = [&](unsigned g, f32x4 (&cnv)[4]) {
            const float* cnp = reinterpret_cast<const float*>(img + (size_t)g * GB + misc_off(D));
#pragma unroll
            for (int q = 0; q < 4; q++) cnv[q] = *reinterpret_cast<const f32x4*>(cnp + 8 * q + 4 * h);
        };
        half8 ah[NS], al[NS];
        f32x4 cnC[4], cnN[4];
        load_frags(gC, ah, al);
        load_cn(gC, cnC);
        // the lo parts of this tile's rows: v - fp16(v), rounded to fp16 (the rows are re-read: L2 / Infinity Cache)
        half8 xl[NS];
        {
            const unsigned r = reinterpret_cast<const unsigned*>(lds + L::stash)[(size_t)T * 128 + 32 + j];
            const f32x4* p = reinterpret_cast<const f32x4*>(X + (size_t)r * D);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const f32x4 u = p[4 * s + 2 * h], v = p[4 * s + 2 * h + 1];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    xl[s][e] = (_Float16)(u[e] - (float)(_Float16)u[e]);
                    xl[s][4 + e] = (_Float16)(v[e] - (float)(_Float16)v[e]);
                }
            }
        }
        while (have) {
            // hi*hi first, as the walk formed it; then the two lo products
            f32x16 a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NS; s++) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xh[b][s], a, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], xh[b][s], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], xl[s], a, 0, 0, 0);
            }
            const bool more = next_mark(gN);
            if (more) {
                load_frags(gN, ah, al);
                load_cn(gN, cnN);
            }
            float P[16];
#pragma unroll
            for (int r = 0; r < 16; r++) P[r] = __builtin_fmaf(-2.0f, a[r], cnC[r >> 2][r & 3]);
            n_ref++;
            if (__builtin_amdgcn_ballot_w64(min16(P) < b3[b]) != 0) {
                float v1 = b1[b], v2 = b2[b], v3 = b3[b];
                unsigned j1 = i1[b], j2 = i2[b];
                const unsigned base = gC * 32u + 4u * h;
#pragma unroll
                for (int r = 0; r < 16; r++) insert3(P[r], base + (unsigned)((r & 3) + 8 * (r >> 2)), v1, v2, v3, j1, j2);
                b1[b] = v1; b2[b] = v2; b3[b] = v3;
                i1[b] = j1; i2[b] = j2;
            }
            have = more;
            gC = gN;
#pragma unroll
            for (int q = 0; q < 4; q++) cnC[q] = cnN[q];
        }
    }

    AT_STAMP(4);
#ifdef AT_WG_STAMPS
    // record (units of 16 cycles): prologue + list, waits of the walk (barrier + vmcnt), the walk's work (DMA issue excluded), refinement
    if (lane == 0 && blk_stats)
        blk_stats[(size_t)blockIdx.x * 4 + w] = make_uint4((unsigned)((tst[2] - tst[0]) >> 4), (unsigned)(t_wait >> 4), (unsigned)(t_comp >> 4),
                                                           (unsigned)((tst[4] - tst[3]) >> 4));
    (void)st_needed; (void)st_total; (void)n_hh; (void)n_ref;
#else
    if (lane == 0 && blk_stats) blk_stats[(size_t)blockIdx.x * 4 + w] = make_uint4(st_needed, st_total, n_hh, n_ref);
#endif

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
