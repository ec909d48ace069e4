// assign.hip -- nearest centroid under squared L2 on gfx950 (at_assign_f32).
//
// Replaces faiss.IndexFlatL2.search(x, 1) at processors/spec_tokenizer.py:77 and inside
// faiss.Kmeans.train (processors/cluster_creator.py:54-56) of danavery/audio-tokens.
//
// Arithmetic contract (identical to oracle/oracle.c orc_assign, bit for bit):
//   ip(i,j)  = fmaf chain over the feature index, ascending, from +0   (v_mfma_f32_32x32x2_f32)
//   xn(i), cn(j) likewise with both operands equal
//   dis(i,j) = max(0, (xn + cn) - 2*ip)            one rounding for the add, one for the fma
//   ids[i]   = lowest j with minimal dis, dist[i] = that dis
//
// Data flow.  The centroid table is re-laid once per call into tiles of 32*NA centroids whose rows
// are already in LDS order (k split even/odd per 8 features so that one ds_read_b128 feeds four
// consecutive MFMA k-steps, 16-byte chunks XOR-swizzled by row so the reads are conflict free),
// each tile followed by its squared norms.  A workgroup of 4 waves streams every tile through a
// double-buffered LDS copy while each wave keeps its 32*NB rows of x in registers for the whole
// sweep (B operand: lane l holds x[row l&31][k = 2s + (l>>5)]); the 32x32 accumulator puts the x
// row on the lane and 16 centroids in the registers, so the running arg-min is lane-local and the
// two half-waves are merged once at the end.  The arg-min of a finished accumulator is scheduled
// between the MFMAs of the next one, including across the tile boundary (the last accumulator of
// a tile is carried into the next loop iteration).  HBM traffic: x once (4d B/row) + 12 B/row out;
// the centroid image (k*d*4 B) is L2/Infinity-Cache resident.  Bound: fp32 MFMA (2*d*k flop/row).
#include <cstdlib>

#include "at_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WG = 256;    // threads per workgroup (4 waves)
constexpr int CN_PAD = 256;  // floats reserved per tile for the squared norms (whole 1 KiB pieces)

__host__ __device__ constexpr int tile_rows(int na) { return 32 * na; }
__host__ __device__ constexpr int tile_floats(int dp, int na) { return tile_rows(na) * dp + CN_PAD; }

// ---------------------------------------------------------------------------------------------
// Centroid image: tile t holds rows t*R .. t*R+R-1 (R = 32*na).  Row r, 16-byte chunk pc
// (physical) holds logical chunk lc = pc ^ (r & 15); lc = 2q + h holds features 8q + {0,2,4,6} + h.
// Features >= d and rows >= k are zero; |c|^2 of a row >= k is +inf so it can never win.
__global__ void __launch_bounds__(WG) prep_centroids_kernel(const float* __restrict__ c, int k, int d,
                                                            int dp, int na, float* __restrict__ img) {
    const int t = blockIdx.x;
    const int R = tile_rows(na);
    float* out = img + (size_t)t * tile_floats(dp, na);
    const int chunks_per_row = dp / 4;
    for (int e = threadIdx.x; e < R * chunks_per_row; e += WG) {
        const int r = e / chunks_per_row, pc = e % chunks_per_row;
        const int lc = pc ^ (r & 15);
        const int q = lc >> 1, h = lc & 1;
        const int row = t * R + r;
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int f = 8 * q + 2 * u + h;
            v[u] = (row < k && f < d) ? c[(size_t)row * d + f] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)r * dp + pc * 4) = v;
    }
    for (int r = threadIdx.x; r < CN_PAD; r += WG) {
        const int row = t * R + r;
        float nrm = __builtin_inff();
        if (r < R && row < k) {
            nrm = 0.0f;
            for (int f = 0; f < d; f++) {
                const float v = c[(size_t)row * d + f];
                nrm = __builtin_fmaf(v, v, nrm);
            }
        }
        out[R * dp + r] = nrm;
    }
}

// One arg-min step with the compare mask kept in VCC (three VALU ops, no SGPR pair to carry):
//   keep = !(dis < bd);  bd = keep ? bd : dis;  br = keep ? br : R
template <int R>
__device__ __forceinline__ void argmin_step(float& bd, unsigned& br, float dis) {
    asm("v_cmp_nlt_f32 vcc, %2, %0\n\t"
        "v_cndmask_b32 %0, %2, %0, vcc\n\t"
        "v_cndmask_b32 %1, %3, %1, vcc"
        : "+v"(bd), "+v"(br)
        : "v"(dis), "n"(R)
        : "vcc");
}

// dis = max(0, (xn + cn) - 2*ip) for accumulator register R, then the arg-min step.
template <int R>
__device__ __forceinline__ void epilogue_from(float& bd, unsigned& br, const f32x16& acc, float xn,
                                              const f32x4 (&cnv)[4]) {
    if constexpr (R < 16) {
        const float t = xn + cnv[R >> 2][R & 3];
        float dis = __builtin_fmaf(-2.0f, acc[R], t);
        dis = __builtin_fmaxf(dis, 0.0f);
        argmin_step<R>(bd, br, dis);
        epilogue_from<R + 1>(bd, br, acc, xn, cnv);
    }
}

// Running arg-min over one finished accumulator (this lane's 16 centroids, ascending index).
__device__ __forceinline__ void epilogue16(float& bestd, unsigned& bestc, const f32x16& acc, float xn,
                                           const f32x4 (&cnv)[4], unsigned codebase) {
    float bd = bestd;
    unsigned br = 16u;  // 16 = "no improvement from this accumulator"
    epilogue_from<0>(bd, br, acc, xn, cnv);
    bestd = bd;
    bestc = br != 16u ? (codebase | br) : bestc;
}

// Speculative form: the 16 unclamped distances and their minimum cost 2.5 VALU per element; the
// exact 4-op update (clamp, compare, two selects) runs only when some lane of the wave improves
// (a clamped distance can beat bestd only if the unclamped one does, since bestd >= 0).
template <int R>
__device__ __forceinline__ void update_from(float& bd, unsigned& br, const float (&dis)[16]) {
    if constexpr (R < 16) {
        argmin_step<R>(bd, br, __builtin_fmaxf(dis[R], 0.0f));
        update_from<R + 1>(bd, br, dis);
    }
}

__device__ __forceinline__ void epilogue16_spec(float& bestd, unsigned& bestc, const f32x16& acc, float xn,
                                                const f32x4 (&cnv)[4], unsigned codebase) {
    float dis[16];
#pragma unroll
    for (int r = 0; r < 16; r++) dis[r] = __builtin_fmaf(-2.0f, acc[r], xn + cnv[r >> 2][r & 3]);
    float m = __builtin_fminf(dis[0], dis[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) m = __builtin_fminf(__builtin_fminf(m, dis[r]), dis[r + 1]);
    if (__builtin_amdgcn_ballot_w64(m < bestd) != 0) {
        float bd = bestd;
        unsigned br = 16u;
        update_from<0>(bd, br, dis);
        bestd = bd;
        bestc = br != 16u ? (codebase | br) : bestc;
    }
}

// 1 KiB of global memory -> LDS without passing through registers (global_load_lds_dwordx4):
// lane l's 16 bytes land at lds_wave_base + 16*l.
__device__ __forceinline__ void dma_1k(const float* gsrc_lane, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc_lane,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------
template <int D, int NB, int NA, bool DMA, int WAVES_PER_SIMD, bool SPEC = false>
__global__ void __launch_bounds__(WG, WAVES_PER_SIMD)
assign_mfma_kernel(const float* __restrict__ X, long n, const float* __restrict__ img, int ntiles,
                   long* __restrict__ ids, float* __restrict__ dist) {
    constexpr int R = tile_rows(NA);
    constexpr int TILE_F = tile_floats(D, NA);
    constexpr int NV = TILE_F / 4;               // float4 per tile image
    constexpr int VPT = (NV + WG - 1) / WG;      // float4 staged per thread (register staging)
    constexpr int PIECES = TILE_F / 256;         // 1 KiB pieces per tile (DMA staging)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * TILE_F floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;   // x row within a 32-row tile == accumulator column
    const int h = lane >> 5;   // which k of each MFMA k-pair this lane feeds
    const long row0 = ((long)blockIdx.x * 4 + wave) * (32 * NB);

    // ---- x rows -> registers (B operand), |x|^2 as the ascending fmaf chain ----
    float xr[NB][D / 2];
    float xn[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        long r = row0 + 32 * b + j;
        if (r >= n) r = n - 1;
        const f32x4* p = reinterpret_cast<const f32x4*>(X + r * D);
        float nrm = 0.0f;
#pragma unroll
        for (int q = 0; q < D / 8; q++) {
            const f32x4 u = p[2 * q], v = p[2 * q + 1];
            nrm = __builtin_fmaf(u[0], u[0], nrm);
            nrm = __builtin_fmaf(u[1], u[1], nrm);
            nrm = __builtin_fmaf(u[2], u[2], nrm);
            nrm = __builtin_fmaf(u[3], u[3], nrm);
            nrm = __builtin_fmaf(v[0], v[0], nrm);
            nrm = __builtin_fmaf(v[1], v[1], nrm);
            nrm = __builtin_fmaf(v[2], v[2], nrm);
            nrm = __builtin_fmaf(v[3], v[3], nrm);
            xr[b][4 * q + 0] = h ? u[1] : u[0];
            xr[b][4 * q + 1] = h ? u[3] : u[2];
            xr[b][4 * q + 2] = h ? v[1] : v[0];
            xr[b][4 * q + 3] = h ? v[3] : v[2];
        }
        xn[b] = nrm;
    }

    float bestd[NB];
    unsigned bestc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        bestd[b] = __builtin_inff();
        bestc[b] = 0xffffffffu;
    }

    // ---- stage tile 0 ----
    auto stage_dma = [&](int tile, float* dst) {
        const float* src = img + (size_t)tile * TILE_F;
#pragma unroll
        for (int p0 = 0; p0 < PIECES; p0 += 4) {
            const int p = p0 + wave;
            if (p < PIECES) dma_1k(src + p * 256 + lane * 4, dst + p * 256);
        }
    };
    f32x4 pf[DMA ? 1 : VPT];
    if constexpr (DMA) {
        stage_dma(0, smem);
    } else {
        const f32x4* src = reinterpret_cast<const f32x4*>(img);
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            const int e = tid + v * WG;
            if (e < NV) reinterpret_cast<f32x4*>(smem)[e] = src[e];
        }
    }
    __syncthreads();

    // the last accumulator of a tile is finished inside the next iteration
    f32x16 accP = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 cnP[4];
#pragma unroll
    for (int g = 0; g < 4; g++) cnP[g] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    unsigned codeP = 0u;

    const int swz = j & 15;
    for (int ct = 0; ct < ntiles; ct++) {
        const float* cur = smem + (ct & 1) * TILE_F;
        const bool more = ct + 1 < ntiles;
        if (more) {
            if constexpr (DMA) {
                stage_dma(ct + 1, smem + ((ct + 1) & 1) * TILE_F);
            } else {
                const f32x4* src = reinterpret_cast<const f32x4*>(img + (size_t)(ct + 1) * TILE_F);
#pragma unroll
                for (int v = 0; v < VPT; v++) {
                    const int e = tid + v * WG;
                    if (e < NV) pf[v] = src[e];
                }
            }
        }

#pragma unroll
        for (int a = 0; a < NA; a++) {
            // |c|^2 of this lane's 16 accumulator rows: rows a*32 + 8g + 4h + (0..3)
            f32x4 cnv[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                cnv[g] = *reinterpret_cast<const f32x4*>(cur + R * D + a * 32 + 8 * g + 4 * h);
            const float* arow = cur + (a * 32 + j) * D;
            const unsigned codebase = (unsigned)(ct * NA + a) * 16u;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < D / 8; q++) {
                    const int pc = (2 * q + h) ^ swz;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow + pc * 4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], xr[b][4 * q + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], xr[b][4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], xr[b][4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], xr[b][4 * q + 3], acc, 0, 0, 0);
                }
                if constexpr (SPEC) {
                    epilogue16_spec(bestd[b], bestc[b], acc, xn[b], cnv, codebase);
                    continue;
                }
                if (a == 0 && b == 0)  // previous tile's last accumulator (no-op on the first tile)
                    epilogue16(bestd[NB - 1], bestc[NB - 1], accP, xn[NB - 1], cnP, codeP);
                if (a == NA - 1 && b == NB - 1) {
                    accP = acc;
#pragma unroll
                    for (int g = 0; g < 4; g++) cnP[g] = cnv[g];
                    codeP = codebase;
                } else {
                    epilogue16(bestd[b], bestc[b], acc, xn[b], cnv, codebase);
                }
            }
        }

        if constexpr (!DMA) {
            if (more) {
                f32x4* dst = reinterpret_cast<f32x4*>(smem + ((ct + 1) & 1) * TILE_F);
#pragma unroll
                for (int v = 0; v < VPT; v++) {
                    const int e = tid + v * WG;
                    if (e < NV) dst[e] = pf[v];
                }
            }
        }
        __syncthreads();  // (with LDS-DMA in flight hipcc makes this vmcnt(0) + s_barrier)
    }
    epilogue16(bestd[NB - 1], bestc[NB - 1], accP, xn[NB - 1], cnP, codeP);

    // ---- merge the two half-waves (rows 4h + ... of every 8-row group) and store ----
#pragma unroll
    for (int b = 0; b < NB; b++) {
        int idx = -1;
        if (bestc[b] != 0xffffffffu) {
            const unsigned r = bestc[b] & 15u;
            idx = (int)((bestc[b] >> 4) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (unsigned)h);
        }
        const float od = __shfl_xor(bestd[b], 32);
        const int oi = __shfl_xor(idx, 32);
        float fd = bestd[b];
        int fi = idx;
        if (od < fd || (od == fd && (unsigned)oi < (unsigned)fi)) {
            fd = od;
            fi = oi;
        }
        const long r = row0 + 32 * b + j;
        if (h == 0 && r < n) {
            ids[r] = (long)fi;
            if (dist) dist[r] = fd;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Hinted sweep (Lloyd iterations after the first).  Every row arrives with a guess -- its
// assignment in the previous iteration -- and rows are visited in an order that groups equal
// guesses (the member-list order the centroid update has just built).  The guess's distance is
// evaluated exactly up front (the same ascending fmaf chain, on the VALU) and becomes the running
// best, so during the sweep an accumulator can matter only if its minimum reaches that best:
// per accumulator the wave computes 16 unclamped distances and their minimum (2.5 VALU per
// element instead of 6) and enters the exact update -- lexicographic (distance, index), so that the
// answer is the brute-force arg-min with lowest-index ties whatever the guess was -- only when some
// lane needs it.  With grouped guesses that is the guess's own tile plus the rows that really move.
template <int D, int NB, int NA>
__global__ void __launch_bounds__(WG, 2)
assign_mfma_hinted_kernel(const float* __restrict__ X, long n, const float* __restrict__ C, int k,
                          const float* __restrict__ img, int ntiles, const uint32_t* __restrict__ order,
                          const long* __restrict__ hint, const uint32_t* __restrict__ hint_sorted,
                          long* __restrict__ ids, float* __restrict__ dist) {
    constexpr int R = tile_rows(NA);
    constexpr int TILE_F = tile_floats(D, NA);
    constexpr int PIECES = TILE_F / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * TILE_F floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const long pos0 = ((long)blockIdx.x * 4 + wave) * (32 * NB);

    float xr[NB][D / 2];
    float xn[NB];
    float bestd[NB];
    unsigned besti[NB];
    long rowid[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        long pos = pos0 + 32 * b + j;
        if (pos >= n) pos = n - 1;
        const long r = order ? (long)order[pos] : pos;
        rowid[b] = r;
        const f32x4* p = reinterpret_cast<const f32x4*>(X + r * D);
        // guesses listed in visiting order are one coalesced read; per-row guesses are a gather
        const long g = hint_sorted ? (long)hint_sorted[pos] : (hint ? hint[r] : -1);
        const bool has = g >= 0 && g < k;
        const f32x4* pc = reinterpret_cast<const f32x4*>(C + (has ? g : 0) * D);
        float nrm = 0.0f, cnn = 0.0f, ip = 0.0f;
        // exact distance to the guess: three ascending fmaf chains (rolled loop: keeps the register
        // footprint of this one-off prologue small)
#pragma clang loop unroll(disable)
        for (int q = 0; q < D / 4; q++) {
            const f32x4 u = p[q], cu = pc[q];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                nrm = __builtin_fmaf(u[e], u[e], nrm);
                cnn = __builtin_fmaf(cu[e], cu[e], cnn);
                ip = __builtin_fmaf(cu[e], u[e], ip);
            }
        }
#pragma unroll
        for (int q = 0; q < D / 8; q++) {
            const f32x4 u = p[2 * q], v = p[2 * q + 1];
            xr[b][4 * q + 0] = h ? u[1] : u[0];
            xr[b][4 * q + 1] = h ? u[3] : u[2];
            xr[b][4 * q + 2] = h ? v[1] : v[0];
            xr[b][4 * q + 3] = h ? v[3] : v[2];
        }
        xn[b] = nrm;
        const float dh = __builtin_fmaxf(__builtin_fmaf(-2.0f, ip, nrm + cnn), 0.0f);
        bestd[b] = has ? dh : __builtin_inff();
        besti[b] = has ? (unsigned)g : 0xffffffffu;
    }

    auto stage_dma = [&](int tile, float* dst) {
        const float* src = img + (size_t)tile * TILE_F;
#pragma unroll
        for (int p0 = 0; p0 < PIECES; p0 += 4) {
            const int p = p0 + wave;
            if (p < PIECES) dma_1k(src + p * 256 + lane * 4, dst + p * 256);
        }
    };
    stage_dma(0, smem);
    __syncthreads();

    const int swz = j & 15;
    for (int ct = 0; ct < ntiles; ct++) {
        const float* cur = smem + (ct & 1) * TILE_F;
        if (ct + 1 < ntiles) stage_dma(ct + 1, smem + ((ct + 1) & 1) * TILE_F);
#pragma unroll
        for (int a = 0; a < NA; a++) {
            f32x4 cnv[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                cnv[g] = *reinterpret_cast<const f32x4*>(cur + R * D + a * 32 + 8 * g + 4 * h);
            const float* arow = cur + (a * 32 + j) * D;
            const unsigned jobbase = (unsigned)(ct * NA + a) * 32u + 4u * (unsigned)h;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < D / 8; q++) {
                    const int pcx = (2 * q + h) ^ swz;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow + pcx * 4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], xr[b][4 * q + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], xr[b][4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], xr[b][4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], xr[b][4 * q + 3], acc, 0, 0, 0);
                }
                // minimum of the 16 unclamped distances (nothing kept: the rare exact update below
                // recomputes them from the accumulator)
                float m = __builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float d0 = __builtin_fmaf(-2.0f, acc[r], xn[b] + cnv[r >> 2][r & 3]);
                    const float d1 = __builtin_fmaf(-2.0f, acc[r + 1], xn[b] + cnv[(r + 1) >> 2][(r + 1) & 3]);
                    m = __builtin_fminf(__builtin_fminf(m, d0), d1);
                }
                // a clamped distance can tie or beat bestd (>= 0) only if the unclamped one does
                if (__builtin_amdgcn_ballot_w64(m <= bestd[b]) != 0) {
                    float bd = bestd[b];
                    unsigned bi = besti[b];
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float dd = __builtin_fmaxf(
                            __builtin_fmaf(-2.0f, acc[r], xn[b] + cnv[r >> 2][r & 3]), 0.0f);
                        const unsigned idx = jobbase + (unsigned)((r & 3) + 8 * (r >> 2));
                        const bool better = dd < bd || (dd == bd && idx < bi);
                        bd = better ? dd : bd;
                        bi = better ? idx : bi;
                    }
                    bestd[b] = bd;
                    besti[b] = bi;
                }
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int b = 0; b < NB; b++) {
        const float od = __shfl_xor(bestd[b], 32);
        const unsigned oi = (unsigned)__shfl_xor((int)besti[b], 32);
        float fd = bestd[b];
        unsigned fi = besti[b];
        if (od < fd || (od == fd && oi < fi)) {
            fd = od;
            fi = oi;
        }
        const long pos = pos0 + 32 * b + j;
        if (h == 0 && pos < n) {
            ids[rowid[b]] = fi == 0xffffffffu ? -1L : (long)fi;
            if (dist) dist[rowid[b]] = fd;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pruned sweep (Lloyd iterations after the first; see prune.hip for the bound).  Same structure as
// the hinted sweep, but (1) the centroid image is in spatially grouped order (a table in the tile's
// pad maps image rows back to centroid indices for the lexicographic update), (2) every 32-row tile
// comes with a bit mask over the 32-centroid groups: an accumulator whose bit is clear is not
// computed at all, and a 128-centroid tile that no wave of the workgroup needs is not even staged.
// The running best starts from the exact distance to the guess (computed by the pre-pass).
__global__ void __launch_bounds__(WG) prep_centroids_perm_kernel(const float* __restrict__ c, int k, int d,
                                                                 const int32_t* __restrict__ cperm, int kp,
                                                                 int na, float* __restrict__ img) {
    const int t = blockIdx.x;
    const int R = tile_rows(na);
    float* out = img + (size_t)t * tile_floats(d, na);
    const int chunks_per_row = d / 4;
    for (int e = threadIdx.x; e < R * chunks_per_row; e += WG) {
        const int r = e / chunks_per_row, pc = e % chunks_per_row;
        const int lc = pc ^ (r & 15);
        const int q = lc >> 1, h = lc & 1;
        const int slot = t * R + r;
        const int row = slot < kp ? cperm[slot] : -1;
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int f = 8 * q + 2 * u + h;
            v[u] = (row >= 0 && f < d) ? c[(size_t)row * d + f] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)r * d + pc * 4) = v;
    }
    for (int r = threadIdx.x; r < 128; r += WG) {
        const int slot = t * R + r;
        const int row = (r < R && slot < kp) ? cperm[slot] : -1;
        float nrm = __builtin_inff();
        if (row >= 0) {
            nrm = 0.0f;
            for (int f = 0; f < d; f++) {
                const float v = c[(size_t)row * d + f];
                nrm = __builtin_fmaf(v, v, nrm);
            }
        }
        out[R * d + r] = nrm;
        reinterpret_cast<unsigned*>(out)[R * d + 128 + r] = row >= 0 ? (unsigned)row : 0xffffffffu;
    }
}

// One wavefront per workgroup: a wave walks only the groups its own 32*NB rows need, staging one
// 32-centroid group (8 or 16 KiB + norms + index table) at a time through its private double buffer.
// No workgroup barrier anywhere: the LDS-DMA is issued and awaited (s_waitcnt vmcnt(0)) by the same
// wave that reads it.
constexpr int GROUP_PAD = 256;  // floats behind a group's rows: |c|^2 at [0,32), indices at [128,160)

template <int D, int NB>
__global__ void __launch_bounds__(64)
assign_mfma_pruned_kernel(const float* __restrict__ X, long n, const float* __restrict__ img, int ng,
                          const uint32_t* __restrict__ order, const uint32_t* __restrict__ hint_sorted,
                          const float* __restrict__ bd_in, const uint32_t* __restrict__ mask, int ngw,
                          long* __restrict__ ids, float* __restrict__ dist) {
    constexpr int GROUP_F = 32 * D + GROUP_PAD;
    constexpr int PIECES = GROUP_F / 256;
    constexpr int MAXW = 16;  // mask words kept in scalar registers (ng <= 512)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * GROUP_F floats

    const int lane = threadIdx.x;
    const int j = lane & 31;
    const int h = lane >> 5;
    const long pos0 = (long)blockIdx.x * (32 * NB);
    const long ntile32 = (n + 31) / 32;

    float xr[NB][D / 2];
    float xn[NB];
    float bestd[NB];
    unsigned besti[NB];
    long rowid[NB];
    uint32_t mw[NB][MAXW];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        long pos = pos0 + 32 * b + j;
        if (pos >= n) pos = n - 1;
        const long r = (long)order[pos];
        rowid[b] = r;
        const f32x4* p = reinterpret_cast<const f32x4*>(X + r * D);
        float nrm = 0.0f;
#pragma unroll
        for (int q = 0; q < D / 8; q++) {
            const f32x4 u = p[2 * q], v = p[2 * q + 1];
#pragma unroll
            for (int e = 0; e < 4; e++) nrm = __builtin_fmaf(u[e], u[e], nrm);
#pragma unroll
            for (int e = 0; e < 4; e++) nrm = __builtin_fmaf(v[e], v[e], nrm);
            xr[b][4 * q + 0] = h ? u[1] : u[0];
            xr[b][4 * q + 1] = h ? u[3] : u[2];
            xr[b][4 * q + 2] = h ? v[1] : v[0];
            xr[b][4 * q + 3] = h ? v[3] : v[2];
        }
        xn[b] = nrm;
        bestd[b] = bd_in[pos];
        besti[b] = bestd[b] < __builtin_inff() ? hint_sorted[pos] : 0xffffffffu;
        const long tile = pos0 / 32 + b;
#pragma unroll
        for (int w = 0; w < MAXW; w++) {
            uint32_t m = 0;
            if (w < ngw && tile < ntile32) m = mask[(size_t)tile * ngw + w];
            mw[b][w] = __builtin_amdgcn_readfirstlane(m);
        }
    }
    uint32_t any[MAXW];
#pragma unroll
    for (int w = 0; w < MAXW; w++) {
        any[w] = 0;
#pragma unroll
        for (int b = 0; b < NB; b++) any[w] |= mw[b][w];
    }
    auto word_of = [&](const uint32_t (&m)[MAXW], int w) -> uint32_t {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < MAXW; i++)
            if (i == w) v = m[i];
        return v;
    };
    auto next_group = [&](int from) {  // first needed group >= from, or ng
        int w = from >> 5;
        if (w >= ngw) return ng;
        uint32_t bits = word_of(any, w) & (0xffffffffu << (from & 31));
        while (bits == 0) {
            if (++w >= ngw) return ng;
            bits = word_of(any, w);
        }
        const int g = (w << 5) + __builtin_ctz(bits);
        return g < ng ? g : ng;
    };
    auto stage_dma = [&](int g, float* dst) {
        const float* src = img + (size_t)g * GROUP_F;
#pragma unroll
        for (int p = 0; p < PIECES; p++) dma_1k(src + p * 256 + lane * 4, dst + p * 256);
    };

    int g = next_group(0);
    int buf = 0;
    if (g < ng) stage_dma(g, smem);
    const int swz = j & 15;
    while (g < ng) {
        const float* cur = smem + buf * GROUP_F;
        // the group being consumed has landed once every DMA this wave issued so far is done; the
        // reads of the other buffer (previous group) were all consumed by that group's MFMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nxt = next_group(g + 1);
        if (nxt < ng) stage_dma(nxt, smem + (buf ^ 1) * GROUP_F);

        f32x4 cnv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cnv[q] = *reinterpret_cast<const f32x4*>(cur + 32 * D + 8 * q + 4 * h);
        const float* arow = cur + j * D;
        const unsigned* idxrow = reinterpret_cast<const unsigned*>(cur) + 32 * D + 128 + 4 * h;
        const uint32_t gbit = 1u << (g & 31);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if ((word_of(mw[b], g >> 5) & gbit) == 0u) continue;  // wave-uniform
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < D / 8; q++) {
                const int pcx = (2 * q + h) ^ swz;
                const f32x4 av = *reinterpret_cast<const f32x4*>(arow + pcx * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], xr[b][4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], xr[b][4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], xr[b][4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], xr[b][4 * q + 3], acc, 0, 0, 0);
            }
            float m = __builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float d0 = __builtin_fmaf(-2.0f, acc[r], xn[b] + cnv[r >> 2][r & 3]);
                const float d1 = __builtin_fmaf(-2.0f, acc[r + 1], xn[b] + cnv[(r + 1) >> 2][(r + 1) & 3]);
                m = __builtin_fminf(__builtin_fminf(m, d0), d1);
            }
            if (__builtin_amdgcn_ballot_w64(m <= bestd[b]) != 0) {
                float bd = bestd[b];
                unsigned bi = besti[b];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float dd = __builtin_fmaxf(__builtin_fmaf(-2.0f, acc[r], xn[b] + cnv[r >> 2][r & 3]), 0.0f);
                    const unsigned idx = idxrow[(r & 3) + 8 * (r >> 2)];
                    const bool better = dd < bd || (dd == bd && idx < bi);
                    bd = better ? dd : bd;
                    bi = better ? idx : bi;
                }
                bestd[b] = bd;
                besti[b] = bi;
            }
        }
        g = nxt;
        buf ^= 1;
    }

#pragma unroll
    for (int b = 0; b < NB; b++) {
        const float od = __shfl_xor(bestd[b], 32);
        const unsigned oi = (unsigned)__shfl_xor((int)besti[b], 32);
        float fd = bestd[b];
        unsigned fi = besti[b];
        if (od < fd || (od == fd && oi < fi)) {
            fd = od;
            fi = oi;
        }
        const long pos = pos0 + 32 * b + j;
        if (h == 0 && pos < n) {
            ids[rowid[b]] = fi == 0xffffffffu ? -1L : (long)fi;
            if (dist) dist[rowid[b]] = fd;
        }
    }
}

// Register-staged form of the pruned sweep: the A fragments of the NEXT needed group are loaded
// from the (L2-resident) image straight into registers while the current group's MFMAs run, so
// there is no LDS, no DMA wait and no barrier at all; two register sets alternate (the loop body is
// written out for both roles).
template <int D, int NB>
__global__ void __launch_bounds__(64, 2)
assign_mfma_pruned_reg_kernel(const float* __restrict__ X, long n, const float* __restrict__ img, int ng,
                              const uint32_t* __restrict__ order, const uint32_t* __restrict__ hint_sorted,
                              const float* __restrict__ bd_in, const uint32_t* __restrict__ mask, int ngw,
                              long* __restrict__ ids, float* __restrict__ dist) {
    constexpr int GROUP_F = 32 * D + GROUP_PAD;
    constexpr int NQ = D / 8;
    constexpr int MAXW = 16;

    const int lane = threadIdx.x;
    const int j = lane & 31;
    const int h = lane >> 5;
    const long pos0 = (long)blockIdx.x * (32 * NB);
    const long ntile32 = (n + 31) / 32;

    float xr[NB][D / 2];
    float xn[NB];
    float bestd[NB];
    unsigned besti[NB];
    long rowid[NB];
    uint32_t mw[NB][MAXW];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        long pos = pos0 + 32 * b + j;
        if (pos >= n) pos = n - 1;
        const long r = (long)order[pos];
        rowid[b] = r;
        const f32x4* p = reinterpret_cast<const f32x4*>(X + r * D);
        float nrm = 0.0f;
#pragma unroll
        for (int q = 0; q < D / 8; q++) {
            const f32x4 u = p[2 * q], v = p[2 * q + 1];
#pragma unroll
            for (int e = 0; e < 4; e++) nrm = __builtin_fmaf(u[e], u[e], nrm);
#pragma unroll
            for (int e = 0; e < 4; e++) nrm = __builtin_fmaf(v[e], v[e], nrm);
            xr[b][4 * q + 0] = h ? u[1] : u[0];
            xr[b][4 * q + 1] = h ? u[3] : u[2];
            xr[b][4 * q + 2] = h ? v[1] : v[0];
            xr[b][4 * q + 3] = h ? v[3] : v[2];
        }
        xn[b] = nrm;
        bestd[b] = bd_in[pos];
        besti[b] = bestd[b] < __builtin_inff() ? hint_sorted[pos] : 0xffffffffu;
        const long tile = pos0 / 32 + b;
#pragma unroll
        for (int w = 0; w < MAXW; w++) {
            uint32_t m = 0;
            if (w < ngw && tile < ntile32) m = mask[(size_t)tile * ngw + w];
            mw[b][w] = __builtin_amdgcn_readfirstlane(m);
        }
    }
    uint32_t any[MAXW];
#pragma unroll
    for (int w = 0; w < MAXW; w++) {
        any[w] = 0;
#pragma unroll
        for (int b = 0; b < NB; b++) any[w] |= mw[b][w];
    }
    auto word_of = [&](const uint32_t (&m)[MAXW], int w) -> uint32_t {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < MAXW; i++)
            if (i == w) v = m[i];
        return v;
    };
    auto next_group = [&](int from) {
        int w = from >> 5;
        if (w >= ngw) return ng;
        uint32_t bits = word_of(any, w) & (0xffffffffu << (from & 31));
        while (bits == 0) {
            if (++w >= ngw) return ng;
            bits = word_of(any, w);
        }
        const int g = (w << 5) + __builtin_ctz(bits);
        return g < ng ? g : ng;
    };

    const int swz = j & 15;
    auto load_group = [&](int g, f32x4 (&av)[NQ], f32x4 (&cn)[4]) {
        const float* base = img + (size_t)g * GROUP_F;
        const float* arow = base + j * D;
#pragma unroll
        for (int q = 0; q < NQ; q++) av[q] = *reinterpret_cast<const f32x4*>(arow + (((2 * q + h) ^ swz) << 2));
#pragma unroll
        for (int q = 0; q < 4; q++) cn[q] = *reinterpret_cast<const f32x4*>(base + 32 * D + 8 * q + 4 * h);
    };
    auto compute_group = [&](int g, const f32x4 (&av)[NQ], const f32x4 (&cnv)[4]) {
        const uint32_t gbit = 1u << (g & 31);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if ((word_of(mw[b], g >> 5) & gbit) == 0u) continue;  // wave-uniform
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][0], xr[b][4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][1], xr[b][4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][2], xr[b][4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][3], xr[b][4 * q + 3], acc, 0, 0, 0);
            }
            float m = __builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float d0 = __builtin_fmaf(-2.0f, acc[r], xn[b] + cnv[r >> 2][r & 3]);
                const float d1 = __builtin_fmaf(-2.0f, acc[r + 1], xn[b] + cnv[(r + 1) >> 2][(r + 1) & 3]);
                m = __builtin_fminf(__builtin_fminf(m, d0), d1);
            }
            if (__builtin_amdgcn_ballot_w64(m <= bestd[b]) != 0) {
                const unsigned* idxrow =
                    reinterpret_cast<const unsigned*>(img + (size_t)g * GROUP_F) + 32 * D + 128 + 4 * h;
                float bd = bestd[b];
                unsigned bi = besti[b];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float dd = __builtin_fmaxf(__builtin_fmaf(-2.0f, acc[r], xn[b] + cnv[r >> 2][r & 3]), 0.0f);
                    const unsigned idx = idxrow[(r & 3) + 8 * (r >> 2)];
                    const bool better = dd < bd || (dd == bd && idx < bi);
                    bd = better ? dd : bd;
                    bi = better ? idx : bi;
                }
                bestd[b] = bd;
                besti[b] = bi;
            }
        }
    };

    f32x4 avA[NQ], cnA[4], avB[NQ], cnB[4];
    int g = next_group(0);
    if (g < ng) load_group(g, avA, cnA);
    while (g < ng) {
        const int g1 = next_group(g + 1);
        if (g1 < ng) load_group(g1, avB, cnB);
        compute_group(g, avA, cnA);
        if (g1 >= ng) break;
        const int g2 = next_group(g1 + 1);
        if (g2 < ng) load_group(g2, avA, cnA);
        compute_group(g1, avB, cnB);
        g = g2;
    }

#pragma unroll
    for (int b = 0; b < NB; b++) {
        const float od = __shfl_xor(bestd[b], 32);
        const unsigned oi = (unsigned)__shfl_xor((int)besti[b], 32);
        float fd = bestd[b];
        unsigned fi = besti[b];
        if (od < fd || (od == fd && oi < fi)) {
            fd = od;
            fi = oi;
        }
        const long pos = pos0 + 32 * b + j;
        if (h == 0 && pos < n) {
            ids[rowid[b]] = fi == 0xffffffffu ? -1L : (long)fi;
            if (dist) dist[rowid[b]] = fd;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Any d that is a multiple of 4 (n_mels other than 64/128, use_convolution's d = 10*n_mels):
// the same MFMA sweep with the feature axis cut into chunks of 64.  The NA*NB accumulators of a
// centroid tile stay live across the chunks (so every inner product is still ONE ascending fmaf
// chain); the centroid image is laid out [tile][chunk][row][64 swizzled] and streamed through LDS
// one (tile, chunk) piece at a time; the x chunk of a wave's rows is re-read from L2 per piece.
constexpr int DC = 64;  // features per chunk

__global__ void __launch_bounds__(WG) prep_centroids_chunked_kernel(const float* __restrict__ c, int k,
                                                                    int d, int nchunks, int na,
                                                                    float* __restrict__ img) {
    const int t = blockIdx.x;
    const int R = tile_rows(na);
    float* out = img + (size_t)t * ((size_t)R * DC * nchunks + CN_PAD);
    const int chunks16 = DC / 4;  // 16-byte chunks per row piece
    for (int e = threadIdx.x; e < nchunks * R * chunks16; e += WG) {
        const int ch = e / (R * chunks16);
        const int r = (e / chunks16) % R, pc = e % chunks16;
        const int lc = pc ^ (r & 15);
        const int q = lc >> 1, h = lc & 1;
        const int row = t * R + r;
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int f = ch * DC + 8 * q + 2 * u + h;
            v[u] = (row < k && f < d) ? c[(size_t)row * d + f] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(out + ((size_t)ch * R + r) * DC + pc * 4) = v;
    }
    for (int r = threadIdx.x; r < CN_PAD; r += WG) {
        const int row = t * R + r;
        float nrm = __builtin_inff();
        if (r < R && row < k) {
            nrm = 0.0f;
            for (int f = 0; f < d; f++) {
                const float v = c[(size_t)row * d + f];
                nrm = __builtin_fmaf(v, v, nrm);
            }
        }
        out[(size_t)R * DC * nchunks + r] = nrm;
    }
}

template <int NB, int NA, int WPS>
__global__ void __launch_bounds__(WG, WPS)
assign_mfma_anyd_kernel(const float* __restrict__ X, long n, int d, int nchunks,
                        const float* __restrict__ img, int ntiles, long* __restrict__ ids,
                        float* __restrict__ dist) {
    constexpr int R = tile_rows(NA);
    constexpr int PIECE_F = R * DC;                 // floats of one (tile, chunk) piece
    constexpr int BUF_F = PIECE_F + CN_PAD;         // + the norms behind the last chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * BUF_F floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const long row0 = ((long)blockIdx.x * 4 + wave) * (32 * NB);
    const size_t tile_f = (size_t)PIECE_F * nchunks + CN_PAD;

    const float* xrow[NB];
    float xn[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        long r = row0 + 32 * b + j;
        if (r >= n) r = n - 1;
        xrow[b] = X + r * d;
        float nrm = 0.0f;
        for (int f = 0; f < d; f += 4) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(xrow[b] + f);
            nrm = __builtin_fmaf(u[0], u[0], nrm);
            nrm = __builtin_fmaf(u[1], u[1], nrm);
            nrm = __builtin_fmaf(u[2], u[2], nrm);
            nrm = __builtin_fmaf(u[3], u[3], nrm);
        }
        xn[b] = nrm;
    }

    float bestd[NB];
    unsigned bestc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        bestd[b] = __builtin_inff();
        bestc[b] = 0xffffffffu;
    }

    // piece s = ct * nchunks + ch lives at img + ct*tile_f + ch*PIECE_F; the last chunk drags the
    // norms along (they sit right behind it)
    auto stage_dma = [&](int ct, int ch, float* dst) {
        const float* src = img + (size_t)ct * tile_f + (size_t)ch * PIECE_F;
        const int pieces = (ch == nchunks - 1 ? BUF_F : PIECE_F) / 256;
        for (int p = wave; p < pieces; p += 4) dma_1k(src + p * 256 + lane * 4, dst + p * 256);
    };
    stage_dma(0, 0, smem);
    __syncthreads();

    const int swz = j & 15;
    const int nstages = ntiles * nchunks;
    int ct = 0, ch = 0;
    f32x16 acc[NA][NB];
    // A wave's x chunk (features chx*64 .. +63 of its 32*NB rows, zero past d) as the B operand.  Two register
    // sets alternate: the chunk of stage s+1 is requested from L2 before the MFMAs of stage s are issued, so its
    // latency runs beside them instead of in front of the next stage (round 1 reloaded it at the top of every stage
    // and waited: 41-55 % of the fp32 MFMA peak at d = 640).
    auto load_x = [&](int chx, float (&xr)[NB][DC / 2]) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int q = 0; q < DC / 8; q++) {
                const int f = chx * DC + 8 * q;
                f32x4 u = {0, 0, 0, 0}, v = {0, 0, 0, 0};
                if (f < d) u = *reinterpret_cast<const f32x4*>(xrow[b] + f);          // d % 4 == 0
                if (f + 4 < d) v = *reinterpret_cast<const f32x4*>(xrow[b] + f + 4);
                xr[b][4 * q + 0] = h ? u[1] : u[0];
                xr[b][4 * q + 1] = h ? u[3] : u[2];
                xr[b][4 * q + 2] = h ? v[1] : v[0];
                xr[b][4 * q + 3] = h ? v[3] : v[2];
            }
        }
    };
    auto stage = [&](int s, const float (&xr)[NB][DC / 2], float (&xr_next)[NB][DC / 2]) {
        const float* cur = smem + (s & 1) * BUF_F;
        {
            int nct = ct, nch = ch + 1;
            if (nch == nchunks) { nch = 0; nct++; }
            if (s + 1 < nstages) {
                stage_dma(nct, nch, smem + ((s + 1) & 1) * BUF_F);
                load_x(nch, xr_next);
            }
        }
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const float* arow = cur + (a * 32 + j) * DC;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                f32x16 cacc = acc[a][b];
                if (ch == 0) cacc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < DC / 8; q++) {
                    const int pc = (2 * q + h) ^ swz;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow + pc * 4);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], xr[b][4 * q + 0], cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], xr[b][4 * q + 1], cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], xr[b][4 * q + 2], cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], xr[b][4 * q + 3], cacc, 0, 0, 0);
                }
                acc[a][b] = cacc;
            }
        }
        if (ch == nchunks - 1) {  // tile finished: arg-min over its NA*NB accumulators
#pragma unroll
            for (int a = 0; a < NA; a++) {
                f32x4 cnv[4];
#pragma unroll
                for (int g = 0; g < 4; g++)
                    cnv[g] = *reinterpret_cast<const f32x4*>(cur + PIECE_F + a * 32 + 8 * g + 4 * h);
                const unsigned codebase = (unsigned)(ct * NA + a) * 16u;
#pragma unroll
                for (int b = 0; b < NB; b++) epilogue16(bestd[b], bestc[b], acc[a][b], xn[b], cnv, codebase);
            }
        }
        if (++ch == nchunks) { ch = 0; ct++; }
        __syncthreads();
    };
    float xrA[NB][DC / 2], xrB[NB][DC / 2];
    load_x(0, xrA);
    for (int s = 0; s < nstages; s += 2) {
        stage(s, xrA, xrB);
        if (s + 1 < nstages) stage(s + 1, xrB, xrA);
    }

#pragma unroll
    for (int b = 0; b < NB; b++) {
        int idx = -1;
        if (bestc[b] != 0xffffffffu) {
            const unsigned r = bestc[b] & 15u;
            idx = (int)((bestc[b] >> 4) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (unsigned)h);
        }
        const float od = __shfl_xor(bestd[b], 32);
        const int oi = __shfl_xor(idx, 32);
        float fd = bestd[b];
        int fi = idx;
        if (od < fd || (od == fd && (unsigned)oi < (unsigned)fi)) {
            fd = od;
            fi = oi;
        }
        const long r = row0 + 32 * b + j;
        if (h == 0 && r < n) {
            ids[r] = (long)fi;
            if (dist) dist[r] = fd;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Any d at all (fallback: scalar fmaf chains, one thread per row; same arithmetic as the MFMA path).
// Used only when d is not a multiple of 4 or the rows are not 16-byte aligned.
__global__ void __launch_bounds__(WG)
assign_generic_kernel(const float* __restrict__ X, long n, int d, const float* __restrict__ C,
                      const float* __restrict__ cn, int k, long* __restrict__ ids,
                      float* __restrict__ dist) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const float* xi = X + i * d;
    float xn = 0.0f;
    for (int f = 0; f < d; f++) xn = __builtin_fmaf(xi[f], xi[f], xn);
    float best = __builtin_inff();
    long bi = -1;
    for (int c = 0; c < k; c++) {
        const float* cc = C + (size_t)c * d;
        float ip = 0.0f;
        for (int f = 0; f < d; f++) ip = __builtin_fmaf(xi[f], cc[f], ip);
        float dis = __builtin_fmaf(-2.0f, ip, xn + cn[c]);
        dis = __builtin_fmaxf(dis, 0.0f);
        if (dis < best) {
            best = dis;
            bi = c;
        }
    }
    ids[i] = bi;
    if (dist) dist[i] = best;
}

__global__ void __launch_bounds__(WG) row_sqnorm_kernel(const float* __restrict__ C, int k, int d,
                                                        float* __restrict__ cn) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= k) return;
    float s = 0.0f;
    for (int f = 0; f < d; f++) s = __builtin_fmaf(C[(size_t)c * d + f], C[(size_t)c * d + f], s);
    cn[c] = s;
}

// faiss exhaustive_L2sqr_seq (n < distance_compute_blas_threshold = 20): direct sum (x-c)^2.
__global__ void assign_small_kernel(const float* __restrict__ X, int n, int d,
                                    const float* __restrict__ C, int k, long* __restrict__ ids,
                                    float* __restrict__ dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = __builtin_inff();
    long bi = -1;
    for (int c = 0; c < k; c++) {
        float acc = 0.0f;
        for (int f = 0; f < d; f++) {
            const float df = X[(size_t)i * d + f] - C[(size_t)c * d + f];
            acc = __builtin_fmaf(df, df, acc);
        }
        if (acc < best) {
            best = acc;
            bi = c;
        }
    }
    ids[i] = bi;
    if (dist) dist[i] = best;
}

template <int D, int NB, int NA, bool DMA, int WPS, bool SPEC = false>
int launch_mfma(at_ctx* ctx, const float* x, int64_t n, const float* c, int k, int64_t* ids, float* dist,
                hipStream_t stream) {
    const int ntiles = (k + tile_rows(NA) - 1) / tile_rows(NA);
    const size_t img_bytes = sizeof(float) * (size_t)ntiles * tile_floats(D, NA);
    float* img = static_cast<float*>(at_ws(ctx, WS_CENT_IMG, img_bytes, stream));
    if (!img) return AT_E_NOMEM;
    AT_LAUNCH(prep_centroids_kernel, dim3(ntiles), dim3(WG), 0, stream, c, k, D, D, NA, img);
    const size_t lds = 2 * sizeof(float) * tile_floats(D, NA);
    const int64_t rows_per_wg = 4 * 32 * NB;
    const int64_t grid = (n + rows_per_wg - 1) / rows_per_wg;
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&assign_mfma_kernel<D, NB, NA, DMA, WPS, SPEC>), lds); if (rcl_) return rcl_; }
    AT_LAUNCH((assign_mfma_kernel<D, NB, NA, DMA, WPS, SPEC>), dim3((unsigned)grid), dim3(WG), lds,
                       stream, x, (long)n, img, ntiles, reinterpret_cast<long*>(ids), dist);
    return AT_OK;
}


template <int NB, int NA, int WPS>
static int launch_anyd(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, int64_t* ids, float* dist,
                       hipStream_t stream) {
    const int nchunks = (d + DC - 1) / DC;
    const int ntiles = (k + tile_rows(NA) - 1) / tile_rows(NA);
    const size_t tile_f = (size_t)tile_rows(NA) * DC * nchunks + CN_PAD;
    float* img = static_cast<float*>(at_ws(ctx, WS_CENT_IMG, sizeof(float) * ntiles * tile_f, stream));
    if (!img) return AT_E_NOMEM;
    AT_LAUNCH(prep_centroids_chunked_kernel, dim3(ntiles), dim3(WG), 0, stream, c, k, d, nchunks, NA, img);
    const size_t lds = 2 * sizeof(float) * (tile_rows(NA) * DC + CN_PAD);
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&assign_mfma_anyd_kernel<NB, NA, WPS>), lds); if (rcl_) return rcl_; }
    const int64_t rows_per_wg = 4 * 32 * NB;
    AT_LAUNCH((assign_mfma_anyd_kernel<NB, NA, WPS>), dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)),
                       dim3(WG), lds, stream, x, (long)n, d, nchunks, img, ntiles, reinterpret_cast<long*>(ids), dist);
    return AT_OK;
}

}  // namespace

extern "C" int at_assign_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                             int64_t* ids, float* dist, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx, "at_assign_f32: ctx is null");
    AT_REQUIRE(n >= 0 && d > 0 && k > 0, "at_assign_f32: bad sizes n=%lld d=%d k=%d", (long long)n, d, k);
    if (n == 0) return AT_OK;
    AT_REQUIRE(x && c && ids, "at_assign_f32: null pointer");
    AT_REQUIRE(k <= (1 << 24), "at_assign_f32: k=%d too large", k);
    AT_HIP(hipSetDevice(ctx->device));

    if (n < 20) {
        AT_LAUNCH(assign_small_kernel, dim3(1), dim3(32), 0, stream, x, (int)n, d, c, k,
                           reinterpret_cast<long*>(ids), dist);
        return AT_OK;
    }

    if (d == 64 || d == 128) {
        AT_REQUIRE(at_aligned16(x), "at_assign_f32: x must be 16-byte aligned");
        // Shipped shapes: LDS-DMA staging, 128-centroid tiles at d=64 (64 at d=128), 2 waves/SIMD.
        // AT_ASSIGN_VARIANT=1 selects the register-staged 64-centroid-tile kernel (A/B aid).
        const int v = ctx->dbg.assign_variant;
        if (d == 64) {
            if (v == 1) return launch_mfma<64, 2, 2, false, 2>(ctx, x, n, c, k, ids, dist, stream);
            if (v == 3) return launch_mfma<64, 2, 4, true, 2, true>(ctx, x, n, c, k, ids, dist, stream);
            return launch_mfma<64, 2, 4, true, 2>(ctx, x, n, c, k, ids, dist, stream);
        }
        if (v == 1) return launch_mfma<128, 1, 2, false, 2>(ctx, x, n, c, k, ids, dist, stream);
        if (v == 3) return launch_mfma<128, 1, 2, true, 2, true>(ctx, x, n, c, k, ids, dist, stream);
        return launch_mfma<128, 1, 2, true, 2>(ctx, x, n, c, k, ids, dist, stream);
    }

    if (d % 4 == 0 && at_aligned16(x) && ctx->dbg.assign_variant != 2) {
        // shapes (assign_variant: A/B aid): 0 = two row tiles x two centroid tiles per wave at one wave per SIMD (both
        // x-chunk register sets fit); 4 = one row tile at two waves per SIMD
        const int v = ctx->dbg.assign_variant;
        if (v == 4) return launch_anyd<1, 2, 2>(ctx, x, n, d, c, k, ids, dist, stream);
        if (v == 5) return launch_anyd<1, 2, 3>(ctx, x, n, d, c, k, ids, dist, stream);
        if (v == 6) return launch_anyd<2, 2, 1>(ctx, x, n, d, c, k, ids, dist, stream);
        return launch_anyd<1, 4, 2>(ctx, x, n, d, c, k, ids, dist, stream);
    }

    float* cn = static_cast<float*>(at_ws(ctx, WS_CENT_IMG, sizeof(float) * (size_t)k, stream));
    if (!cn) return AT_E_NOMEM;
    AT_LAUNCH(row_sqnorm_kernel, dim3((k + WG - 1) / WG), dim3(WG), 0, stream, c, k, d, cn);
    AT_LAUNCH(assign_generic_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0,
                       stream, x, (long)n, d, c, cn, k, reinterpret_cast<long*>(ids), dist);
    return AT_OK;
}

extern "C" int at_assign_hinted_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                                    const int64_t* hint_ids, const uint32_t* order,
                                    const uint32_t* hint_sorted, int64_t* ids, float* dist, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx, "at_assign_hinted_f32: ctx is null");
    // shapes without a hinted kernel, or no hints at all: the plain sweep gives the same answer
    if (hint_sorted && !order) hint_sorted = nullptr;  // positions mean nothing without the order
    if ((!hint_ids && !hint_sorted) || n < 20 || !(d == 64 || d == 128) || !at_aligned16(x) || !at_aligned16(c))
        return at_assign_f32(ctx, x, n, d, c, k, ids, dist, stream_);
    AT_REQUIRE(n >= 0 && k > 0 && k <= (1 << 24) && n < (int64_t)UINT32_MAX, "at_assign_hinted_f32: bad sizes");
    AT_REQUIRE(x && c && ids, "at_assign_hinted_f32: null pointer");
    AT_REQUIRE(ids != hint_ids, "at_assign_hinted_f32: ids must not alias hint_ids");
    AT_HIP(hipSetDevice(ctx->device));
    if (d == 64) {
        constexpr int D = 64, NB = 2, NA = 4;
        const int ntiles = (k + tile_rows(NA) - 1) / tile_rows(NA);
        float* img = static_cast<float*>(at_ws(ctx, WS_CENT_IMG, sizeof(float) * (size_t)ntiles * tile_floats(D, NA), stream));
        if (!img) return AT_E_NOMEM;
        AT_LAUNCH(prep_centroids_kernel, dim3(ntiles), dim3(WG), 0, stream, c, k, D, D, NA, img);
        const size_t lds = 2 * sizeof(float) * tile_floats(D, NA);
        { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&assign_mfma_hinted_kernel<D, NB, NA>), lds); if (rcl_) return rcl_; }
        const int64_t rows_per_wg = 4 * 32 * NB;
        AT_LAUNCH((assign_mfma_hinted_kernel<D, NB, NA>), dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)),
                           dim3(WG), lds, stream, x, (long)n, c, k, img, ntiles, order,
                           reinterpret_cast<const long*>(hint_ids), hint_sorted, reinterpret_cast<long*>(ids), dist);
        return AT_OK;
    }
    constexpr int D = 128, NB = 1, NA = 2;
    const int ntiles = (k + tile_rows(NA) - 1) / tile_rows(NA);
    float* img = static_cast<float*>(at_ws(ctx, WS_CENT_IMG, sizeof(float) * (size_t)ntiles * tile_floats(D, NA), stream));
    if (!img) return AT_E_NOMEM;
    AT_LAUNCH(prep_centroids_kernel, dim3(ntiles), dim3(WG), 0, stream, c, k, D, D, NA, img);
    const size_t lds = 2 * sizeof(float) * tile_floats(D, NA);
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&assign_mfma_hinted_kernel<D, NB, NA>), lds); if (rcl_) return rcl_; }
    const int64_t rows_per_wg = 4 * 32 * NB;
    AT_LAUNCH((assign_mfma_hinted_kernel<D, NB, NA>), dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)),
                       dim3(WG), lds, stream, x, (long)n, c, k, img, ntiles, order,
                       reinterpret_cast<const long*>(hint_ids), hint_sorted, reinterpret_cast<long*>(ids), dist);
    return AT_OK;
}

int at_prune_prepass(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                     const uint32_t* order, const uint32_t* hint_sorted, const float* dmin, int ng,
                     float* bd_out, uint32_t* mask, int ngw, int mode, hipStream_t stream);

template <int D, int NB>
static int launch_pruned(at_ctx* ctx, const float* x, int64_t n, const float* c, int k, const uint32_t* order,
                         const uint32_t* hint_sorted, const int32_t* cperm, int ng, const float* dmin, int mode,
                         bool prepass_done, bool filter, int64_t* ids, float* dist, hipStream_t stream) {
    const int kp = ng * 32;
    const int ngw = (ng + 31) / 32;
    const size_t img_bytes = sizeof(float) * (size_t)ng * tile_floats(D, 1);
    const int64_t ntile32 = (n + 31) / 32;
    float* img = static_cast<float*>(at_ws(ctx, WS_CENT_IMG, img_bytes, stream));
    float* bd = static_cast<float*>(at_ws(ctx, WS_PRUNE_BD, sizeof(float) * (size_t)n, stream));
    uint32_t* mask = static_cast<uint32_t*>(at_ws(ctx, WS_PRUNE_MASK, sizeof(uint32_t) * (size_t)ntile32 * ngw, stream));
    if (!img || !bd || !mask) return AT_E_NOMEM;
    // fp32 image (one tile per group, NA = 1: 32 rows, then |c|^2 at [0,32) and indices at [128,160)): built
    // only where the fp32 sweep runs -- without the filter, or for its long-list redo
    auto prep_fp32_image = [&]() -> int {
        AT_LAUNCH(prep_centroids_perm_kernel, dim3(ng), dim3(WG), 0, stream, c, k, D, cperm, kp, 1, img);
        return AT_OK;
    };
    if (!filter) {
        int rc = prep_fp32_image();
        if (rc) return rc;
    }
    // exact filtered calls without a pre-pass done by the caller: the sweep does it in its prologue
    const bool fuse = filter && mode == 0 && !prepass_done && ctx->dbg.filter_fused != 0;   // (switch: 0 = separate pre-pass kernel)
    if (!prepass_done && !fuse) {
        int rc = at_prune_prepass(ctx, x, n, D, c, k, order, hint_sorted, dmin, ng, bd, mask, ngw, mode, stream);
        if (rc) return rc;
    }
    const size_t lds = 2 * sizeof(float) * tile_floats(D, 1);
    const int64_t rows_per_wg = 32 * NB;
    const uint32_t* order2 = order;
    const uint32_t* hint2 = hint_sorted;
    int64_t n2 = n;
    if (filter) {
        // Stage 1: fp16-split filter (filter.hip) names the winner of every row whose runner-up is
        // provably out of reach and lists the others; stage 2 below redoes the listed rows with the
        // fp32 sweep.  Coarse mode (guesses only) needs neither the list nor stage 2.
        // statistics (and the list-length verdicts) of earlier calls whose words have arrived: polled, not waited for
        int rcp = at_filter_resolve_pending(ctx, false);
        if (rcp) return rcp;
        const bool async_form = mode == 0 && !ctx->filter_force_sync && ctx->dbg.filter_sync != 1;
        int slot = AT_FILTER_RING;
        if (async_form) {
            if (ctx->fring_count == AT_FILTER_RING) {   // the host is a whole ring ahead of the device: wait for the oldest
                AT_HIP(hipEventSynchronize(ctx->fring[ctx->fring_head].copied));
                rcp = at_filter_resolve_pending(ctx, false);
                if (rcp) return rcp;
            }
            slot = (ctx->fring_head + ctx->fring_count) % AT_FILTER_RING;
        } else if (mode == 0) {
            rcp = at_filter_resolve_pending(ctx, true);   // the synchronous form reads its words at once: keep the order
            if (rcp) return rcp;
        }
        rcp = at_filter_use_slot(ctx, slot);
        if (rcp) return rcp;
        unsigned* misc = static_cast<unsigned*>(at_ws(ctx, WS_FILTER_MISC, 1024, stream));
        const size_t lstride = at_amb_stride(n);   // five arrays of 64 sub-lists: list | sorted | order | hints | aux
        const unsigned amb_cap = at_amb_cap(n);
        uint32_t* list = static_cast<uint32_t*>(at_ws(ctx, WS_FILTER_LIST, sizeof(uint32_t) * 5 * lstride, stream));
        if (!misc || !list) return AT_E_NOMEM;
        uint32_t* aux = list + 4 * lstride;
        int rc = at_filter_sweep(ctx, x, n, D, c, k, order, cperm, ng, bd, mask, ngw, mode == 0 ? 1 : 0, ids, misc, list,
                                 aux, nullptr, fuse ? hint_sorted : nullptr, fuse ? dmin : nullptr, fuse ? bd : nullptr,
                                 (fuse || mode != 0) ? dist : nullptr, amb_cap, stream);
        if (rc) return rc;
        // (asynchronous fused calls: the distance pass rides in the launch that redoes the listed rows, below)
        const bool finish_fused = dist && fuse && async_form;
        if (dist && fuse && !finish_fused) {  // the sweep wrote the guess distances; the rows that moved get theirs here
            rc = at_exact_dist_todo(ctx, x, n, D, c, k, ids, dist, stream);
            if (rc) return rc;
        } else if (dist && mode == 0 && !fuse) {
            rc = at_exact_dist_rows(ctx, x, n, D, c, k, ids, dist, order, hint_sorted, bd, stream);
            if (rc) return rc;
        }  // (guess generators wrote an approximate distance themselves: it only orders the next visit)
        if (mode != 0) return AT_OK;
        // The list of rows to redo is normally short.  Asynchronous form: the redo kernel reads the list
        // length on the device, the statistics words go to pinned memory and are folded in at the next
        // call or query -- no host round trip here.  A call that turns out to have listed more than
        // n/16 rows (badly scaled data) was still answered correctly (the redo strides over any length),
        // but switches this context to the synchronous form below, whose redo of long lists is the
        // fp32 MFMA sweep.
        if (async_form) {
            at_filter_slot& fs = ctx->fring[slot];
            int64_t wgs = n / 64;                      // enough workgroups for a list of 1.5 % of the rows in one go
            if (wgs < 256) wgs = 256;
            if (wgs > 65535) wgs = 65535;
            if (finish_fused)
                rc = at_filter_finish(ctx, x, n, D, c, k, list, wgs, order, cperm, dmin, ng, misc, aux, ids, dist, misc + 64, amb_cap,
                                      stream);
            else
                rc = at_filter_redo_rows(ctx, x, D, c, k, list, wgs, order, cperm, dmin, ng, misc, aux, ids, dist, misc + 64, amb_cap,
                                         stream);
            if (rc) return rc;
            AT_HIP(hipMemcpyAsync(fs.host_misc, misc, 128 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            AT_HIP(hipEventRecord(fs.copied, stream));
            fs.rows = n;
            ctx->fring_count++;
            return AT_OK;
        }
        unsigned host_misc[128];
        AT_HIP(hipMemcpyAsync(host_misc, misc, sizeof host_misc, hipMemcpyDeviceToHost, stream));
        AT_HIP(hipStreamSynchronize(stream));
        unsigned listed = 0;
        for (unsigned s2 = 0; s2 < AT_AMB_SUBLISTS; s2++) listed += host_misc[64 + s2];
        ctx->filter_rows += n;
        ctx->filter_listed += listed;
        ctx->filter_tiles += host_misc[4];
        ctx->filter_refined += host_misc[5];
        {
            at_filter_slot& spare = ctx->fring[AT_FILTER_RING];
            float ms = 0.0f;
            if (spare.timed && AT_HIP_TOLERATE(hipEventElapsedTime(&ms, spare.ev[0], spare.ev[1])) == hipSuccess) {
                ctx->filter_ms += ms;
                ctx->filter_launches++;
            }
            spare.timed = 0;
        }
        if ((int64_t)listed * 16 <= n) ctx->filter_force_sync = 0;   // the data behave again
        if (listed == 0) return AT_OK;
        // short lists: one workgroup per row on the vector ALU (it walks the sub-lists itself); long ones (badly
        // conditioned data, centroids outside the fp16 range): the fp32 MFMA sweep over the listed rows, which wants
        // them in one piece
        if ((int64_t)listed * 16 <= n) {
            int64_t wgs = listed < 65535u ? (int64_t)listed : 65535;
            return at_filter_redo_rows(ctx, x, D, c, k, list, wgs, order, cperm, dmin, ng, misc, aux, ids, dist, misc + 64, amb_cap,
                                       stream);
        }
        rc = prep_fp32_image();
        if (rc) return rc;
        n2 = listed < 64 ? 64 : (int64_t)listed;
        uint32_t* sorted = list + lstride;
        uint32_t* order_amb = sorted + lstride;
        uint32_t* hint_amb = order_amb + lstride;
        // (listed <= n < the stride: the contiguous copy fits the `sorted` / `order` arrays; it then moves to the front of `list`)
        rc = at_amb_compact(ctx, misc, amb_cap, list, aux, sorted, order_amb, stream);
        if (rc) return rc;
        AT_HIP(hipMemcpyAsync(list, sorted, sizeof(uint32_t) * listed, hipMemcpyDeviceToDevice, stream));
        rc = at_filter_gather_ambiguous(ctx, list, sorted, listed, n2, order, ids, order_amb, hint_amb, stream);
        if (rc) return rc;
        order2 = order_amb;
        hint2 = hint_amb;
        rc = at_prune_prepass(ctx, x, n2, D, c, k, order2, hint2, dmin, ng, bd, mask, ngw, 0, stream);
        if (rc) return rc;
    }
    // d = 64: register-staged A operand (no LDS); AT_PRUNE_KERNEL=0 selects the LDS-DMA form (A/B aid)
    if (ctx->dbg.prune_kernel != 0 && D == 64 && NB <= 2)
        AT_LAUNCH((assign_mfma_pruned_reg_kernel<D, NB>), dim3((unsigned)((n2 + rows_per_wg - 1) / rows_per_wg)),
                           dim3(64), 0, stream, x, (long)n2, img, ng, order2, hint2, bd, mask, ngw,
                           reinterpret_cast<long*>(ids), dist);
    else
        AT_LAUNCH((assign_mfma_pruned_kernel<D, NB>), dim3((unsigned)((n2 + rows_per_wg - 1) / rows_per_wg)),
                           dim3(64), lds, stream, x, (long)n2, img, ng, order2, hint2, bd, mask, ngw,
                           reinterpret_cast<long*>(ids), dist);
    return AT_OK;
}

extern "C" int at_assign_pruned_f32(at_ctx* ctx, const at_pruned_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && a, "at_assign_pruned_f32: ctx / args is null");
    const float *x = a->x, *c = a->c, *dmin = a->bounds;
    const int64_t n = a->n;
    const int d = a->d, k = a->k, ng = a->ng;
    const uint32_t *order = a->order, *hint_sorted = a->hint_sorted;
    const int32_t* cperm = a->cperm;
    int64_t* ids = a->ids;
    float* dist = a->dist_or_null;
    const int mode = a->guess_only ? 1 : 0;
    const bool filter = a->use_filter != 0;
    const int prepass_done = a->prepass_done ? 1 : 0;
    AT_REQUIRE(x && c && order && hint_sorted && cperm && (dmin || mode == 1) && ids, "at_assign_pruned_f32: null pointer");
    AT_REQUIRE(d == 64 || d == 128, "at_assign_pruned_f32: d must be 64 or 128");
    AT_REQUIRE(n >= 20 && n < (int64_t)UINT32_MAX && k > 0 && ng > 0 && ng <= 512 && ng * 32 >= k,
               "at_assign_pruned_f32: bad sizes n=%lld k=%d ng=%d", (long long)n, k, ng);
    AT_REQUIRE(at_aligned16(x) && at_aligned16(c), "at_assign_pruned_f32: x and c must be 16-byte aligned");
    AT_HIP(hipSetDevice(ctx->device));
    ctx->img16_trusted = a->image_current != 0;  // consumed (and cleared) by the filter sweep
    if (d == 64) {
        const int nbsw = ctx->dbg.prune_nb;
        if (nbsw == 4) return launch_pruned<64, 4>(ctx, x, n, c, k, order, hint_sorted, cperm, ng, dmin, mode, prepass_done != 0, filter, ids, dist, stream);
        if (nbsw == 1) return launch_pruned<64, 1>(ctx, x, n, c, k, order, hint_sorted, cperm, ng, dmin, mode, prepass_done != 0, filter, ids, dist, stream);
        return launch_pruned<64, 2>(ctx, x, n, c, k, order, hint_sorted, cperm, ng, dmin, mode, prepass_done != 0, filter, ids, dist, stream);
    }
    return launch_pruned<128, 2>(ctx, x, n, c, k, order, hint_sorted, cperm, ng, dmin, mode, prepass_done != 0, filter, ids, dist, stream);
}

// The pre-pass of at_assign_pruned_f32 on its own (per-row bound + per-tile group masks, kept in the
// context's workspace): lets a caller time or overlap it separately, then call
// at_assign_pruned_f32(..., prepass_done = 1) with the same arguments.
extern "C" int at_prune_mask_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                                 const uint32_t* order, const uint32_t* hint_sorted, int ng, const float* dmin,
                                 int mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && x && c && order && hint_sorted && (dmin || mode == 1), "at_prune_mask_f32: null pointer");
    AT_REQUIRE((d == 64 || d == 128) && n >= 20 && n < (int64_t)UINT32_MAX && ng > 0 && ng <= 512 && ng * 32 >= k,
               "at_prune_mask_f32: bad sizes");
    AT_HIP(hipSetDevice(ctx->device));
    const int ngw = (ng + 31) / 32;
    const int64_t ntile32 = (n + 31) / 32;
    float* bd = static_cast<float*>(at_ws(ctx, WS_PRUNE_BD, sizeof(float) * (size_t)n, stream));
    uint32_t* mask = static_cast<uint32_t*>(at_ws(ctx, WS_PRUNE_MASK, sizeof(uint32_t) * (size_t)ntile32 * ngw, stream));
    if (!bd || !mask) return AT_E_NOMEM;
    return at_prune_prepass(ctx, x, n, d, c, k, order, hint_sorted, dmin, ng, bd, mask, ngw, mode, stream);
}

// The statistics ring of the asynchronous exact calls (at_internal.h: at_filter_slot).
// Life of a slot: claimed by at_filter_use_slot (timed = 0) -> the sweep records ev[0], ev[1] around its kernel only
// under the switch filter_timing and then sets timed = 1 -> the call queues the copy of its statistics words and
// records `copied` -> pushed (fring_count++) -> at_filter_resolve_pending reads the words once `copied` has
// completed and the two timing events only if timed is set.  hipEventElapsedTime on an event that was never
// recorded returns hipErrorInvalidResourceHandle and leaves it pending in the thread (the error round 2's launch
// check then blamed on the next kernel launch): it is not called on such events any more, and its result is
// consumed where it is tolerated.
int at_filter_use_slot(at_ctx* ctx, int slot) {
    AT_REQUIRE(slot >= 0 && slot <= AT_FILTER_RING, "at_filter_use_slot: slot %d out of range", slot);
    if (!ctx->filter_host_misc) {
        AT_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->filter_host_misc), (size_t)(AT_FILTER_RING + 1) * 128 * sizeof(unsigned),
                             hipHostMallocDefault));
        for (int s = 0; s <= AT_FILTER_RING; s++) ctx->fring[s].host_misc = ctx->filter_host_misc + (size_t)s * 128;
    }
    at_filter_slot& fs = ctx->fring[slot];
    if (!fs.copied) AT_HIP(hipEventCreateWithFlags(&fs.copied, hipEventDisableTiming));
    fs.timed = 0;
    ctx->filter_slot = slot;
    return AT_OK;
}

int at_filter_resolve_pending(at_ctx* ctx, bool wait_all) {
    while (ctx->fring_count > 0) {
        at_filter_slot& fs = ctx->fring[ctx->fring_head];
        if (wait_all) {
            AT_HIP(hipEventSynchronize(fs.copied));
        } else {
            const hipError_t q = hipEventQuery(fs.copied);
            if (q == hipErrorNotReady) break;   // an answer, not a failure (and the one code HIP does not keep pending)
            AT_HIP(q);
        }
        const unsigned* hm = fs.host_misc;
        unsigned listed = 0;
        for (unsigned s2 = 0; s2 < AT_AMB_SUBLISTS; s2++) listed += hm[64 + s2];
        ctx->filter_rows += fs.rows;
        ctx->filter_listed += listed;
        ctx->filter_tiles += hm[4];
        ctx->filter_refined += hm[5];
        if (fs.timed) {   // (both events were recorded before `copied`, on the same stream: they have completed)
            float ms = 0.0f;
            if (AT_HIP_TOLERATE(hipEventElapsedTime(&ms, fs.ev[0], fs.ev[1])) == hipSuccess) {
                ctx->filter_ms += ms;
                ctx->filter_launches++;
            }
            fs.timed = 0;
        }
        if ((int64_t)listed * 16 > fs.rows) ctx->filter_force_sync = 1;
        ctx->fring_head = (ctx->fring_head + 1) % AT_FILTER_RING;
        ctx->fring_count--;
    }
    return AT_OK;
}

extern "C" int at_filter_stats(at_ctx* ctx, int64_t* rows, int64_t* listed, double* sweep_ms, int64_t* sweeps,
                               int64_t* tiles, int64_t* refined, int reset) {
    AT_REQUIRE(ctx && rows && listed, "at_filter_stats: bad arguments");
    {
        int rcp = at_filter_resolve_pending(ctx, true);
        if (rcp) return rcp;
    }
    *rows = ctx->filter_rows;
    *listed = ctx->filter_listed;
    if (sweep_ms) *sweep_ms = ctx->filter_ms;
    if (sweeps) *sweeps = ctx->filter_launches;
    if (tiles) *tiles = ctx->filter_tiles;
    if (refined) *refined = ctx->filter_refined;
    if (reset) {
        ctx->filter_rows = ctx->filter_listed = ctx->filter_launches = ctx->filter_tiles = ctx->filter_refined = 0;
        ctx->filter_ms = 0.0;
    }
    return AT_OK;
}

// Test hook: pre-pass + stage 1 only.  approx[2i] = approximate |c|^2 - 2 x.c of row i's winner,
// approx[2i+1] = gap to the runner-up; ids = the winners; *listed = rows the filter would hand to the
// fp32 sweep; tau_ab[0..1] = the coefficients of the acceptance threshold tau = a (|x|^2 + max|c|^2) + b.
extern "C" int at_filter_probe_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                                   const uint32_t* order, const uint32_t* hint_sorted, const int32_t* cperm, int ng,
                                   const float* dmin, int64_t* ids, float* approx, int64_t* listed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx && x && c && order && hint_sorted && cperm && dmin && ids && approx && listed,
               "at_filter_probe_f32: null pointer");
    AT_REQUIRE((d == 64 || d == 128) && n >= 20 && n < (int64_t)UINT32_MAX && k > 0 && ng > 0 && ng <= 512 && ng * 32 >= k,
               "at_filter_probe_f32: bad sizes");
    AT_HIP(hipSetDevice(ctx->device));
    const int ngw = (ng + 31) / 32;
    const int64_t ntile32 = (n + 31) / 32;
    float* bd = static_cast<float*>(at_ws(ctx, WS_PRUNE_BD, sizeof(float) * (size_t)n, stream));
    uint32_t* mask = static_cast<uint32_t*>(at_ws(ctx, WS_PRUNE_MASK, sizeof(uint32_t) * (size_t)ntile32 * ngw, stream));
    unsigned* misc = static_cast<unsigned*>(at_ws(ctx, WS_FILTER_MISC, 1024, stream));
    const size_t lstride = at_amb_stride(n);
    uint32_t* list = static_cast<uint32_t*>(at_ws(ctx, WS_FILTER_LIST, sizeof(uint32_t) * 5 * lstride, stream));
    if (!bd || !mask || !misc || !list) return AT_E_NOMEM;
    int rc = at_prune_prepass(ctx, x, n, d, c, k, order, hint_sorted, dmin, ng, bd, mask, ngw, 0, stream);
    if (rc) return rc;
    rc = at_filter_use_slot(ctx, AT_FILTER_RING);   // (a call outside the ring: the spare slot)
    if (rc) return rc;
    rc = at_filter_sweep(ctx, x, n, d, c, k, order, cperm, ng, bd, mask, ngw, 1, ids, misc, list,
                         list + 4 * lstride, approx, nullptr, nullptr, nullptr, nullptr, at_amb_cap(n), stream);
    if (rc) return rc;
    unsigned cnts[AT_AMB_SUBLISTS];
    AT_HIP(hipMemcpyAsync(cnts, misc + 64, sizeof cnts, hipMemcpyDeviceToHost, stream));
    AT_HIP(hipStreamSynchronize(stream));
    int64_t cnt = 0;
    for (unsigned s2 = 0; s2 < AT_AMB_SUBLISTS; s2++) cnt += cnts[s2];
    *listed = cnt;
    return AT_OK;
}

// Guess generator without a pre-sort, for rows whose own order is coherent (the frames of a clip follow
// one another): nearest of the ng group means -> the groups its neighbour table names -> best centroid
// among them, in one launch.  ids are guesses (feed at_visit_order_f32 / at_assign_pruned_f32), dist (optional)
// approximate distances.
extern "C" int at_assign_coarse_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                                    const int32_t* cperm, int ng, const float* means, const uint32_t* gnbr,
                                    int64_t* ids, float* dist, void* stream_) {
    AT_REQUIRE(ctx && x && c && cperm && means && gnbr && ids, "at_assign_coarse_f32: null pointer");
    AT_REQUIRE((d == 64 || d == 128) && n >= 1 && n < (int64_t)UINT32_MAX && k > 0 && ng > 0 && ng <= 512 && ng * 32 >= k,
               "at_assign_coarse_f32: bad sizes");
    AT_REQUIRE(at_aligned16(x) && at_aligned16(c) && at_aligned16(means), "at_assign_coarse_f32: pointers must be 16-byte aligned");
    AT_HIP(hipSetDevice(ctx->device));
    return at_filter_coarse(ctx, x, n, d, c, k, cperm, ng, means, gnbr, ids, dist, (hipStream_t)stream_);
}
