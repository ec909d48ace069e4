// logmel.hip -- fused waveform -> STFT -> |.|^2 -> mel -> dB on gfx950 (at_logmel_f32).
//
// Replaces SpectrogramGenerator.generate_mel_spectrogram (processors/spectrogram_generator.py:
// 123-126 of danavery/audio-tokens): torchaudio MelSpectrogram(sample_rate, n_mels, n_fft=512,
// hop_length) with its defaults (center=True, reflect padding, periodic Hann, power=2, htk mel
// scale, norm=None, f_min=0, f_max=sr//2) followed by AmplitudeToDB() (10*log10(max(x, 1e-10))).
// Optionally emits frame-major rows and the row L2 normalisation the two consumers apply next
// (cluster_creator.py:52,64-66; spec_tokenizer.py:76,106-109), which removes the reference's
// transpose + normalise round trip through memory.
//
// One workgroup = FPB consecutive frames of one clip.  Their samples are read from HBM once,
// coalesced, into LDS with the reflect padding applied on the way (every sample is shared by
// n_fft/hop = 4 frames).  A frame is owned by 16 lanes (4 frames per wavefront): 16-point FFTs
// in registers down the columns, one transpose through a conflict-free LDS tile, 16-point FFTs
// along the rows, even/odd untangling to the 257 power bins (logmel_core.h), then the mel
// filterbank as a banded dot product per filter (only the non-zero taps), 10*log10, and a staged
// coalesced store.  Algorithmic HBM traffic: hop*4 bytes in + n_mels*4 bytes out per frame.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "at_internal.h"
#include "l2norm_core.h"
#include "logmel_core.h"

namespace {

using namespace logmel;

constexpr int WG = 256;
constexpr int PREFETCH_REGS = 5;   // x 16 bytes x 256 lanes = 5120 samples per block: 32 frames at hop <= 148, 16 at hop <= 307
constexpr int TAB_WIN = 0, TAB_TW256 = 512, TAB_TW512 = 1024, TAB_FLOATS = 1536;
// A frame's LDS tile (FRAME_LDS_FLOATS = 2304 B) is a multiple of the 256-byte bank row, so the two frames of a half-wave
// hit the same banks with every 64-bit tile access; 128 bytes of slack between tiles put neighbouring frames on
// opposite halves of the banks.
constexpr int FRAME_STRIDE = FRAME_LDS_FLOATS + 32;

struct LogmelParams {
    const float* wave;
    long n_clips, L, wave_stride;
    int hop, T, n_mels, fpb;
    long n_blocks;          // (clip, block of fpb frames) pairs, walked by persistent workgroups
    int blocks_per_clip;
    const float* tabs;      // TAB_FLOATS floats: window, W256 twiddles, W512 twiddles
    const int* fb_start;    // [n_mels]
    const int* fb_len;      // [n_mels]
    const int* fb_off;      // [n_mels] offset into fb_wts
    const float* fb_wts;    // concatenated non-zero bands
    int fb_nw;              // number of weights; tables are copied to LDS when fb_lds != 0
    int fb_lds;
    int fb_quads;           // bands padded to 4-aligned quads of bins (16-byte reads) or stored as they are
    float* out;
    int frame_major, fuse_l2norm;
    int* bad;               // set to 1 when a frame's squared norm is not finite (fused unit rows only)
    unsigned* minmax;       // optional [n_clips][4]: ordered keys of the clip's smallest / largest dB value, a NaN flag
                            // (at_logmel_minmax_f32: SpectrogramGenerator's normalize option without a reduction pass)
};

// floats as unsigned keys that order the same way (atomicMin / atomicMax on them)
__device__ __forceinline__ unsigned ordered_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned kx) {
    return __uint_as_float((kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx);
}

__global__ void __launch_bounds__(256) minmax_init_kernel(unsigned* __restrict__ mm, long n_clips) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c < n_clips) {
        mm[4 * c] = 0xffffffffu;   // smallest key seen
        mm[4 * c + 1] = 0u;        // largest
        mm[4 * c + 2] = 0u;        // a NaN was seen
        mm[4 * c + 3] = 0u;
    }
}

// (spec - min) / (max - min) per clip with the extremes the log-mel kernel collected: two subtractions and one
// IEEE division per element, torch's bits (processors/spectrogram_generator.py:129-131); a NaN anywhere in the clip
// makes the whole clip NaN, as torch.min / max propagate it.
__global__ void __launch_bounds__(256) minmax_apply_kernel(float* __restrict__ x, long clip_elems, const unsigned* __restrict__ mm) {
    const long clip = blockIdx.y;
    float lo = key_to_float(mm[4 * clip]), hi = key_to_float(mm[4 * clip + 1]);
    if (mm[4 * clip + 2]) lo = hi = __builtin_nanf("");
    const float range = hi - lo;
    float* p = x + (size_t)clip * clip_elems;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < clip_elems; e += (long)gridDim.x * 256) p[e] = __fdiv_rn(p[e] - lo, range);
}

// PF: the next block's samples are prefetched through registers (needs a block of at most PREFETCH_REGS x WG x 4
// samples); otherwise they are staged at the top of the block.
template <bool PF>
__global__ void __launch_bounds__(WG, 2) logmel_kernel(LogmelParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, grp = lane >> 4;
    const int nsamp = (p.fpb - 1) * p.hop + NFFT;
    const int opitch = p.n_mels + 1;

    float* win = sm;                               // NFFT window values
    float* tw512 = win + NFFT;                     // 256 x (cos, -sin)
    float* samp = tw512 + NFFT;                    // nsamp (rounded up to 4)
    float* work = samp + ((nsamp + 3) & ~3);       // 16 frames x FRAME_LDS_FLOATS
    float* ostage = work + 16 * FRAME_STRIDE;      // fpb x opitch  (+ fpb denominators)
    // filterbank tables behind the staging area, 16-byte aligned: start4 | quads | off | padded weights
    int* fbi = reinterpret_cast<int*>(sm + ((2 * NFFT + ((nsamp + 3) & ~3) + 16 * FRAME_STRIDE + p.fpb * opitch + p.fpb + 3) & ~3));
    const int* fb_start = p.fb_start;
    const int* fb_len = p.fb_len;
    const int* fb_off = p.fb_off;
    const float* fb_wts = p.fb_wts;
    if (p.fb_lds) {
        const int nint = (3 * p.n_mels + 3) & ~3;
        for (int i = tid; i < nint + p.fb_nw; i += WG) fbi[i] = p.fb_start[i];  // one contiguous blob
        fb_start = fbi;
        fb_len = fbi + p.n_mels;
        fb_off = fbi + 2 * p.n_mels;
        fb_wts = reinterpret_cast<const float*>(fbi + nint);
    }
    // A lane's twiddles do not depend on the frame: they live in registers for the whole (persistent)
    // workgroup instead of being read from LDS once per frame (the window stays in LDS: the register file
    // does not hold more beside two 16-point FFTs; window and W512 stay in LDS).
    cpx twr[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++)
        twr[k1] = {p.tabs[TAB_TW256 + 2 * (k1 * 16 + l16)], p.tabs[TAB_TW256 + 2 * (k1 * 16 + l16) + 1]};
    for (int i = tid; i < NFFT; i += WG) {
        win[i] = p.tabs[TAB_WIN + i];
        tw512[i] = p.tabs[TAB_TW512 + i];
    }

    float* mybuf = work + (wave * 4 + grp) * FRAME_STRIDE;
    const int n_mel_iter = (p.n_mels + 15) >> 4;

    // The samples of a block travel global -> registers -> LDS: the loads of the NEXT block are issued
    // before this block's frames are computed and land in LDS after them, so their latency is not waited
    // for.  Blocks in the interior of a clip (no reflection, 16-byte aligned) move 16 bytes per lane.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int nsamp4 = (nsamp + 3) >> 2;
    f4 pre[PREFETCH_REGS];
    auto prefetch = [&](long blk) {
        const long clip = blk / p.blocks_per_clip;
        const int t0 = (int)(blk - clip * p.blocks_per_clip) * p.fpb;
        const float* w = p.wave + clip * p.wave_stride;
        const long s0 = (long)t0 * p.hop - NFFT / 2;
        const bool fast = s0 >= 0 && s0 + 4L * nsamp4 <= p.L && ((reinterpret_cast<uintptr_t>(w + s0) & 15) == 0);
        if (fast) {
            const f4* src = reinterpret_cast<const f4*>(w + s0);
#pragma unroll
            for (int j = 0; j < PREFETCH_REGS; j++) {
                const int i4 = tid + WG * j;
                if (i4 < nsamp4) pre[j] = src[i4];
            }
        } else {
#pragma unroll
            for (int j = 0; j < PREFETCH_REGS; j++) {
                const int i4 = tid + WG * j;
                if (i4 < nsamp4) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        long q = s0 + 4 * i4 + e;
                        if (q < 0) q = -q;                      // reflect, no edge repeat
                        if (q >= p.L) q = 2 * (p.L - 1) - q;
                        if (q < 0) q = 0;                       // only reachable for frames past T (never stored)
                        if (q >= p.L) q = p.L - 1;
                        pre[j][e] = w[q];
                    }
                }
            }
        }
    };
    if constexpr (PF)
        if ((long)blockIdx.x < p.n_blocks) prefetch(blockIdx.x);
    for (long blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
        const long clip = blk / p.blocks_per_clip;
        const int t0 = (int)(blk - clip * p.blocks_per_clip) * p.fpb;
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < PREFETCH_REGS; j++) {
                const int i4 = tid + WG * j;
                if (i4 < nsamp4) *reinterpret_cast<f4*>(samp + 4 * i4) = pre[j];
            }
        } else {
            const float* w = p.wave + clip * p.wave_stride;
            const long s0 = (long)t0 * p.hop - NFFT / 2;
            for (int i = tid; i < nsamp; i += WG) {
                long q = s0 + i;
                if (q < 0) q = -q;                      // reflect, no edge repeat
                if (q >= p.L) q = 2 * (p.L - 1) - q;
                if (q < 0) q = 0;                       // only reachable for frames past T (never stored)
                if (q >= p.L) q = p.L - 1;
                samp[i] = w[q];
            }
        }
        __syncthreads();  // samples in place; the previous block's staged output has been stored by everybody
        if constexpr (PF)
            if (blk + gridDim.x < p.n_blocks) prefetch(blk + gridDim.x);

        for (int pass = 0; pass * 16 < p.fpb; pass++) {
            const int f = pass * 16 + wave * 4 + grp;   // frame within the workgroup
            // every DS access below stays inside this frame's 16 lanes' private tile; a wavefront's
            // LDS instructions execute in order, so no barrier is needed between the phases
            phase1_w(l16, samp + f * p.hop, win, twr, mybuf);
            cpx z[16];
            phase2(l16, mybuf, z);
            phase3_publish(l16, z, mybuf);
            float pw[16], p256 = 0.0f;
            phase3_power(l16, z, mybuf, tw512, pw, p256);
#pragma unroll
            for (int e = 0; e < 16; e++) mybuf[l16 + 16 * e] = pw[e];
            if (l16 < 4) mybuf[256 + l16] = l16 == 0 ? p256 : 0.0f;  // bins 257..259 pad the last quad
            for (int i = 0; i < n_mel_iter; i++) {
                const int m = l16 + 16 * i;
                if (m < p.n_mels) {
                    const float s = p.fb_quads ? mel_band(mybuf, fb_start[m], fb_len[m], fb_wts + fb_off[m])
                                               : mel_band_plain(mybuf, fb_start[m], fb_len[m], fb_wts + fb_off[m]);
                    // clamp(x, 1e-10) then 10*log10: a clamped bin is exactly -100 dB (what a correctly
                    // rounded log10 of 1e-10f gives; the device log10f is 1 ulp off there)
                    // (v_log_f32 is within an ulp of log2 on normal inputs -- s > 1e-10 here --: 1.2e-5 dB at most, a
                    // quarter of what the stated tolerance leaves; the library log10f costs twenty instructions)
                    ostage[f * opitch + m] = !(s <= 1e-10f) ? 3.0102999566398120f * __builtin_amdgcn_logf(s) : -100.0f;   // (a NaN power stays NaN, as torch.clamp leaves it)
                }
            }
        }
        __syncthreads();

        const int nf = min(p.fpb, p.T - t0);  // frames of this block that exist
        if (p.minmax) {   // the clip's extremes, from the staged block: one pair of atomics per workgroup and block
            float lo = __builtin_inff(), hi = -__builtin_inff();
            int nan = 0;
            for (int e = tid; e < nf * p.n_mels; e += WG) {
                const int f = e / p.n_mels, m = e - f * p.n_mels;
                const float v = ostage[f * opitch + m];
                nan |= v != v;
                lo = __builtin_fminf(lo, v);
                hi = __builtin_fmaxf(hi, v);
            }
            for (int off = 32; off > 0; off >>= 1) {
                lo = __builtin_fminf(lo, __shfl_xor(lo, off));
                hi = __builtin_fmaxf(hi, __shfl_xor(hi, off));
                nan |= __shfl_xor(nan, off);
            }
            if (lane == 0) {
                if (lo <= hi) {
                    atomicMin(&p.minmax[4 * clip], ordered_key(lo));
                    atomicMax(&p.minmax[4 * clip + 1], ordered_key(hi));
                }
                if (nan) atomicOr(&p.minmax[4 * clip + 2], 1u);
            }
        }
        if (p.frame_major) {
            float* den = ostage + p.fpb * opitch;
            if (p.fuse_l2norm) {
                // eight lanes per frame (fpb <= 32 frames: all of them at once); the host fuses only 8 <= n_mels <= 128
                const int f = tid >> 3;
                if (f < nf) {
                    const float ss = l2n::pairwise_sumsq_8lanes(ostage + f * opitch, p.n_mels, tid & 7);
                    if ((tid & 7) == 0) {
                        den[f] = __builtin_sqrtf(ss) + 1e-10f;
                        if (!(ss < __builtin_inff())) atomicOr(p.bad, 1);   // a NaN / Inf in the frame (faiss' input check, for free)
                    }
                }
                __syncthreads();
            }
            float* dst = p.out + ((long)clip * p.T + t0) * p.n_mels;
            if ((p.n_mels & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {   // 16 bytes per lane
                const int mq = p.n_mels >> 2, total4 = nf * mq;
                for (int e4 = tid; e4 < total4; e4 += WG) {
                    const int f = e4 / mq, m = (e4 - f * mq) * 4;
                    const float* src = ostage + f * opitch + m;
                    f4 v = {src[0], src[1], src[2], src[3]};
                    if (p.fuse_l2norm) {
                        const float dn = den[f];
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = l2n::divide(v[e], dn);
                    }
                    reinterpret_cast<f4*>(dst)[e4] = v;
                }
            } else {
                const int total = nf * p.n_mels;
                for (int e = tid; e < total; e += WG) {
                    const int f = e / p.n_mels, m = e - f * p.n_mels;
                    float v = ostage[f * opitch + m];
                    if (p.fuse_l2norm) v = l2n::divide(v, den[f]);
                    dst[e] = v;
                }
            }
        } else {
            float* dst = p.out + (long)clip * p.n_mels * p.T + t0;
            const int total = p.n_mels * p.fpb;
            for (int e = tid; e < total; e += WG) {
                const int m = e / p.fpb, f = e - m * p.fpb;
                if (f < nf) dst[(long)m * p.T + f] = ostage[f * opitch + m];
            }
        }
        // (no barrier here: the next block's samples go to `samp`, which nobody reads any more, and its
        // barrier above comes before anybody writes the staged output again)
    }
}

// Host side: window / twiddles / banded filterbank, uploaded once per (sr, n_mels, fb) change.
// LDS bytes of one workgroup of `fpb` frames (kernel layout), with `fb_ints` words of filterbank tables
size_t lds_bytes(int fpb, int hop, int n_mels, size_t fb_ints) {
    const int nsamp = (fpb - 1) * hop + NFFT;
    return sizeof(float) * ((size_t)2 * NFFT + ((nsamp + 3) & ~3) + 16 * FRAME_STRIDE + (size_t)fpb * (n_mels + 1) + fpb) +
           fb_ints * 4 + 16;
}
constexpr size_t LDS_TWO_PER_CU = 80 * 1024;  // two workgroups of this size share a CU

int build_tables(at_ctx* ctx, int sample_rate, int n_mels, int hop, const float* fb_user_dev, hipStream_t stream,
                 const float** tabs, const int** st, const int** ln, const int** of, const float** wt,
                 int* n_weights, int* quads) {
    bool cached = ctx->ws[WS_LOGMEL_FB] && ctx->fb_sr == sample_rate && ctx->fb_nfft == NFFT &&
                  ctx->fb_nmels == n_mels && (ctx->fb_user != nullptr) == (fb_user_dev != nullptr) && ctx->fb_hop == hop;
    std::vector<float> fb((size_t)NBIN * n_mels);
    if (fb_user_dev) {
        // A caller's filterbank is never trusted by address (allocators hand the same address to the next
        // tensor of the same shape, and a tensor can be rewritten in place): its VALUES are read back and
        // compared with the copy the resident tables were built from.  66 KB and one stream synchronisation
        // per call -- the price of the option; the library's own filterbank (fb_or_null == NULL) pays nothing.
        AT_HIP(hipStreamSynchronize(stream));
        AT_HIP(hipMemcpy(fb.data(), fb_user_dev, fb.size() * sizeof(float), hipMemcpyDeviceToHost));
        cached = cached && ctx->fb_user_copy && std::memcmp(ctx->fb_user_copy, fb.data(), fb.size() * sizeof(float)) == 0;
    }
    std::vector<int> start(n_mels), len(n_mels), off(n_mels);
    std::vector<float> wts;
    size_t bytes = 0;
    char* base = nullptr;
    if (!cached) {
        if (!fb_user_dev) {
            int rc = at_mel_filterbank_host(sample_rate, NFFT, n_mels, fb.data());
            if (rc) return rc;
        }
        // Bands padded with zero weights to whole 4-aligned quads of bins (16-byte LDS reads, see
        // logmel_core.h mel_band) -- unless the padding is what pushes a 32-frame workgroup past half
        // a CU's LDS, in which case the bands are stored as they are.
        const size_t nint0 = ((size_t)3 * n_mels + 3) & ~(size_t)3;
        for (int gran = 4; gran >= 1; gran -= 3) {
            wts.clear();
            for (int m = 0; m < n_mels; m++) {
                int lo = NBIN, hi = -1;
                for (int f = 0; f < NBIN; f++)
                    if (fb[(size_t)f * n_mels + m] != 0.0f) { lo = f < lo ? f : lo; hi = f; }
                const int s4 = hi < 0 ? 0 : (lo / gran) * gran;
                const int e4 = hi < 0 ? 0 : ((hi + gran) / gran) * gran;
                start[m] = s4;
                len[m] = (e4 - s4) / gran;
                off[m] = (int)wts.size();
                for (int f = s4; f < e4; f++) wts.push_back(f < NBIN ? fb[(size_t)f * n_mels + m] : 0.0f);
            }
            ctx->fb_quads = gran == 4;
            const bool fits = lds_bytes(32, hop, n_mels, nint0 + wts.size()) <= LDS_TWO_PER_CU;
            const bool lds_ok = (nint0 + wts.size()) * 4 <= 14 * 1024;
            if (gran == 1 || fits || !lds_ok) break;  // (a dense user filterbank stays in global memory: quads)
        }
    }
    // layout: [tabs TAB_FLOATS f32][start4 n_mels i32][quads][off][pad to 4 ints][wts ...]; worst case all bins
    const size_t nint = ((size_t)3 * n_mels + 3) & ~(size_t)3;
    const size_t cap = sizeof(float) * TAB_FLOATS + sizeof(int) * nint + sizeof(float) * (size_t)(NBIN + 3) * n_mels;
    if (!cached) {
        base = static_cast<char*>(at_ws(ctx, WS_LOGMEL_FB, cap, stream));
        if (!base) return AT_E_NOMEM;
        std::vector<float> t(TAB_FLOATS);
        for (int i = 0; i < NFFT; i++) t[TAB_WIN + i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / NFFT));
        for (int j = 0; j < 256; j++) {
            const int e = (j / 16) * (j % 16);  // layout [k1][m2] -> W256^(m2*k1) (logmel_core.h phase1)
            t[TAB_TW256 + 2 * j] = (float)std::cos(2.0 * M_PI * e / 256.0);
            t[TAB_TW256 + 2 * j + 1] = (float)-std::sin(2.0 * M_PI * e / 256.0);
            t[TAB_TW512 + 2 * j] = (float)std::cos(2.0 * M_PI * j / 512.0);
            t[TAB_TW512 + 2 * j + 1] = (float)-std::sin(2.0 * M_PI * j / 512.0);
        }
        std::vector<char> blob(sizeof(float) * TAB_FLOATS + sizeof(int) * nint + sizeof(float) * wts.size(), 0);
        char* q = blob.data();
        std::memcpy(q, t.data(), sizeof(float) * TAB_FLOATS); q += sizeof(float) * TAB_FLOATS;
        std::memcpy(q, start.data(), sizeof(int) * n_mels); q += sizeof(int) * n_mels;
        std::memcpy(q, len.data(), sizeof(int) * n_mels); q += sizeof(int) * n_mels;
        std::memcpy(q, off.data(), sizeof(int) * n_mels); q += sizeof(int) * n_mels;
        q = blob.data() + sizeof(float) * TAB_FLOATS + sizeof(int) * nint;
        if (!wts.empty()) std::memcpy(q, wts.data(), sizeof(float) * wts.size());
        bytes = blob.size();
        // a launch on ANY stream may still be reading the tables about to be replaced
        AT_HIP(hipDeviceSynchronize());
        AT_HIP(hipMemcpy(base, blob.data(), bytes, hipMemcpyHostToDevice));
        std::free(ctx->fb_user_copy);
        ctx->fb_user_copy = nullptr;
        if (fb_user_dev) {
            ctx->fb_user_copy = static_cast<float*>(std::malloc(fb.size() * sizeof(float)));
            if (!ctx->fb_user_copy) return at_fail(AT_E_NOMEM, "at_logmel_f32: out of host memory");
            std::memcpy(ctx->fb_user_copy, fb.data(), fb.size() * sizeof(float));
        }
        ctx->fb_sr = sample_rate; ctx->fb_nfft = NFFT; ctx->fb_nmels = n_mels; ctx->fb_user = fb_user_dev;
        ctx->fb_nw = (int)wts.size();
        ctx->fb_hop = hop;
    } else {
        base = static_cast<char*>(ctx->ws[WS_LOGMEL_FB]);
    }
    *tabs = reinterpret_cast<const float*>(base);
    const int* ints = reinterpret_cast<const int*>(base + sizeof(float) * TAB_FLOATS);
    *st = ints; *ln = ints + n_mels; *of = ints + 2 * n_mels;
    *wt = reinterpret_cast<const float*>(ints + nint);
    *n_weights = ctx->fb_nw;
    *quads = ctx->fb_quads;
    return AT_OK;
}

}  // namespace

static int logmel_impl(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L,
                       int64_t wave_stride, int sample_rate, int n_fft, int hop, int n_mels,
                       const float* fb_or_null, float* out, int layout, int fuse_l2norm,
                       unsigned* minmax, hipStream_t stream) {
    AT_REQUIRE(ctx, "at_logmel_f32: ctx is null");
    AT_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0,
               "at_logmel_f32: n_fft=%d not supported (a power of two from 64 to 4096)", n_fft);
    AT_REQUIRE(hop >= 1 && hop <= n_fft, "at_logmel_f32: hop=%d out of range [1, %d]", hop, n_fft);
    AT_REQUIRE(n_mels >= 1 && n_mels <= 1024, "at_logmel_f32: n_mels=%d out of range", n_mels);
    AT_REQUIRE(n_clips >= 0 && n_clips <= 65535 * 1024L, "at_logmel_f32: n_clips out of range");
    AT_REQUIRE(L > n_fft / 2, "at_logmel_f32: clip length %lld must exceed n_fft/2 (reflect padding)", (long long)L);
    AT_REQUIRE(wave_stride >= L, "at_logmel_f32: wave_stride < L");
    AT_REQUIRE(layout == AT_LAYOUT_MEL_MAJOR || layout == AT_LAYOUT_FRAME_MAJOR, "at_logmel_f32: bad layout");
    AT_REQUIRE(!fuse_l2norm || layout == AT_LAYOUT_FRAME_MAJOR, "at_logmel_f32: fuse_l2norm needs the frame-major layout");
    if (n_clips == 0) return AT_OK;
    AT_REQUIRE(wave && out, "at_logmel_f32: null pointer");
    AT_HIP(hipSetDevice(ctx->device));
    if (n_fft != NFFT) {   // the general form (logmel_any.hip); unit rows by the stand-alone kernel behind it
        AT_REQUIRE(at_num_frames(L, hop) < (1LL << 31), "at_logmel_f32: too many frames per clip");
        int rc = at_logmel_any(ctx, wave, n_clips, L, wave_stride, sample_rate, n_fft, hop, n_mels, fb_or_null, out,
                               layout == AT_LAYOUT_FRAME_MAJOR, stream);
        if (rc) return rc;
        if (fuse_l2norm) {
            int* bad = at_row_flag(ctx, stream);
            if (!bad) return AT_E_NOMEM;
            return at_l2norm_rows_flagged(ctx, out, n_clips * at_num_frames(L, hop), n_mels, out, bad, stream);
        }
        return AT_OK;
    }

    LogmelParams p;
    int rc = build_tables(ctx, sample_rate, n_mels, hop, fb_or_null, stream, &p.tabs, &p.fb_start, &p.fb_len,
                          &p.fb_off, &p.fb_wts, &p.fb_nw, &p.fb_quads);
    if (rc) return rc;
    const int64_t T = at_num_frames(L, hop);
    AT_REQUIRE(T < (1LL << 31), "at_logmel_f32: too many frames per clip");
    p.wave = wave; p.n_clips = n_clips; p.L = L; p.wave_stride = wave_stride;
    p.hop = hop; p.T = (int)T; p.n_mels = n_mels;
    // unit rows are fused for 8 <= n_mels <= 128 (numpy's one-level pairwise sum, eight lanes per frame);
    // other widths get the standalone kernel behind this one, in place
    const bool fuse_here = fuse_l2norm && n_mels >= 8 && n_mels <= 128;
    p.out = out; p.frame_major = layout == AT_LAYOUT_FRAME_MAJOR; p.fuse_l2norm = fuse_here;
    p.bad = fuse_l2norm ? at_row_flag(ctx, stream) : nullptr;
    if (fuse_l2norm && !p.bad) return AT_E_NOMEM;
    p.minmax = minmax;
    // the banded filterbank rides in LDS too unless a dense user filterbank makes it too big
    const size_t fb_ints = (((size_t)3 * n_mels + 3) & ~(size_t)3) + p.fb_nw;
    p.fb_lds = fb_ints * 4 <= 14 * 1024;
    // 32 frames per workgroup when two workgroups of that size still share a CU's 160 KiB of LDS
    // (the kernel is latency-bound: one workgroup per CU runs at half the rate), else 16
    size_t lds = 0;
    for (p.fpb = 32; p.fpb >= 16; p.fpb -= 16) {
        lds = lds_bytes(p.fpb, hop, n_mels, p.fb_lds ? fb_ints : 0);
        if (lds <= LDS_TWO_PER_CU || p.fpb == 16) break;
    }
    AT_REQUIRE(lds <= 160 * 1024, "at_logmel_f32: n_mels=%d needs %zu bytes of LDS", n_mels, lds);
    const bool pf = ((p.fpb - 1) * hop + NFFT + 3) / 4 <= PREFETCH_REGS * WG;   // the block's samples fit the prefetch registers
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&logmel_kernel<true>), lds); if (rcl_) return rcl_; }
    { const int rcl_ = at_raise_lds(ctx, reinterpret_cast<const void*>(&logmel_kernel<false>), lds); if (rcl_) return rcl_; }
    p.blocks_per_clip = (int)((T + p.fpb - 1) / p.fpb);
    p.n_blocks = (long)p.blocks_per_clip * n_clips;
    // persistent workgroups: two per CU (what the LDS footprint allows), each walking a strided share
    // of the blocks with its window / twiddle tables in registers
    long grid = 2L * ctx->n_cus;
    if (grid > p.n_blocks) grid = p.n_blocks;
    if (pf) AT_LAUNCH(logmel_kernel<true>, dim3((unsigned)grid), dim3(WG), lds, stream, p);
    else AT_LAUNCH(logmel_kernel<false>, dim3((unsigned)grid), dim3(WG), lds, stream, p);
    if (fuse_l2norm && !fuse_here) return at_l2norm_rows_flagged(ctx, out, n_clips * T, n_mels, out, p.bad, stream);
    return AT_OK;
}

extern "C" int at_logmel_f32(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L,
                             int64_t wave_stride, int sample_rate, int n_fft, int hop, int n_mels,
                             const float* fb_or_null, float* out, int layout, int fuse_l2norm,
                             void* stream_) {
    return logmel_impl(ctx, wave, n_clips, L, wave_stride, sample_rate, n_fft, hop, n_mels, fb_or_null, out, layout,
                       fuse_l2norm, nullptr, (hipStream_t)stream_);
}

// MelSpectrogram + AmplitudeToDB + normalize_spectrogram (processors/spectrogram_generator.py:123-131 with
// config.normalize = True): the log-mel kernel collects every clip's smallest and largest dB value while the block it
// has just computed is still in LDS, so the scaling is ONE pass over the spectrogram (8 B per value) instead of a
// reduction pass plus a scaling pass (12 B).  Other n_fft than 512: the general log-mel kernel, then the two-pass form.
extern "C" int at_logmel_minmax_f32(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride,
                                    int sample_rate, int n_fft, int hop, int n_mels, const float* fb_or_null, float* out,
                                    int layout, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AT_REQUIRE(ctx, "at_logmel_minmax_f32: ctx is null");
    if (n_clips == 0) return AT_OK;
    AT_REQUIRE(hop >= 1, "at_logmel_minmax_f32: hop=%d out of range", hop);
    const int64_t clip_elems = at_num_frames(L, hop) * n_mels;
    if (n_fft != NFFT) {
        int rc = logmel_impl(ctx, wave, n_clips, L, wave_stride, sample_rate, n_fft, hop, n_mels, fb_or_null, out, layout, 0,
                             nullptr, stream);
        if (rc) return rc;
        return at_minmax_scale_clips_f32(ctx, out, n_clips, clip_elems, stream_);
    }
    AT_REQUIRE(n_clips <= 65535, "at_logmel_minmax_f32: at most 65535 clips per call");
    AT_HIP(hipSetDevice(ctx->device));
    unsigned* mm = static_cast<unsigned*>(at_ws(ctx, WS_LOGMEL_MINMAX, (size_t)n_clips * 16, stream));
    if (!mm) return AT_E_NOMEM;
    AT_LAUNCH(minmax_init_kernel, dim3((unsigned)((n_clips + 255) / 256)), dim3(256), 0, stream, mm, (long)n_clips);
    int rc = logmel_impl(ctx, wave, n_clips, L, wave_stride, sample_rate, n_fft, hop, n_mels, fb_or_null, out, layout, 0, mm,
                         stream);
    if (rc) return rc;
    int bx = (int)((clip_elems + 256 * 8 - 1) / (256 * 8));
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    AT_LAUNCH(minmax_apply_kernel, dim3((unsigned)bx, (unsigned)n_clips), dim3(256), 0, stream, out, (long)clip_elems, mm);
    return AT_OK;
}
