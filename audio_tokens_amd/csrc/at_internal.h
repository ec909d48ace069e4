// at_internal.h -- shared by the translation units of libaudio_tokens_amd.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <tuple>
#include <utility>

#include "../../include/audio_tokens_amd.h"
#include "../../include/at_debug.h"

// Kernels that a caller may put on a stream of its own beside the k-means / tokenise sweeps (the pipeline computes the
// log-mel frames of later batches on the context's background stream) are compiled WITHOUT packed-fp32 vector
// instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32): Makefile, NOPK_OBJS.  Measured on gfx950
// (ROCm 7.2.0, round 3, profiles/r03_packed_fp32_beside_mfma.txt): while a wave of another kernel keeps the same SIMD's
// matrix pipe busy with back-to-back v_mfma_f32_32x32x16_f16 (the fp16 filter in guess mode: three products per tile), the
// packed ops of the 512-point log-mel kernel returned wrong values in lanes 48..63 of single registers -- 1.3e4 wrong
// frames of 1.0e7, always the frame owned by the wave's last 16 lanes, with either register allocation (223 / 190 VGPRs),
// with or without an s_waitcnt behind every LDS access, at 32 or 16 frames per block; the same source compiled without
// packed ops gave 0 wrong frames under the same load, at the same speed.  The form that fails (tools/pk_probe.py: one
// packed instruction per victim kernel, checked in place against the unpacked one) is v_pk_add_f32 with an op_sel half-swizzle
// of a vector-register source -- what the compiler emits for complex arithmetic; plain, negated, scalar- and constant-operand
// forms showed 0 mismatches in 1e11 operations each.  The sweeps' own packed ops have never differed
// from the dense reference (bench.py `verified`, tests/test_gpu_*) and produce the same guesses with and without, so the
// rule is applied to the producers only.  (A target("no-packed-fp32-ops") attribute on the kernels does the same but keeps
// the always-inline helpers from being inlined into them -- 30 calls and a scratch frame in the log-mel kernel --, and
// -Xarch_device -mno-packed-fp32-ops is accepted and ignored.)

// Workspace slots of a context (grown on demand, never shrunk).
enum at_ws_slot {
    WS_CENT_IMG = 0,   // assign: tiled/swizzled centroid image + |c|^2
    WS_SORT_KEYS_A,    // centroid_accum: keys / values, double buffered, + rocprim temp storage
    WS_SORT_KEYS_B,
    WS_SORT_VALS_A,
    WS_SORT_VALS_B,
    WS_SORT_TMP,
    WS_SEG_OFFSETS,    // centroid_accum: k+1 segment starts
    WS_REDUCE,         // at_sum_f32 / at_any_nonfinite partials
    WS_LOGMEL_FB,      // log-mel: banded filterbank tables
    WS_PRUNE_BD,       // pruned sweep: exact distance to the guess, per visiting position
    WS_PRUNE_MASK,     // pruned sweep: per 32-row tile, one bit per 32-centroid group
    WS_PRUNE_STATS,    // pruned sweep: {accumulators computed, accumulators of the dense sweep}
    WS_RESAMPLE_TAPS,  // resampler: polyphase filter taps [new][2*width + orig]
    WS_CENT_IMG16,     // fp16-split filter: centroid fragments (hi, lo) per group + |c|^2 + indices
    WS_FILTER_LIST,    // fp16-split filter: listed positions, sorted copy, order/hint of the redo pass
    WS_VISIT_KEYS_A,   // at_visit_order_f32: its own sort buffers (it may run while an accumulation still reads the others)
    WS_VISIT_KEYS_B,
    WS_VISIT_VALS_A,
    WS_VISIT_VALS_B,
    WS_VISIT_TMP,
    WS_CENT_IMG16B,    // fused coarse pass: fp16 image of the group means
    WS_IOTA,           // fused coarse pass: identity permutation of the means
    WS_FILTER_MISC,    // fp16-split filter: max|c|^2 bits, list length; running totals for at_filter_stats
    WS_PERM_RAW,       // at_rand_perm_prefix_device: draws, (position, step) sort buffers, links
    WS_PERM_KEYS_A,
    WS_PERM_KEYS_B,
    WS_PERM_VALS_A,
    WS_PERM_VALS_B,
    WS_PERM_TMP,
    WS_PERM_PREV,
    WS_PERM_LAST,
    WS_SPLIT_LIST,     // at_split_clusters_f32: the clusters that came out empty
    WS_TSTAT_IOTA,     // at_token_stats_f64: token ids before the sort, rocprim temp storage
    WS_TSTAT_TMP,
    WS_MT_RAW,         // at_mt_cached_draws: the resident start of the mt19937(1234) stream + the state behind it
    WS_LONG_PRED,      // centroid_accum: the long clusters of the last call (the next call's early set) + generation marks
    WS_LONG_EARLY,     // centroid_accum: block counts / bases and the member lists of the early set
    WS_LONG_LATE,      // centroid_accum: long clusters left to the pass behind the sort
    WS_BUCKETS,        // centroid_accum bucket path: counts, cursors, the list of long clusters
    WS_LOGMEL_ANY,     // log-mel, general n_fft: window, twiddles, banded filterbank
    WS_SUM_TICKET,     // at_sum_f32: arrival counter of its workgroups (the last one adds the partials up), zero between calls
    WS_ROW_FLAG,       // one int: a unit-row pass of at_logmel_f32 met a row whose squared norm is not finite
    WS_LOGMEL_MINMAX,  // at_logmel_minmax_f32: per clip {min key, max key, NaN flag, pad}
    WS_FILTER_BLKSTATS, // fp16-split filter: one statistics record per workgroup of a sweep (switch filter_stats)
    WS_NSLOTS
};

// Asynchronous exact calls leave their statistics words (and the events that time their stage-1 kernel) in a
// ring of slots; slots are folded into the totals when their copy has arrived -- polled, never waited for, unless
// the ring is full or a query asks for the totals.
constexpr int AT_FILTER_RING = 64;
struct at_filter_slot {
    unsigned* host_misc;   // pinned, 128 words
    hipEvent_t copied;     // behind the D2H copy of the words
    hipEvent_t ev[2];      // around the stage-1 kernel (created on first use, and only under the switch filter_timing)
    int timed;             // both of ev[] were recorded by the call that holds this slot (else they are not read)
    int64_t rows;
};

// Development switches (include/at_debug.h).  Read from the environment ONCE, in at_create -- never on a call
// path -- and changed afterwards only through at_debug_set.  Every one selects another route to the same bits
// (tests/test_gpu_ops.py::test_ab_switches_leave_the_bits_alone).
struct at_debug {
    int assign_variant;   // AT_ASSIGN_VARIANT   shape of the dense fp32 sweep (0 = default)
    int filter_fused;     // AT_FILTER_FUSED     1 = pre-pass inside the filter sweep (default), 0 = separate kernel
    int filter_sync;      // AT_FILTER_SYNC      1 = exact calls always take the synchronous form
    int prune_kernel;     // AT_PRUNE_KERNEL     0 = LDS-DMA form of the d = 64 fp32 pruned sweep
    int prune_nb;         // AT_PRUNE_NB         row tiles per wave of the fp32 pruned sweep (1, 2, 4; 0 = default)
    int filter_screen;    // AT_FILTER_SCREEN    0 = always evaluate all three fp16 products
    int filter_nb;        // AT_FILTER_NB        row tiles per wave of the filter sweep (1, 2, 4; 0 = by size)
    int filter_wps2;      // AT_FILTER_WPS2      1 = two waves per SIMD in the Lloyd-sized filter sweeps
    int dmin_kernel;      // AT_DMIN_KERNEL      0 = fp32 vector-ALU kernel for the centroid-to-group bounds
    int resample_simple;  // AT_RESAMPLE_SIMPLE  1 = one-thread-per-sample resampler
    int visit_bits;       // AT_VISIT_BITS       distance bits of the visiting-order key (0..8; default 8)
    int filter_stats;     // AT_FILTER_STATS     1 = the sweeps count accumulators / tiles for at_prune_stats, at_filter_stats
    int accum_buckets;    // AT_ACCUM_BUCKETS    0 = member lists by radix sort (the only form for k > 16 384)
    int filter_timing;    // AT_FILTER_TIMING    1 = exact calls bracket their stage-1 kernel with two timing events (bench.py; off in the product)
};

struct at_ctx {
    int device;
    at_debug dbg;
    void* ws[WS_NSLOTS];
    size_t ws_bytes[WS_NSLOTS];
    // cached description of what WS_LOGMEL_FB currently holds
    int fb_sr, fb_nfft, fb_nmels, fb_nw, fb_hop, fb_quads;
    const float* fb_user;
    hipEvent_t mt_ready;   // behind the generation of WS_MT_RAW
    int mt_have;
    uint32_t mt_seed;
    int any_sr, any_nfft, any_nmels;   // what WS_LOGMEL_ANY holds
    float* any_user_copy;
    float* fb_user_copy;   // host copy of the user filterbank the tables were built from (malloc'd; compared per call)
    int n_cus;             // multiProcessorCount of the device (read once in at_create)
    int rs_orig, rs_new;  // what WS_RESAMPLE_TAPS currently holds
    int64_t filter_rows, filter_listed;  // fp16-split filter: rows swept / rows handed to the fp32 redo
    int filter_slot;                     // the ring slot (or AT_FILTER_RING, the spare) the sweep being queued belongs to
    double filter_ms;                    // summed stage-1 kernel time, over filter_launches launches
    int64_t filter_launches;
    int64_t filter_tiles, filter_refined;
    // asynchronous exact calls: the statistics words of the last call are copied to pinned memory and folded
    // into the totals at the next call / query; a call whose list was long switches the context to the
    // synchronous form (with its fp32 MFMA redo) from then on
    unsigned* filter_host_misc;          // pinned, (AT_FILTER_RING + 1) x 64 words
    at_filter_slot fring[AT_FILTER_RING + 1];   // (the extra slot lends its timing events to the synchronous form)
    int fring_head, fring_count;         // oldest pending slot, number of pending slots
    int filter_force_sync;
  // 32x32 tiles multiplied (hi*hi) / refined (lo products too), exact calls
    hipStream_t side_stream;             // centroid_accum: long member lists beside the short ones
    hipStream_t background_stream;       // at_background_stream: lowest priority, lent to the caller
    hipEvent_t side_ev[2];
    hipEvent_t side_ev2;                 // behind the segment offsets computed beside the sort (centroid_accum, radix path)
    int defer_join, join_pending;        // at_centroid_accum_defer / at_centroid_accum_join
    int buckets_k;                       // table size WS_BUCKETS was last zeroed for
    int long_pred_k;                     // table size WS_LONG_PRED was last used with
    unsigned long_gen;                   // generation of its 'summed early' marks
    // what at_group_min_dist_f32 left in WS_CENT_IMG16 / WS_FILTER_MISC[0]; a sweep may reuse it when its
    // caller vouches (prepass_done bit 1) that the centroids are unchanged
    const float* img16_c;
    const int32_t* img16_cperm;
    const unsigned* img16_misc;
    int img16_k, img16_d, img16_ng, img16_trusted;
    int img16_misc_clean;   // the words behind max|c|^2 in img16_misc are still zero (nobody has swept since they were cleared)
    // dynamic-LDS limits raised so far ON THIS DEVICE (hipFuncAttributeMaxDynamicSharedMemorySize is per device: a
    // process-wide "already raised" flag would leave the second device of a process at the default limit)
    struct { const void* func; size_t bytes; } lds_raised[48];
    int n_lds_raised;
    hipEvent_t sum_ev;        // behind the last at_sum_f32 launch (its partials and counter are per context)
    hipStream_t sum_stream;
    int sum_used;
};

// makes launches of `func` with `bytes` of dynamic LDS legal on the context's device (at most one runtime call per
// kernel and size step; nothing below the default limit)
int at_raise_lds(at_ctx* ctx, const void* func, size_t bytes);

int at_fail(int code, const char* fmt, ...);
// randperm.hip: resident prefix of the mt19937(seed) stream (6722 blocks of 624 outputs: covers the 4 194 304
// swaps of the largest subsample of BASELINE.json and ~500 cluster repairs at k = 8192)
constexpr int64_t AT_MT_CACHE_DRAWS = 624LL * 6722;
int at_mt_cached_draws(at_ctx* ctx, uint32_t seed, hipStream_t stream, const uint32_t** raw, int64_t* raw_n,
                       const uint32_t** state_end);
// Returns a device buffer of at least `bytes` for `slot` (contents undefined after growth).
void* at_ws(at_ctx* ctx, int slot, size_t bytes, hipStream_t stream);

// ---- error handling ----------------------------------------------------------------------------------------
// On ROCm 7.2 the error a failing HIP call leaves behind is STICKY: it stays in the calling thread until somebody
// calls hipGetLastError(), whatever succeeds in between (measured: tools/probes/event_probe.hip; hipErrorNotReady is
// the one code that is not kept).  Round 2 checked launches with a bare hipGetLastError() behind them, which
// therefore reported whatever any earlier call of this thread -- the library's, torch's, rocPRIM's -- had failed
// with and nobody had consumed ("kernel launch failed: invalid resource handle" out of a launch that was fine).
// Rules since round 3:
//   * every call whose failure is an error goes through AT_HIP: the call's OWN return code decides;
//   * every launch goes through AT_LAUNCH: hipLaunchKernel's OWN return code decides, never the thread's last error;
//   * a call whose failure is tolerated goes through AT_HIP_TOLERATE, which consumes the sticky error the failure
//     left (so it cannot surface in the host application's, or rocPRIM's, next hipGetLastError) and counts it;
//   * an error that is already pending when a launch is about to be made is not this library's launch failing:
//     it is consumed, counted and remembered (at_diag_*), and fails the call only under the switch strict_errors
//     ("stale error from an earlier call"), which the test-suite turns on.
struct at_diag_counters {
    long stale_seen;          // errors found pending in front of a launch
    int stale_last_code;
    const char* stale_last_file;
    int stale_last_line;
    long tolerated;           // failing calls whose failure was tolerated (AT_HIP_TOLERATE)
    int tolerated_last_code;
    const char* tolerated_last_file;
    int tolerated_last_line;
};
extern at_diag_counters g_at_diag;          // host.cpp (one per process; the counts are diagnostics, not state)
extern int g_at_strict_errors;              // host.cpp: switch strict_errors (AT_STRICT_ERRORS), process-wide

hipError_t at_hip_tolerated(hipError_t e, const char* file, int line);
// returns hipSuccess, or -- strict mode -- the pending error
hipError_t at_stale_check(const char* file, int line);

#define AT_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (void)hipGetLastError(); /* reported here: do not leave it pending as well */  \
            return at_fail(AT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                            \
        }                                                                                  \
    } while (0)

// the call's code, with the sticky copy of a failure consumed and counted
#define AT_HIP_TOLERATE(expr) at_hip_tolerated((expr), __FILE__, __LINE__)

// Kernel launch by hipLaunchKernel, whose return value is this launch's own verdict.  Arguments are converted to the
// kernel's parameter types first (what the <<< >>> stub does), then passed by address.
template <typename... KArgs, typename... Args>
static inline hipError_t at_launch_raw(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                       Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "at_launch_raw: argument count does not match the kernel's");
    std::tuple<KArgs...> params{static_cast<KArgs>(std::forward<Args>(args))...};
    void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1];
    int i = 0;
    std::apply([&](auto&... p) { ((ptrs[i++] = const_cast<void*>(static_cast<const void*>(&p))), ...); }, params);
    return hipLaunchKernel(reinterpret_cast<const void*>(kern), grid, block, ptrs, lds, stream);
}

#define AT_LAUNCH(kern, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                  \
        hipError_t s_ = at_stale_check(__FILE__, __LINE__);                                               \
        if (s_ != hipSuccess)                                                                             \
            return at_fail(AT_E_HIP, "stale error from an earlier call (pending before the launch of %s): %s (%s:%d)", \
                           #kern, hipGetErrorString(s_), __FILE__, __LINE__);                             \
        hipError_t e_ = at_launch_raw(kern, grid, block, lds, stream, __VA_ARGS__);                       \
        if (e_ != hipSuccess) {                                                                           \
            (void)hipGetLastError();                                                                      \
            return at_fail(AT_E_HIP, "launch of %s failed: %s (%s:%d)", #kern, hipGetErrorString(e_),     \
                           __FILE__, __LINE__);                                                           \
        }                                                                                                 \
    } while (0)

#define AT_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return at_fail(AT_E_INVALID, __VA_ARGS__); \
    } while (0)

// filter.hip (fp16-split filter of the pruned sweep)
int at_filter_sweep(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const uint32_t* order,
                    const int32_t* cperm, int ng, const float* bd, const uint32_t* mask, int ngw, int collect,
                    int64_t* ids, unsigned* misc, uint32_t* amb_list, uint32_t* amb_aux, float* approx_out,
                    const uint32_t* fuse_hint_sorted, const float* fuse_dmin, float* fuse_bd_out, float* fuse_dist_out,
                    unsigned amb_cap, hipStream_t stream);
// the rows a sweep lists for the redo: 64 sub-lists of amb_cap slots each (counters at misc[64 ..]), arrays of
// at_amb_stride(n) words
constexpr unsigned AT_AMB_SUBLISTS = 64;
static inline unsigned at_amb_cap(int64_t n) { return (unsigned)(n / AT_AMB_SUBLISTS + 128); }
static inline size_t at_amb_stride(int64_t n) { return (size_t)at_amb_cap(n) * AT_AMB_SUBLISTS; }
int at_amb_compact(at_ctx* ctx, const unsigned* misc, unsigned amb_cap, const uint32_t* list, const uint32_t* aux,
                   uint32_t* list_out, uint32_t* aux_out, hipStream_t stream);
int at_exact_dist_todo(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const int64_t* ids,
                       float* dist, hipStream_t stream);
int at_exact_dist_rows(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const int64_t* ids,
                       float* dist, const uint32_t* order, const uint32_t* hint_sorted, const float* bd,
                       hipStream_t stream);
int at_filter_gather_ambiguous(at_ctx* ctx, uint32_t* amb_list, uint32_t* amb_sorted, int64_t m_valid, int64_t m,
                               const uint32_t* order, const int64_t* ids, uint32_t* order_amb, uint32_t* hint_amb,
                               hipStream_t stream);

int at_filter_finish(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const uint32_t* list, int64_t redo_wgs,
                     const uint32_t* order, const int32_t* cperm, const float* dmin, int ng, const unsigned* misc,
                     const uint32_t* aux, int64_t* ids, float* dist, const unsigned* count_dev, unsigned amb_cap,
                     hipStream_t stream);
int at_filter_redo_rows(at_ctx* ctx, const float* x, int d, const float* c, int k, const uint32_t* list, int64_t m,
                        const uint32_t* order, const int32_t* cperm, const float* dmin, int ng, const unsigned* misc,
                        const uint32_t* aux, int64_t* ids, float* dist, const unsigned* count_dev, unsigned amb_cap,
                        hipStream_t stream);

// logmel_any.hip: every power-of-two n_fft other than 512
int at_logmel_any(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride, int sample_rate,
                  int n_fft, int hop, int n_mels, const float* fb_user_dev, float* out, int frame_major,
                  hipStream_t stream);

int at_group_min_dist_f16(at_ctx* ctx, const float* c, int k, int d, const int32_t* cperm, int ng, float* dmin,
                          hipStream_t stream);

int at_filter_coarse(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k, const int32_t* cperm, int ng,
                     const float* means, const uint32_t* gnbr, int64_t* ids, float* dist, hipStream_t stream);

// wait_all: fold every pending slot (blocking); otherwise only those whose copy has already arrived
int at_filter_resolve_pending(at_ctx* ctx, bool wait_all);
// the sweep queued next belongs to ring slot `slot` (AT_FILTER_RING: the spare, for calls outside the ring); creates
// what the slot needs on first use and marks its timing events as not recorded
int at_filter_use_slot(at_ctx* ctx, int slot);

static inline bool at_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// unit rows with a verdict: *bad (device int, may be null) is set when a row's norm is not finite (l2norm.hip)
int at_l2norm_rows_flagged(at_ctx* ctx, const float* x, int64_t n, int d, float* y, int* bad, hipStream_t stream);
int* at_row_flag(at_ctx* ctx, hipStream_t stream);   // the context's flag word (WS_ROW_FLAG), cleared when first allocated

