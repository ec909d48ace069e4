/*
 * at_debug.h -- development and test hooks of libaudio_tokens_amd.so.  NOT part of the operator surface
 * (include/audio_tokens_amd.h): nothing a binding of the reference's operators needs is declared here.
 *
 *   at_debug_set / at_debug_get   A/B switches of a context.  Each selects another route to the SAME bits
 *                                 (another kernel shape, a separate pre-pass, the synchronous form ...).  They are
 *                                 initialised from the AT_* environment variables once, in at_create, and never read
 *                                 from the environment again; names:
 *                                   assign_variant filter_fused filter_sync prune_kernel prune_nb filter_screen
 *                                   filter_nb filter_wps2 dmin_kernel resample_simple accum_buckets filter_stats visit_bits
 *                                   filter_timing strict_errors
 *                                 (filter_stats = 1 makes the fp16-split sweeps count for the two calls below: one record
 *                                 per workgroup and a small reduction kernel behind every sweep; off by default)
 *                                 filter_timing = 1 brackets the stage-1 kernel of every exact call with two timing events
 *                                 (what at_filter_stats' sweep_ms sums; bench.py turns it on, the product leaves it off);
 *                                 strict_errors (process-wide, also AT_STRICT_ERRORS): a HIP error found pending in the
 *                                 calling thread in front of one of the library's launches fails the call ("stale error
 *                                 from an earlier call") instead of being consumed and counted
 *   at_diag_errors                what the library met and did not treat as its own failure: errors pending in front of a
 *                                 launch (somebody's earlier call failed and nobody consumed the error) and failures it
 *                                 tolerates by design; counts, last codes and source positions.  No device work.
 *   at_debug_leave_error_pending  test hook: makes a HIP call fail without consuming its error, as a swallowed return code does
 *   at_prune_stats                running totals over the context's exact pruned sweeps: 32x32 accumulators computed /
 *                                 accumulators of the dense sweep.  Synchronises the device; reset != 0 clears them.
 *   at_filter_stats               fp16-split filter: rows swept / rows handed to the fp32 redo, the summed HIP-event
 *                                 time of the stage-1 kernel over `sweeps` exact calls, 32x32 tiles multiplied hi*hi /
 *                                 refined with the lo products (NULL skips a field).  Waits for the calls in flight.
 *   at_filter_probe_f32           stage 1 of an exact call only: approx[2i], approx[2i+1] = approximate |c|^2 - 2 x.c
 *                                 of row i's winner and its gap to the runner-up; *listed = rows it would hand on.
 */
#ifndef AUDIO_TOKENS_AMD_DEBUG_H
#define AUDIO_TOKENS_AMD_DEBUG_H

#include "audio_tokens_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

int at_debug_set(at_ctx* ctx, const char* name, int value);
int at_debug_get(const at_ctx* ctx, const char* name, int* value);
int at_diag_errors(int64_t* stale_seen, int* stale_last_code, int64_t* tolerated, int* tolerated_last_code, char* where,
                   int where_bytes, int reset);
int at_debug_leave_error_pending(void);

int at_prune_stats(at_ctx* ctx, int64_t* needed_host, int64_t* total_host, int reset);
int at_filter_stats(at_ctx* ctx, int64_t* rows, int64_t* listed, double* sweep_ms, int64_t* sweeps,
                    int64_t* tiles, int64_t* refined, int reset);
int at_filter_probe_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                        const uint32_t* order, const uint32_t* hint_sorted, const int32_t* cperm, int ng,
                        const float* dmin, int64_t* ids, float* approx, int64_t* listed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIO_TOKENS_AMD_DEBUG_H */
