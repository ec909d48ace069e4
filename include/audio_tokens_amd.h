/*
 * audio_tokens_amd.h -- C ABI of the MI355X (gfx950) audio-tokenisation hot path.
 *
 * The reference (danavery/audio-tokens) has no FFI of its own: its three stage classes call two
 * third-party operators directly.  The entry points below are what a native replacement of those
 * operators binds; each cites the reference call site it stands in for (paths relative to the
 * reference repository).
 *
 *   torchaudio.transforms.MelSpectrogram(...)(wave) -> AmplitudeToDB()      at_logmel_f32
 *       processors/spectrogram_generator.py:28-34, 123-126
 *   torchaudio.transforms.Resample(sr, 22050)(wave)                        at_resample_f32
 *       processors/spectrogram_generator.py:117-121
 *   np.linalg.norm(axis=1) row normalisation                               at_l2norm_rows_f32
 *       processors/cluster_creator.py:64-66, processors/spec_tokenizer.py:106-109
 *   faiss.IndexFlatL2(d).add(c); .search(x, 1)                             at_assign_f32
 *       processors/spec_tokenizer.py:77, 123-127 (and inside faiss.Kmeans.train)
 *   faiss.Kmeans(d, k, niter).train(x, init_centroids)                     at_rand_perm_mt19937,
 *       processors/cluster_creator.py:42-56                                at_gather_rows_f32,
 *                                                                          at_assign_f32, at_assign_hinted_f32,
 *                                                                          at_centroid_accum_f32,
 *                                                                          at_centroid_finalize_f32,
 *                                                                          at_split_clusters_host,
 *                                                                          at_sum_f32
 *
 * Conventions
 *   - plain C: raw pointers and sizes, no C++/torch types.  Every `const float*`/`float*` named
 *     x, c, wave, out, sums ... is a DEVICE pointer on the context's GPU unless the function name
 *     ends in _host or the parameter says (host).  `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  All device work is enqueued on that stream and is not
 *     synchronised unless stated.
 *   - the caller owns every buffer it passes; the library owns only the workspace behind
 *     at_ctx (grown on demand, freed by at_destroy).  A context is bound to one device and is
 *     not re-entrant: use one per host thread / stream.
 *   - return value: 0 = ok, negative = error (AT_E_*); at_last_error() gives the message of the
 *     calling thread's last failure.  Nothing throws across this boundary.
 */
#ifndef AUDIO_TOKENS_AMD_H
#define AUDIO_TOKENS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AT_VERSION 100 /* 0.1.0 */

#define AT_OK 0
#define AT_E_INVALID (-1)  /* bad argument (null pointer, size, unsupported shape) */
#define AT_E_HIP (-2)      /* a HIP runtime call failed */
#define AT_E_NOMEM (-3)    /* workspace allocation failed */
#define AT_E_TOO_FEW (-4)  /* faiss: "Number of training points should be at least ..." */
#define AT_E_NONFINITE (-5)/* faiss: "input contains NaN's or Inf's" */
#define AT_E_COMM (-6)     /* RCCL is not available in the process, or a collective failed */

/* at_logmel_f32 output layouts */
#define AT_LAYOUT_MEL_MAJOR 0   /* out[clip][n_mels][T]  -- the reference's .npy file layout */
#define AT_LAYOUT_FRAME_MAJOR 1 /* out[clip*T + t][n_mels] -- what ClusterCreator/SpecTokenizer
                                   build with np.load(f).T + np.concatenate */

typedef struct at_ctx at_ctx;

/* ---- context -------------------------------------------------------------------------------- */
int at_version(void);
const char* at_last_error(void);
int at_create(int device, at_ctx** out);
void at_destroy(at_ctx* ctx);
/* bytes of device workspace currently held by the context */
int64_t at_workspace_bytes(const at_ctx* ctx);
/* A stream of the LOWEST priority the device offers, owned by the context (created on first use, destroyed by
 * at_destroy): for work that should fill the gaps of the caller's main stream without delaying it -- the pipeline
 * computes the log-mel frames of the k-means batches to come there while the batch before them is trained
 * (cluster_creator.py:42-56 loads batch by batch too).  *stream_out is a hipStream_t. */
int at_background_stream(at_ctx* ctx, void** stream_out);

/* ---- host helpers (no GPU work) --------------------------------------------------------------*/
/* faiss utils/random.cpp rand_perm(perm, n, seed): std::mt19937 Fisher-Yates. perm: host [n]. */
int at_rand_perm_mt19937(int64_t n, int64_t seed, int32_t* perm_host);
/* The first m entries of that same permutation (m <= n) without running the remaining n-m
 * Fisher-Yates steps, which cannot touch them.  prefix_host: host [m]. */
int at_rand_perm_prefix_mt19937(int64_t n, int64_t seed, int64_t m, int32_t* prefix_host);
/* torchaudio.functional.melscale_fbanks(n_fft/2+1, 0, sr//2, n_mels, sr, norm=None, "htk").
 * fb_host: [n_fft/2+1][n_mels]. */
int at_mel_filterbank_host(int sample_rate, int n_fft, int n_mels, float* fb_host);
/* number of STFT frames of an L-sample clip with center=True: 1 + L / hop */
int64_t at_num_frames(int64_t L, int hop);
/* faiss Clustering.cpp split_clusters(d, k, n, 0, hassign, centroids) on HOST buffers
 * (hassign [k], centroids [k][d]); *nsplit receives the number of re-seeded clusters. */
int at_split_clusters_host(int d, int k, int64_t n, float* hassign_host, float* centroids_host,
                           int* nsplit);

/* ---- device operators ------------------------------------------------------------------------*/
/* Fused STFT -> |.|^2 -> mel -> 10*log10(max(.,1e-10)).
 *   wave: n_clips clips of L samples, clip i at wave + i*wave_stride (floats).
 *   fb_or_null: DEVICE [n_fft/2+1][n_mels] filterbank, or NULL = the library's own
 *   (at_mel_filterbank_host values).  n_fft: a power of two from 64 to 4096 (512, the reference's default, takes
 *   the tuned kernel; the others a general radix-2 form); 1 <= hop <= n_fft.
 *   out: n_clips*n_mels*T floats in `layout`.  fuse_l2norm != 0 (frame-major only) additionally
 *   applies at_l2norm_rows_f32 to every frame before it is stored. */
int at_logmel_f32(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride,
                  int sample_rate, int n_fft, int hop, int n_mels, const float* fb_or_null,
                  float* out, int layout, int fuse_l2norm, void* stream);

/* torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (sinc_interp_hann,
 * lowpass_filter_width 6, rolloff 0.99) -- processors/spectrogram_generator.py:117-121.
 * out: n_clips rows of at_resample_length(L, ...) samples, row stride out_stride floats.
 * at_resample_taps_host builds the polyphase taps [new][2*width + orig] on the host (pass
 * taps_host = NULL to query orig/new/width only). */
int64_t at_resample_length(int64_t L, int orig_freq, int new_freq);
int at_resample_taps_host(int orig_freq, int new_freq, int* orig_out, int* new_out, int* width_out,
                          float* taps_host, int64_t taps_capacity);
int at_resample_f32(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride,
                    int orig_freq, int new_freq, float* out, int64_t out_stride, void* stream);

/* y[i] = x[i] / (||x[i]||_2 + 1e-10), fp32, numpy's pairwise summation order (bit-exact with
 * numpy for finite inputs).  x == y is allowed. */
int at_l2norm_rows_f32(at_ctx* ctx, const float* x, int64_t n, int d, float* y, void* stream);

/* SpectrogramGenerator.normalize_spectrogram (processors/spectrogram_generator.py:129-131, config.normalize):
 * every clip of clip_elems floats (a [n_mels][T] spectrogram) becomes (x - min) / (max - min), in place, with torch's
 * fp32 operations (same bits as the reference's torch expression). */
int at_minmax_scale_clips_f32(at_ctx* ctx, float* x, int64_t n_clips, int64_t clip_elems, void* stream);

/* ClusterCreator.apply_convolution / SpecTokenizer.apply_convolution (processors/cluster_creator.py:68-81,
 * processors/spec_tokenizer.py:92-104,115-121; config.use_convolution): nn.Conv1d(1, num_kernels, kernel_size,
 * padding = (kernel_size - 1) / 2) along the mel axis of every frame.  x: [n][n_mels]; weight: DEVICE
 * [num_kernels][kernel_size] (the module's weight[:, 0, :]); bias_or_null: DEVICE [num_kernels];
 * out: [n][n_mels * num_kernels] with feature = mel * num_kernels + kernel (the reference's transpose + reshape).
 * out(m, j) = fma chain over the taps in ascending order, started from the bias. */
int at_conv1d_mel_f32(at_ctx* ctx, const float* x, int64_t n, int n_mels, const float* weight, const float* bias_or_null,
                      int num_kernels, int kernel_size, int padding, float* out, void* stream);

/* generate_mel_spectrogram followed by normalize_spectrogram (processors/spectrogram_generator.py:123-131 with
 * config.normalize = True) for a batch of clips: at_logmel_f32(..., fuse_l2norm = 0) and at_minmax_scale_clips_f32
 * in one call, same bits.  The log-mel kernel collects every clip's extremes while a computed block is still in LDS,
 * so the scaling costs one pass over the spectrogram instead of a reduction pass plus a scaling pass.  At most
 * 65535 clips per call. */
int at_logmel_minmax_f32(at_ctx* ctx, const float* wave, int64_t n_clips, int64_t L, int64_t wave_stride,
                         int sample_rate, int n_fft, int hop, int n_mels, const float* fb_or_null,
                         float* out, int layout, void* stream);

/* Nearest centroid under squared L2 (IndexFlatL2.search(x, 1)):
 *   dis(i,j) = max(0, (|x_i|^2 + |c_j|^2) - 2 <x_i, c_j>), all fp32, inner products and norms as
 *   ascending-index fmaf chains (v_mfma_f32_32x32x2_f32); ids[i] = lowest j attaining the minimum.
 *   n < 20 uses faiss's small-batch form sum (x-c)^2 instead (distance_compute_blas_threshold).
 *   ids: int64 [n]; dist_or_null: float [n]. */
int at_assign_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                  int64_t* ids, float* dist_or_null, void* stream);

/* The same answer as at_assign_f32, bit for bit, computed faster when most rows come with a correct
 * guess (a Lloyd iteration guessing the previous assignment).  Guesses are values in [0, k), or
 * anything else for "none", given either per row (hint_ids: int64 [n]) or per visiting position
 * (hint_sorted: uint32 [n], the guess of row order[p] at position p; needs order).  order_or_null:
 * uint32 [n], a permutation of the rows that groups equal guesses.  at_centroid_accum_f32 produces
 * both arrays for the ids it was given (order_out, sorted_ids_out).  ids must not alias hint_ids.
 * Results do not depend on the hints or the order. */
int at_assign_hinted_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                         const int64_t* hint_ids_or_null, const uint32_t* order_or_null,
                         const uint32_t* hint_sorted_or_null, int64_t* ids, float* dist_or_null,
                         void* stream);

/* ---- exact pruning for Lloyd iterations (none of this changes any result) ----------------------
 * at_group_rows_kd_host: spatial grouping of the centroid table into groups of `leaf` rows (host
 *   arrays; perm_out: ceil(k/leaf)*leaf entries, -1 padded).
 * at_group_min_dist_f32: dmin[p][g] (DEVICE float [k][ng]) = a lower bound of the distance from
 *   centroid p to the nearest member of group g (groups of 32 as listed in cperm, DEVICE int32).
 * at_visit_order_f32: rows sorted by (previous id, previous distance): order_out and the ids in
 *   that order (both DEVICE uint32 [n]).
 * at_group_means_f32: means[g] (DEVICE float [ng][d]) = mean row of group g.
 * at_group_neighbours_f32: gnbr[g] (DEVICE uint32 [ng][ceil(ng/32)]) = bit set of the nnb groups whose
 *   means are nearest to mean g, g itself included (ng <= 512, d <= 128); input of the guess generators.
 * at_assign_pruned_f32 (arguments: struct at_pruned_args below).  guess_only = 0: the answer of at_assign_f32, bit
 *   for bit (d = 64 or 128, n >= 20, ng <= 512).  Every row's guess hint_sorted[p] (for row order[p]) is scored
 *   exactly first; a 32-centroid group is then skipped for a 32-row tile when the triangle inequality, with a margin
 *   that covers the fp32 rounding of the distances, rules it out for all of the tile's rows.
 *   guess_only = 1 (a guess generator, NOT exact): hint_sorted holds a GROUP id per row; each row gets the best
 *   centroid among the groups named by its 32-row tile; `bounds` is then read as an optional uint32
 *   [ng][ceil(ng/32)] table: bit set = group worth searching for a row naming g.  The unguided exact search is
 *   nearest group mean -> guess_only = 1 -> guess_only = 0 with those answers as guesses.
 *   use_filter != 0: stage 1 is the fp16-split filter (csrc/filter.hip: three fp16 MFMAs per fp32 one, a winner
 *   accepted only when the runner-up is provably out of reach of the fp32 contract), the remaining rows are redone in
 *   fp32: ids/dist are bit-identical to use_filter = 0, and the call needs no host round trip (the redo kernel reads
 *   the list length on the device) unless earlier calls on this context listed more than n/16 rows.
 *   image_current != 0: the caller vouches that the fp16 image at_group_min_dist_f32 built is still current (same c,
 *   cperm; centroids unchanged since), so it is reused. */
int at_group_rows_kd_host(const float* rows_host, int k, int d, int leaf, int32_t* perm_out_host);
int at_group_min_dist_f32(at_ctx* ctx, const float* c, int k, int d, const int32_t* cperm, int ng,
                          float* dmin, void* stream);
int at_visit_order_f32(at_ctx* ctx, const int64_t* ids, const float* dis_or_null, int64_t n, int k,
                       uint32_t* order_out, uint32_t* hint_sorted_out, void* stream);
int at_group_means_f32(at_ctx* ctx, const float* c, int k, int d, const int32_t* cperm, int ng,
                       float* means, void* stream);
int at_group_neighbours_f32(at_ctx* ctx, const float* means, int ng, int d, int nnb, uint32_t* gnbr,
                            void* stream);
typedef struct at_pruned_args {
    const float* x;              /* DEVICE [n][d] rows */
    int64_t n;
    int d;
    const float* c;              /* DEVICE [k][d] centroids */
    int k;
    const uint32_t* order;       /* DEVICE [n]: rows in visiting order (at_visit_order_f32) */
    const uint32_t* hint_sorted; /* DEVICE [n]: the guess of row order[p] (guess_only: its group) */
    const int32_t* cperm;        /* DEVICE [ng*32]: spatial grouping (at_group_rows_kd_host), -1 padded */
    int ng;
    const float* bounds;         /* DEVICE: dmin [k][ng] (at_group_min_dist_f32); guess_only: optional bit table */
    int guess_only;              /* 0 = exact search, 1 = guess generator */
    int use_filter;              /* 0 = fp32 pruned sweep, 1 = fp16-split filter first (same bits) */
    int prepass_done;            /* 1 = at_prune_mask_f32 ran just before with the same arguments */
    int image_current;           /* 1 = the fp16 image of (c, cperm) left by at_group_min_dist_f32 is still valid */
    int64_t* ids;                /* DEVICE [n] out */
    float* dist_or_null;         /* DEVICE [n] out */
} at_pruned_args;
int at_assign_pruned_f32(at_ctx* ctx, const at_pruned_args* args, void* stream);
/* The pre-pass of at_assign_pruned_f32 alone (per-row bound and per-tile group masks into the
 * context's workspace).  at_assign_pruned_f32 runs it itself unless prepass_done != 0, in which case
 * it must directly follow this call with the same arguments on the same stream. */
int at_prune_mask_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                      const uint32_t* order, const uint32_t* hint_sorted, int ng,
                      const float* dmin_or_null, int mode, void* stream);

/* (Counters of the pruned / filtered sweeps and the stage-1 test hook live in at_debug.h.) */
/* Guess generator for rows in their own (coherent) order -- consecutive frames of clips, i.e. tokenise:
 * nearest of the ng group means (means [ng, d], at_group_means_f32) -> the groups the neighbour table
 * gnbr [ng][ceil(ng/32)] names -> best centroid among them, one launch, rows read once.  ids are guesses
 * for at_visit_order_f32 / at_assign_pruned_f32; dist (may be NULL) approximate distances. */
int at_assign_coarse_f32(at_ctx* ctx, const float* x, int64_t n, int d, const float* c, int k,
                         const int32_t* cperm, int ng, const float* means, const uint32_t* gnbr,
                         int64_t* ids, float* dist, void* stream);


/* The same m entries as at_rand_perm_prefix_mt19937(n, seed, m, .), bit for bit, computed on the device
 * (csrc/randperm.hip: the mt19937 stream by one workgroup, then the Fisher-Yates prefix resolved with a
 * sort of the swap partners instead of 2 M dependent host cache misses).  prefix: DEVICE int32 [m].  Uses
 * workspace of its own, so it may run on a stream beside other calls on the same context. */
int at_rand_perm_prefix_device(at_ctx* ctx, int64_t n, int64_t seed, int64_t m, int32_t* prefix, void* stream);

/* out[i] = x[idx[i]] (rows of d floats).  idx: DEVICE int32 [m]. */
int at_gather_rows_f32(at_ctx* ctx, const float* x, int d, const int32_t* idx, int64_t m,
                       float* out, void* stream);

/* faiss compute_centroids, accumulation half: sums[c] = sum of x[i] with ids[i] == c taken in
 * ASCENDING i with fp32 adds (bitwise what a single FAISS thread produces), counts[c] = number of
 * members (exact in fp32).  sums [k][d], counts [k] are overwritten.  order_out_or_null: uint32 [n],
 * receives the rows sorted by (id, row) -- the member lists, back to back; sorted_ids_out_or_null:
 * uint32 [n], the id of each of those rows (k for an id outside [0, k)). */
int at_centroid_accum_f32(at_ctx* ctx, const float* x, int64_t n, int d, const int64_t* ids, int k,
                          float* sums, float* counts, uint32_t* order_out_or_null,
                          uint32_t* sorted_ids_out_or_null, void* stream);

/* faiss compute_centroids, scaling half, over n_parts partial results (p = data-parallel rank):
 * part p has its sums [k][d] at sums_parts + p*sums_part_stride and its counts [k] at
 * counts_parts + p*counts_part_stride (strides in floats).  Partials are added in ascending p,
 * then centroids[c] = sum * (1/count) where count > 0 and 0 where the cluster is empty;
 * hassign[c] = count. */
/* Long member lists are accumulated on a side stream of the context.  By default at_centroid_accum_f32
 * makes `stream` wait for it before returning; after at_centroid_accum_defer(ctx, 1) that wait is left to
 * at_centroid_accum_join(ctx, stream), which must precede any use of sums / counts -- work queued on
 * `stream` in between overlaps the long lists. */
int at_centroid_accum_defer(at_ctx* ctx, int on);
int at_centroid_accum_join(at_ctx* ctx, void* stream);

int at_centroid_finalize_f32(at_ctx* ctx, const float* sums_parts, int64_t sums_part_stride,
                             const float* counts_parts, int64_t counts_part_stride, int n_parts,
                             int k, int d, float* centroids, float* hassign, void* stream);

/* out[i] = parts[0][i] + parts[1][i] + ... added in ascending part order (part p at parts + p*part_stride floats, m
 * floats each): the fixed-order reduction of the data-parallel exchange when it runs as all-to-all + local sum +
 * all-gather instead of an all-gather of whole partials (same bits as at_centroid_finalize_f32 over all parts). */
int at_sum_parts_f32(at_ctx* ctx, const float* parts, int64_t part_stride, int n_parts, int64_t m, float* out,
                     void* stream);

/* The exchange itself, for hosts that do not go through torch.distributed (the Python layer does: ops._Dist):
 * thin wrappers over RCCL on the CALLER's communicator (`nccl_comm` is an ncclComm_t; RCCL is looked up at run time,
 * the library has no link-time dependency on it).  at_comm_allgather_f32: parts [n_ranks][count] <- every rank's
 * part [count].  at_comm_allreduce_ordered_f32: out [count] <- the parts added in ascending rank order (all-gather into
 * parts_scratch [n_ranks * count], then at_sum_parts_f32): the all-reduce of the sharded Lloyd iteration, the same
 * bits on every rank and in the oracle's n_shards mode -- which ncclAllReduce(sum) does not promise. */
int at_comm_allgather_f32(at_ctx* ctx, void* nccl_comm, const float* part, float* parts, int64_t count, void* stream);
int at_comm_allreduce_ordered_f32(at_ctx* ctx, void* nccl_comm, const float* part, float* parts_scratch, float* out,
                                  int64_t count, void* stream);

/* at_split_clusters_host on DEVICE buffers (hassign [k], centroids [k][d], both updated in place), same bits:
 * one workgroup regenerates the mt19937(1234) stream in LDS and runs the cyclic acceptance scans, so a Lloyd
 * iteration needs no host round trip to learn whether a cluster came out empty.  *nsplit_out (DEVICE int32)
 * receives the number of re-seeded clusters (-1: no donor exists, i.e. every cluster has at most one member). */
int at_split_clusters_f32(at_ctx* ctx, int d, int k, int64_t n, float* hassign, float* centroids,
                          int32_t* nsplit_out, void* stream);

/* Statistics of one Lloyd iteration, left on the device (faiss ClusteringIterationStats): stats[0] = objective =
 * the doubles obj_parts[p * obj_part_stride], p < n_parts, added in ascending p (one per data-parallel rank: its
 * at_sum_f32 of the distances); stats[1] = imbalance factor k * sum(h^2) / (sum h)^2 of hassign [k]. */
int at_lloyd_stats_f64(at_ctx* ctx, const float* hassign, int k, const double* obj_parts,
                       int64_t obj_part_stride, int n_parts, double* stats, void* stream);

/* at_lloyd_stats_f64 followed by at_split_clusters_f32 in ONE launch (both are single-workgroup kernels over the k
 * counts; a Lloyd iteration of a sharded run is short enough for the launch between them to matter).  stats may be
 * null: then exactly at_split_clusters_f32. */
int at_lloyd_stats_split_f32(at_ctx* ctx, int d, int k, int64_t n, float* hassign, float* centroids, int32_t* nsplit_out,
                             const double* obj_parts, int64_t obj_part_stride, int n_parts, double* stats, void* stream);

/* *out (DEVICE double) = sum of v[0..n) accumulated in double with a fixed reduction tree. */
int at_sum_f32(at_ctx* ctx, const float* v, int64_t n, double* out, void* stream);

/* 1 if any of v[0..n) is NaN/Inf else 0, written to *flag (DEVICE int32). */
int at_any_nonfinite_f32(at_ctx* ctx, const float* v, int64_t n, int32_t* flag, void* stream);

/* The same verdict for the unit rows at_logmel_f32(..., fuse_l2norm = 1) has written since the last call, without
 * reading them again: the unit-row pass sets a flag in the context when a row's squared norm is not finite (a NaN or
 * Inf in the row; also squares that overflow, so a caller confirms a raised flag with at_any_nonfinite_f32).  Writes
 * 0 / 1 to *flag (DEVICE int32) and clears the context's flag, in stream order.  Replaces the scan faiss makes of
 * its training input (Clustering::train: FAISS_THROW_IF_NOT_MSG(std::isfinite(x_in[i]), ...), reached from
 * processors/cluster_creator.py:54-56) on the path where this library produced that input itself. */
int at_logmel_nonfinite_take(at_ctx* ctx, int32_t* flag, void* stream);

/* counts[t] = number of i with ids[i] == t, t in [0, k) (ids outside are skipped): the token statistics of
 * processors/spec_tokenizer.py:129-147 (a Python Counter over tokens.tolist()) without leaving the device. */
int at_token_histogram_i64(at_ctx* ctx, const int64_t* ids, int64_t n, int k, int64_t* counts, void* stream);

/* Rank-frequency statistics of that histogram, still on the device (processors/spec_tokenizer.py:146-240:
 * sorted(Counter.items()), np.cumsum / np.searchsorted, scipy.stats.linregress on the log-log curve).
 *   sorted_counts[r], sorted_tokens[r] (DEVICE int64 / int32 [k]): the r-th most frequent token and its count
 *     (ties in ascending token id; tokens that never occur come last with count 0);
 *   stats (DEVICE double [8]): [0] total occurrences; [1] U = tokens that occur; [2] the number of ranks whose
 *     cumulative share of the occurrences is below 0.8 (np.searchsorted(cumsum / total, 0.8)); [3] slope,
 *     [4] intercept, [5] r of the least-squares line through (ln rank, ln count) over the ranks
 *     [int(0.1 U), int(0.9 U)) (linregress' formulas, double); [6] the number of points of that fit. */
int at_token_stats_f64(at_ctx* ctx, const int64_t* counts, int k, int64_t* sorted_counts,
                       int32_t* sorted_tokens, double* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIO_TOKENS_AMD_H */
