/* oracle.h -- CPU restatement of the audio-tokens hot path (TEST INFRASTRUCTURE, see oracle.c). */
#ifndef AUDIO_TOKENS_ORACLE_H
#define AUDIO_TOKENS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

void orc_mt19937_raw(uint32_t seed, int64_t n, uint32_t* out);
void orc_rand_perm(int32_t* perm, int64_t n, int64_t seed);

void orc_l2norm_rows(const float* x, int64_t n, int d, float* y);

int orc_mel_filterbank(int sample_rate, int n_fft, int n_mels, float* fb);
int64_t orc_num_frames(int64_t L, int hop);
int orc_logmel(const float* wave, int64_t L, int sample_rate, int n_fft, int hop, int n_mels,
               const float* fb_or_null, float* out);

int64_t orc_resample_length(int64_t L, int orig_freq, int new_freq);
int orc_resample(const float* wave, int64_t L, int orig_freq, int new_freq, float* out);

void orc_assign_ref(const float* x, int64_t n, int d, const float* c, int k, int64_t* ids,
                    float* dis);
void orc_assign(const float* x, int64_t n, int d, const float* c, int k, int64_t* ids,
                float* dis);

int orc_split_clusters(int d, int k, int64_t n, float* hassign, float* centroids);
int orc_kmeans_train(const float* x, int64_t n, int d, int k, int niter, const float* init,
                     const int32_t* shard, int n_shards, float* centroids_out, double* stats,
                     int32_t* sub_perm_out, int64_t* assign_out);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
