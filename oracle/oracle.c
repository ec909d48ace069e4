/*
 * oracle.c -- CPU restatement of the audio-tokens hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP path in audio_tokens_amd/csrc.  It is imported,
 * linked or executed only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under audio_tokens_amd/ may depend on it.
 *
 * PARITY UNPINNED.  The reference (danavery/audio-tokens @ 2024-10-16) keeps none of this
 * arithmetic in its own tree: it calls torchaudio==2.4.1 (environment.yml:265) and
 * faiss-gpu==1.8.0 / libfaiss==1.8.0 (environment.yml:69,137), neither of which exists in the
 * build container, and the reference has no tests, fixtures or golden vectors.  What follows
 * restates the published algorithms of those two packages at the reference's call sites:
 *
 *   log-mel      processors/spectrogram_generator.py:28-34,123-126
 *                (torchaudio MelSpectrogram(sample_rate,n_mels,n_fft,hop_length) defaults:
 *                 center=True, pad_mode="reflect", periodic Hann, power=2, htk mel, norm=None,
 *                 f_min=0, f_max=sr//2;  AmplitudeToDB(): 10*log10(clamp(x,1e-10)), top_db=None)
 *   row L2 norm  processors/cluster_creator.py:64-66, processors/spec_tokenizer.py:106-109
 *                (numpy: x / (sqrt(add.reduce(x*x, axis=1)) + 1e-10), fp32, pairwise summation)
 *   k-means      processors/cluster_creator.py:42-56  (faiss.Kmeans(d,k,niter=20).train(x[,init]):
 *                faiss Clustering::train_encoded, subsample_training_set, compute_centroids,
 *                split_clusters, utils/random.cpp rand_perm / RandomGenerator)
 *   search       processors/spec_tokenizer.py:77, 123-127 (faiss.IndexFlatL2.search(x,1):
 *                utils/distances.cpp exhaustive_L2sqr_blas / exhaustive_L2sqr_seq)
 *
 * Two free choices that FAISS leaves to MKL/AVX2 and that cannot be pinned without the binaries
 * are fixed here so that a GPU implementation can agree bit for bit:
 *   - an inner product <x,c> is a sequential fmaf chain over the feature axis, ascending index,
 *     starting from +0.0f (this is exactly what gfx950's v_mfma_f32_32x32x2_f32 computes);
 *   - a squared norm is the same chain with c = x.
 * Independent pins that do exist in the container are applied by tests/test_oracle_*.py:
 * std::mt19937 (libstdc++), numpy (row norm, bitwise), torch.stft, transformers' mel_filter_bank,
 * sklearn KMeans (Lloyd, no empty clusters).
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* mt19937 (32-bit), as std::mt19937 -- faiss utils/random.cpp RandomGenerator wraps this.     */

typedef struct {
    uint32_t s[624];
    int pos;
} orc_mt;

static void mt_seed(orc_mt* m, uint32_t seed) {
    m->s[0] = seed;
    for (int i = 1; i < 624; i++)
        m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    m->pos = 624;
}

static uint32_t mt_next(orc_mt* m) {
    if (m->pos >= 624) {
        uint32_t* s = m->s;
        for (int i = 0; i < 624; i++) {
            uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = s[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            s[i] = v;
        }
        m->pos = 0;
    }
    uint32_t y = m->s[m->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* RandomGenerator::rand_float(): mt() / float(mt.max())  -- mt() is a 64-bit unsigned long
 * holding a 32-bit value, converted to float (round to nearest), divided by 2^32 (float). */
static float mt_rand_float(orc_mt* m) {
    return (float)(uint64_t)mt_next(m) / (float)4294967295.0;
}

void orc_mt19937_raw(uint32_t seed, int64_t n, uint32_t* out) {
    orc_mt m;
    mt_seed(&m, seed);
    for (int64_t i = 0; i < n; i++) out[i] = mt_next(&m);
}

/* faiss utils/random.cpp rand_perm(int* perm, size_t n, int64_t seed):
 *   identity; for i in [0, n-1): i2 = i + rng.rand_int(n - i); swap(perm[i], perm[i2])
 *   with rand_int(max) = mt() % max and RandomGenerator(seed) = mt19937((unsigned)seed). */
void orc_rand_perm(int32_t* perm, int64_t n, int64_t seed) {
    orc_mt m;
    mt_seed(&m, (uint32_t)seed);
    for (int64_t i = 0; i < n; i++) perm[i] = (int32_t)i;
    for (int64_t i = 0; i + 1 < n; i++) {
        int32_t mx = (int32_t)(n - i);
        int64_t i2 = i + (int64_t)((uint64_t)mt_next(&m) % (uint64_t)mx);
        int32_t t = perm[i];
        perm[i] = perm[i2];
        perm[i2] = t;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Row L2 normalisation, numpy semantics (cluster_creator.py:64-66).                           */

/* numpy's pairwise float sum (loops_utils.h.src, @TYPE@_pairwise_sum; unchanged 1.x -> 2.x),
 * here applied to the squares a[i]*a[i] which numpy materialises as a float32 array first. */
static float np_pairwise_sumsq(const float* a, int64_t n) {
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; i++) res += a[i] * a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j] * a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j] * a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i] * a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sumsq(a, n2) + np_pairwise_sumsq(a + n2, n - n2);
    }
}

void orc_l2norm_rows(const float* x, int64_t n, int d, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const float* xi = x + i * (int64_t)d;
        float nrm = sqrtf(np_pairwise_sumsq(xi, d));
        float den = nrm + 1e-10f; /* float32 array + python float -> float32 */
        float* yi = y + i * (int64_t)d;
        for (int j = 0; j < d; j++) yi[j] = xi[j] / den;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Log-mel (spectrogram_generator.py:28-34,123-126).  Evaluated in double: the oracle is the    */
/* exact value that both torchaudio's fp32 pipeline and the HIP kernel approximate.            */

static double hz_to_mel_htk(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel_to_hz_htk(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

/* torchaudio.functional.melscale_fbanks(n_freqs, f_min=0, f_max=sr//2, n_mels, sr, norm=None,
 * mel_scale="htk") -> fb[n_freqs][n_mels]; triangles max(0, min(down, up)). */
int orc_mel_filterbank(int sample_rate, int n_fft, int n_mels, float* fb) {
    int n_freqs = n_fft / 2 + 1;
    double f_max = (double)(sample_rate / 2);
    double m_min = hz_to_mel_htk(0.0), m_max = hz_to_mel_htk(f_max);
    double* f_pts = (double*)malloc(sizeof(double) * (size_t)(n_mels + 2));
    if (!f_pts) return -1;
    for (int i = 0; i < n_mels + 2; i++) {
        double m = m_min + (m_max - m_min) * (double)i / (double)(n_mels + 1);
        f_pts[i] = mel_to_hz_htk(m);
    }
    for (int f = 0; f < n_freqs; f++) {
        double freq = f_max * (double)f / (double)(n_freqs - 1);
        for (int m = 0; m < n_mels; m++) {
            double down = (freq - f_pts[m]) / (f_pts[m + 1] - f_pts[m]);
            double up = (f_pts[m + 2] - freq) / (f_pts[m + 2] - f_pts[m + 1]);
            double v = down < up ? down : up;
            fb[(size_t)f * n_mels + m] = (float)(v > 0.0 ? v : 0.0);
        }
    }
    free(f_pts);
    return 0;
}

int64_t orc_num_frames(int64_t L, int hop) { return 1 + L / hop; }

/* in-place radix-2 FFT; tw[2j], tw[2j+1] = cos, -sin of 2*pi*j/n for j < n/2 */
static void fft_radix2(double* re, double* im, int n, const double* tw) {
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < len / 2; k++) {
                double wr = tw[2 * k * step], wi = tw[2 * k * step + 1];
                int a = i + k, b = i + k + len / 2;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

/* One clip: wave[L] -> out[n_mels][T] (mel-major, the reference's file layout).
 * fb may be NULL (built with orc_mel_filterbank). n_fft must be a power of two. */
int orc_logmel(const float* wave, int64_t L, int sample_rate, int n_fft, int hop, int n_mels,
               const float* fb_in, float* out) {
    if (n_fft < 2 || (n_fft & (n_fft - 1)) || hop < 1 || L <= n_fft / 2) return -1;
    int n_freqs = n_fft / 2 + 1, pad = n_fft / 2;
    int64_t T = orc_num_frames(L, hop);
    float* fb_own = NULL;
    const float* fb = fb_in;
    if (!fb) {
        fb_own = (float*)malloc(sizeof(float) * (size_t)n_freqs * n_mels);
        if (!fb_own || orc_mel_filterbank(sample_rate, n_fft, n_mels, fb_own)) return -1;
        fb = fb_own;
    }
    float* win = (float*)malloc(sizeof(float) * (size_t)n_fft);
    for (int i = 0; i < n_fft; i++) /* torch.hann_window(n_fft, periodic=True), fp32 */
        win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)n_fft));
    double* tw = (double*)malloc(sizeof(double) * (size_t)n_fft);
    for (int j = 0; j < n_fft / 2; j++) {
        tw[2 * j] = cos(2.0 * M_PI * (double)j / (double)n_fft);
        tw[2 * j + 1] = -sin(2.0 * M_PI * (double)j / (double)n_fft);
    }
#pragma omp parallel
    {
        double* re = (double*)malloc(sizeof(double) * (size_t)n_fft);
        double* im = (double*)malloc(sizeof(double) * (size_t)n_fft);
        double* pw = (double*)malloc(sizeof(double) * (size_t)n_freqs);
#pragma omp for schedule(static)
        for (int64_t t = 0; t < T; t++) {
            for (int i = 0; i < n_fft; i++) {
                int64_t p = t * hop + i - pad; /* index into the un-padded signal */
                if (p < 0) p = -p;             /* reflect (no edge repeat) */
                if (p >= L) p = 2 * (L - 1) - p;
                re[i] = (double)wave[p] * (double)win[i];
                im[i] = 0.0;
            }
            fft_radix2(re, im, n_fft, tw);
            for (int f = 0; f < n_freqs; f++) pw[f] = re[f] * re[f] + im[f] * im[f];
            for (int m = 0; m < n_mels; m++) {
                double s = 0.0;
                for (int f = 0; f < n_freqs; f++) {
                    const float wgt = fb[(size_t)f * n_mels + m];
                    if (wgt != 0.0f) s += pw[f] * (double)wgt;
                }
                if (s < 1e-10) s = 1e-10;
                out[(size_t)m * T + t] = (float)(10.0 * log10(s));
            }
        }
        free(re); free(im); free(pw);
    }
    free(win);
    free(tw);
    free(fb_own);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Nearest centroid (faiss IndexFlatL2.search(x, 1)).                                          */

static inline float dot_chain(const float* a, const float* b, int d) {
    float acc = 0.0f;
    for (int j = 0; j < d; j++) acc = fmaf(a[j], b[j], acc);
    return acc;
}

/* Plain restatement, one (row, centroid) pair at a time.  n >= 20: exhaustive_L2sqr_blas
 * (dis = x_norm + y_norm - 2*ip, negative -> 0, strict '<' so the lowest index wins a tie);
 * n < 20 (distance_compute_blas_threshold): exhaustive_L2sqr_seq (direct sum (x-y)^2). */
void orc_assign_ref(const float* x, int64_t n, int d, const float* c, int k, int64_t* ids,
                    float* dis) {
    if (n < 20) {
        for (int64_t i = 0; i < n; i++) {
            float best = INFINITY;
            int64_t bi = -1;
            for (int j = 0; j < k; j++) {
                float acc = 0.0f;
                for (int t = 0; t < d; t++) {
                    float df = x[i * d + t] - c[(int64_t)j * d + t];
                    acc = fmaf(df, df, acc);
                }
                if (acc < best) { best = acc; bi = j; }
            }
            ids[i] = bi;
            if (dis) dis[i] = best;
        }
        return;
    }
    float* cn = (float*)malloc(sizeof(float) * (size_t)k);
    for (int j = 0; j < k; j++) cn[j] = dot_chain(c + (int64_t)j * d, c + (int64_t)j * d, d);
    for (int64_t i = 0; i < n; i++) {
        const float* xi = x + i * d;
        float xn = dot_chain(xi, xi, d);
        float best = INFINITY;
        int64_t bi = -1;
        for (int j = 0; j < k; j++) {
            float ip = dot_chain(xi, c + (int64_t)j * d, d);
            float v = (xn + cn[j]) - 2.0f * ip;
            if (v < 0.0f) v = 0.0f;
            if (v < best) { best = v; bi = j; }
        }
        ids[i] = bi;
        if (dis) dis[i] = best;
    }
    free(cn);
}

/* Same results, bit for bit, arranged for speed (centroids transposed so the fmaf chains of
 * many centroids advance together; rows spread over OpenMP threads).  This is the version the
 * cpu_baseline leg of bench.py times. */
void orc_assign(const float* x, int64_t n, int d, const float* c, int k, int64_t* ids,
                float* dis) {
    if (n < 20) { orc_assign_ref(x, n, d, c, k, ids, dis); return; }
    float* cn = (float*)malloc(sizeof(float) * (size_t)k);
    float* ct = (float*)malloc(sizeof(float) * (size_t)k * d);
    for (int j = 0; j < k; j++) {
        cn[j] = dot_chain(c + (int64_t)j * d, c + (int64_t)j * d, d);
        for (int t = 0; t < d; t++) ct[(size_t)t * k + j] = c[(int64_t)j * d + t];
    }
    enum { JB = 512 };
#pragma omp parallel
    {
        float acc[JB];
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            const float* xi = x + i * d;
            float xn = dot_chain(xi, xi, d);
            float best = INFINITY;
            int64_t bi = -1;
            for (int j0 = 0; j0 < k; j0 += JB) {
                int jn = k - j0 < JB ? k - j0 : JB;
                for (int j = 0; j < jn; j++) acc[j] = 0.0f;
                for (int t = 0; t < d; t++) {
                    const float xv = xi[t];
                    const float* row = ct + (size_t)t * k + j0;
                    for (int j = 0; j < jn; j++) acc[j] = fmaf(xv, row[j], acc[j]);
                }
                for (int j = 0; j < jn; j++) {
                    float v = (xn + cn[j0 + j]) - 2.0f * acc[j];
                    if (v < 0.0f) v = 0.0f;
                    if (v < best) { best = v; bi = j0 + j; }
                }
            }
            ids[i] = bi;
            if (dis) dis[i] = best;
        }
    }
    free(cn);
    free(ct);
}

/* ------------------------------------------------------------------------------------------ */
/* faiss Clustering (ClusteringParameters defaults except niter): nredo=1, spherical=false,    */
/* int_centroids=false, update_index=false, frozen_centroids=false, min_points_per_centroid=39,*/
/* max_points_per_centroid=256, seed=1234.                                                     */

/* split_clusters: for every empty cluster ci pick a donor cj by the cyclic acceptance scan,
 * copy it, perturb the pair symmetrically with EPS = 1/1024, halve the donor's count. */
static int split_clusters(int d, int k, int64_t n, float* hassign, float* cent) {
    const double EPS = 1.0 / 1024.0;
    int nsplit = 0;
    orc_mt rng;
    mt_seed(&rng, 1234u);
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] != 0.0f) continue;
        int cj = 0;
        for (;; cj = (cj + 1) % k) {
            float p = (float)(((double)hassign[cj] - 1.0) / (double)(float)(n - k));
            float r = mt_rand_float(&rng);
            if (r < p) break;
        }
        memcpy(cent + (size_t)ci * d, cent + (size_t)cj * d, sizeof(float) * (size_t)d);
        for (int j = 0; j < d; j++) {
            if (j % 2 == 0) {
                cent[(size_t)ci * d + j] = (float)((double)cent[(size_t)ci * d + j] * (1 + EPS));
                cent[(size_t)cj * d + j] = (float)((double)cent[(size_t)cj * d + j] * (1 - EPS));
            } else {
                cent[(size_t)ci * d + j] = (float)((double)cent[(size_t)ci * d + j] * (1 - EPS));
                cent[(size_t)cj * d + j] = (float)((double)cent[(size_t)cj * d + j] * (1 + EPS));
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
        nsplit++;
    }
    return nsplit;
}

int orc_split_clusters(int d, int k, int64_t n, float* hassign, float* centroids) {
    return split_clusters(d, k, n, hassign, centroids);
}

/* compute_centroids + the 1/count scaling.  FAISS: each centroid is owned by one thread that
 * walks i = 0..n-1, so a centroid's sum is the ascending-i fp32 sum of its members.
 * shard (optional, [n], values in [0, n_shards)): the data-parallel variant -- each shard forms
 * its own ascending-i partial sums and the partials are added in shard order (what the
 * multi-GPU path does; n_shards == 1 is FAISS exactly). */
static void compute_centroids(const float* x, int64_t n, int d, int k, const int64_t* assign,
                              const int32_t* shard, int n_shards, float* hassign, float* cent) {
    if (!shard) n_shards = 1;
    size_t kd = (size_t)k * d;
    float* part = (float*)calloc(kd * (size_t)n_shards, sizeof(float));
    float* hpart = (float*)calloc((size_t)k * (size_t)n_shards, sizeof(float));
    for (int64_t i = 0; i < n; i++) {
        int s = shard ? shard[i] : 0;
        int64_t ci = assign[i];
        float* c = part + (size_t)s * kd + (size_t)ci * d;
        const float* xi = x + i * d;
        hpart[(size_t)s * k + ci] += 1.0f;
        for (int j = 0; j < d; j++) c[j] += xi[j];
    }
    memset(cent, 0, sizeof(float) * kd);
    memset(hassign, 0, sizeof(float) * (size_t)k);
    for (int s = 0; s < n_shards; s++) {
        for (size_t t = 0; t < kd; t++) cent[t] += part[(size_t)s * kd + t];
        for (int c = 0; c < k; c++) hassign[c] += hpart[(size_t)s * k + c];
    }
    free(part);
    free(hpart);
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] == 0.0f) continue;
        float norm = 1.0f / hassign[ci];
        float* c = cent + (size_t)ci * d;
        for (int j = 0; j < d; j++) c[j] *= norm;
    }
}

static double imbalance_factor(int64_t n, int k, const int64_t* assign) {
    double* hist = (double*)calloc((size_t)k, sizeof(double));
    for (int64_t i = 0; i < n; i++) hist[assign[i]] += 1.0;
    double tot = 0, uf = 0;
    for (int i = 0; i < k; i++) { tot += hist[i]; uf += hist[i] * hist[i]; }
    free(hist);
    return uf * k / (tot * tot);
}

/* faiss.Kmeans(d, k, niter).train(x, init_centroids): one Clustering::train call.
 *   x [n][d]; init may be NULL; centroids_out [k][d];
 *   stats (optional) [niter][4] = {obj, imbalance_factor, nsplit, n_points_used};
 *   sub_perm_out (optional, [min(n, 256k)]): the rows of x that were actually clustered, in
 *   order; assign_out (optional, same length): assignment of those rows in the LAST iteration.
 *   shard (optional, [n]): owner of every row of x, see compute_centroids.
 * Returns 0; -2 if n < k; -3 if x holds a NaN/Inf (faiss throws in both cases). */
int orc_kmeans_train(const float* x, int64_t n, int d, int k, int niter, const float* init,
                     const int32_t* shard, int n_shards, float* centroids_out, double* stats,
                     int32_t* sub_perm_out, int64_t* assign_out) {
    const int max_ppc = 256, min_ppc = 39;
    const int64_t seed = 1234;
    if (n < k) return -2;
    for (int64_t i = 0; i < n * d; i++)
        if (!isfinite(x[i])) return -3;

    int64_t nx = n;
    const float* xs = x;
    float* x_new = NULL;
    int32_t* shard_new = NULL;
    const int32_t* sh = shard;
    if (nx > (int64_t)k * max_ppc) { /* subsample_training_set */
        int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)nx);
        orc_rand_perm(perm, nx, seed);
        nx = (int64_t)k * max_ppc;
        x_new = (float*)malloc(sizeof(float) * (size_t)nx * d);
        if (shard) shard_new = (int32_t*)malloc(sizeof(int32_t) * (size_t)nx);
        for (int64_t i = 0; i < nx; i++) {
            memcpy(x_new + i * d, x + (int64_t)perm[i] * d, sizeof(float) * (size_t)d);
            if (shard) shard_new[i] = shard[perm[i]];
            if (sub_perm_out) sub_perm_out[i] = perm[i];
        }
        free(perm);
        xs = x_new;
        sh = shard_new;
    } else {
        if (nx < (int64_t)k * min_ppc)
            fprintf(stderr,
                    "WARNING clustering %ld points to %d centroids: please provide at least %ld "
                    "training points\n", (long)nx, k, (long)k * min_ppc);
        if (sub_perm_out)
            for (int64_t i = 0; i < nx; i++) sub_perm_out[i] = (int32_t)i;
    }

    float* cent = centroids_out;
    if (nx == k) { /* corner case: the training set becomes the centroids */
        memcpy(cent, xs, sizeof(float) * (size_t)k * d);
        if (stats && niter > 0) { stats[0] = 0; stats[1] = 1.0; stats[2] = 0; stats[3] = (double)nx; }
        if (assign_out) for (int64_t i = 0; i < nx; i++) assign_out[i] = i;
        free(x_new); free(shard_new);
        return 0;
    }
    if (init) {
        memcpy(cent, init, sizeof(float) * (size_t)k * d);
    } else { /* random points of the (subsampled) set, seed + 1 */
        int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)nx);
        orc_rand_perm(perm, nx, seed + 1);
        for (int i = 0; i < k; i++)
            memcpy(cent + (size_t)i * d, xs + (int64_t)perm[i] * d, sizeof(float) * (size_t)d);
        free(perm);
    }

    int64_t* assign = (int64_t*)malloc(sizeof(int64_t) * (size_t)nx);
    float* dis = (float*)malloc(sizeof(float) * (size_t)nx);
    float* hassign = (float*)malloc(sizeof(float) * (size_t)k);
    for (int it = 0; it < niter; it++) {
        orc_assign(xs, nx, d, cent, k, assign, dis);
        float obj = 0;
        for (int64_t j = 0; j < nx; j++) obj += dis[j];
        double imb = imbalance_factor(nx, k, assign);
        compute_centroids(xs, nx, d, k, assign, sh, n_shards, hassign, cent);
        int nsplit = split_clusters(d, k, nx, hassign, cent);
        if (stats) {
            stats[4 * it + 0] = (double)obj;
            stats[4 * it + 1] = imb;
            stats[4 * it + 2] = (double)nsplit;
            stats[4 * it + 3] = (double)nx;
        }
    }
    if (assign_out) memcpy(assign_out, assign, sizeof(int64_t) * (size_t)nx);
    free(assign); free(dis); free(hassign);
    free(x_new); free(shard_new);
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * torchaudio.transforms.Resample(orig_freq, new_freq)(wave) -- the call at
 * processors/spectrogram_generator.py:117-121.  torchaudio 2.4.1 (requirements.txt pin, absent
 * from /root/reference): functional._get_sinc_resample_kernel + _apply_sinc_resample_kernel with
 * the defaults sinc_interp_hann, lowpass_filter_width = 6, rolloff = 0.99:
 *   orig, new    = freqs / gcd;  base = min(orig, new) * rolloff;  width = ceil(6 * orig / base)
 *   t[j][k]      = clamp((-j/new + (k - width)/orig) * base, -6, 6)          (float64)
 *   kernel[j][k] = (t == 0 ? 1 : sin(pi t)/(pi t)) * cos(pi t / 12)^2 * base/orig -> float32
 *   y[i*new + j] = sum_k kernel[j][k] * xpad[i*orig + k],  xpad = x with `width` zeros in front
 *   output length ceil(new * L / orig)
 * The dot product is accumulated in double here (the order of torch's fp32 conv1d is not
 * specified); the device kernel is compared within a stated tolerance.  out_taps may be NULL. */
static int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

int64_t orc_resample_length(int64_t L, int orig_freq, int new_freq) {
    int g = gcd_int(orig_freq, new_freq);
    int64_t orig = orig_freq / g, nw = new_freq / g;
    return (nw * L + orig - 1) / orig;
}

int orc_resample(const float* wave, int64_t L, int orig_freq, int new_freq, float* out) {
    if (!wave || !out || L <= 0 || orig_freq <= 0 || new_freq <= 0) return -1;
    const int g = gcd_int(orig_freq, new_freq);
    const int orig = orig_freq / g, nw = new_freq / g;
    const double lpw = 6.0, rolloff = 0.99;
    const double base = (orig < nw ? orig : nw) * rolloff;
    const int width = (int)ceil(lpw * orig / base);
    const int K = 2 * width + orig;
    float* taps = (float*)malloc(sizeof(float) * (size_t)nw * K);
    if (!taps) return -2;
    for (int j = 0; j < nw; j++)
        for (int k = 0; k < K; k++) {
            double t = ((double)(-j) / nw + (double)(k - width) / orig) * base;
            if (t < -lpw) t = -lpw;
            if (t > lpw) t = lpw;
            double w = cos(t * M_PI / lpw / 2);
            w *= w;
            t *= M_PI;
            double sinc = t == 0.0 ? 1.0 : sin(t) / t;
            taps[(size_t)j * K + k] = (float)(sinc * w * (base / orig));
        }
    const int64_t out_len = orc_resample_length(L, orig_freq, new_freq);
#pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < out_len; o++) {
        const int64_t i = o / nw;
        const int j = (int)(o - i * nw);
        const float* t = taps + (size_t)j * K;
        double acc = 0.0;
        for (int k = 0; k < K; k++) {
            const int64_t s = i * orig - width + k;
            if (s >= 0 && s < L) acc += (double)t[k] * (double)wave[s];
        }
        out[o] = (float)acc;
    }
    free(taps);
    return 0;
}
