// host.cpp -- context, error reporting and the host-side (sequential, RNG-driven) pieces of the
// FAISS clustering recipe that have no business on the GPU.
//
// Reference call sites replaced (paths in danavery/audio-tokens):
//   processors/cluster_creator.py:42-56   faiss.Kmeans.train -> Clustering::train_encoded:
//       subsample_training_set / centroid init  -> at_rand_perm_mt19937
//       split_clusters                          -> at_split_clusters_host
//   processors/spectrogram_generator.py:28-33  MelSpectrogram's mel filterbank -> at_mel_filterbank_host
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "at_internal.h"

static thread_local std::string g_last_error;

at_diag_counters g_at_diag = {};
int g_at_strict_errors = 0;

hipError_t at_hip_tolerated(hipError_t e, const char* file, int line) {
    if (e != hipSuccess && e != hipErrorNotReady) {   // (hipErrorNotReady is an answer, not a failure, and is not kept)
        (void)hipGetLastError();                      // the sticky copy of this failure: consumed here, where it is known
        g_at_diag.tolerated++;
        g_at_diag.tolerated_last_code = (int)e;
        g_at_diag.tolerated_last_file = file;
        g_at_diag.tolerated_last_line = line;
    }
    return e;
}

hipError_t at_stale_check(const char* file, int line) {
    const hipError_t s = hipPeekAtLastError();
    if (s == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    g_at_diag.stale_seen++;
    g_at_diag.stale_last_code = (int)s;
    g_at_diag.stale_last_file = file;
    g_at_diag.stale_last_line = line;
    return g_at_strict_errors ? s : hipSuccess;
}

int at_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

namespace {
struct DebugField { const char* name; const char* env; int at_debug::*field; int def; };
const DebugField kDebugFields[] = {
    {"assign_variant", "AT_ASSIGN_VARIANT", &at_debug::assign_variant, 0},
    {"filter_fused", "AT_FILTER_FUSED", &at_debug::filter_fused, 1},
    {"filter_sync", "AT_FILTER_SYNC", &at_debug::filter_sync, 0},
    {"prune_kernel", "AT_PRUNE_KERNEL", &at_debug::prune_kernel, 1},
    {"prune_nb", "AT_PRUNE_NB", &at_debug::prune_nb, 0},
    {"filter_screen", "AT_FILTER_SCREEN", &at_debug::filter_screen, 1},
    {"filter_nb", "AT_FILTER_NB", &at_debug::filter_nb, 0},
    {"filter_wps2", "AT_FILTER_WPS2", &at_debug::filter_wps2, 0},
    {"dmin_kernel", "AT_DMIN_KERNEL", &at_debug::dmin_kernel, 1},
    {"resample_simple", "AT_RESAMPLE_SIMPLE", &at_debug::resample_simple, 0},
    {"accum_buckets", "AT_ACCUM_BUCKETS", &at_debug::accum_buckets, 1},
    {"filter_stats", "AT_FILTER_STATS", &at_debug::filter_stats, 0},
    {"visit_bits", "AT_VISIT_BITS", &at_debug::visit_bits, 8},
    {"filter_timing", "AT_FILTER_TIMING", &at_debug::filter_timing, 0},
};
}  // namespace

extern "C" {

int at_version(void) { return AT_VERSION; }

int at_debug_set(at_ctx* ctx, const char* name, int value) {
    AT_REQUIRE(ctx && name, "at_debug_set: null argument");
    if (std::strcmp(name, "strict_errors") == 0) {   // process-wide (the launch macro has no context at hand)
        g_at_strict_errors = value != 0;
        return AT_OK;
    }
    for (const DebugField& f : kDebugFields)
        if (std::strcmp(f.name, name) == 0) {
            ctx->dbg.*(f.field) = value;
            return AT_OK;
        }
    return at_fail(AT_E_INVALID, "at_debug_set: no switch named '%s'", name);
}

int at_debug_get(const at_ctx* ctx, const char* name, int* value) {
    AT_REQUIRE(ctx && name && value, "at_debug_get: null argument");
    if (std::strcmp(name, "strict_errors") == 0) {
        *value = g_at_strict_errors;
        return AT_OK;
    }
    for (const DebugField& f : kDebugFields)
        if (std::strcmp(f.name, name) == 0) {
            *value = ctx->dbg.*(f.field);
            return AT_OK;
        }
    return at_fail(AT_E_INVALID, "at_debug_get: no switch named '%s'", name);
}

const char* at_last_error(void) { return g_last_error.c_str(); }

int at_diag_errors(int64_t* stale_seen, int* stale_last_code, int64_t* tolerated, int* tolerated_last_code, char* where,
                   int where_bytes, int reset) {
    if (stale_seen) *stale_seen = g_at_diag.stale_seen;
    if (stale_last_code) *stale_last_code = g_at_diag.stale_last_code;
    if (tolerated) *tolerated = g_at_diag.tolerated;
    if (tolerated_last_code) *tolerated_last_code = g_at_diag.tolerated_last_code;
    if (where && where_bytes > 0)
        snprintf(where, (size_t)where_bytes, "stale: %s:%d; tolerated: %s:%d",
                 g_at_diag.stale_last_file ? g_at_diag.stale_last_file : "-", g_at_diag.stale_last_line,
                 g_at_diag.tolerated_last_file ? g_at_diag.tolerated_last_file : "-", g_at_diag.tolerated_last_line);
    if (reset) g_at_diag = at_diag_counters{};
    return AT_OK;
}

// Test hook: leaves a failed HIP call's error pending in the calling thread, the way a call that swallows its
// return code does (hipEventElapsedTime on two events that were never recorded: hipErrorInvalidResourceHandle).
int at_debug_leave_error_pending(void) {
    hipEvent_t a = nullptr, b = nullptr;
    AT_HIP(hipEventCreate(&a));
    AT_HIP(hipEventCreate(&b));
    float ms = 0.0f;
    const hipError_t e = hipEventElapsedTime(&ms, a, b);   // deliberately not consumed
    const hipError_t d0 = hipEventDestroy(a), d1 = hipEventDestroy(b);   // (successful calls do not clear it)
    (void)d0; (void)d1;
    return e == hipSuccess ? at_fail(AT_E_INVALID, "at_debug_leave_error_pending: the call did not fail") : AT_OK;
}

int at_create(int device, at_ctx** out) {
    AT_REQUIRE(out != nullptr, "at_create: out is null");
    int ndev = 0;
    AT_HIP(hipGetDeviceCount(&ndev));
    AT_REQUIRE(device >= 0 && device < ndev, "at_create: device %d out of range (%d visible)",
               device, ndev);
    hipDeviceProp_t prop;
    AT_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return at_fail(AT_E_INVALID, "at_create: device %d is %s; this library is built for gfx950 only",
                       device, prop.gcnArchName);
    at_ctx* c = new (std::nothrow) at_ctx();
    if (!c) return at_fail(AT_E_NOMEM, "at_create: out of host memory");
    std::memset(c, 0, sizeof *c);
    c->device = device;
    c->filter_slot = AT_FILTER_RING;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (const DebugField& f : kDebugFields) {   // the only place the environment is read
        const char* e = std::getenv(f.env);
        c->dbg.*(f.field) = e ? std::atoi(e) : f.def;
    }
    if (const char* e = std::getenv("AT_STRICT_ERRORS")) g_at_strict_errors = std::atoi(e) != 0;
    *out = c;
    return AT_OK;
}

void at_destroy(at_ctx* ctx) {
    if (!ctx) return;
    // (a destructor has nobody to report to: failures are tolerated, and consumed so that they do not stay pending)
    int prev = 0;
    (void)AT_HIP_TOLERATE(hipGetDevice(&prev));
    (void)AT_HIP_TOLERATE(hipSetDevice(ctx->device));
    (void)AT_HIP_TOLERATE(hipDeviceSynchronize());
    for (int i = 0; i < WS_NSLOTS; i++)
        if (ctx->ws[i]) (void)AT_HIP_TOLERATE(hipFree(ctx->ws[i]));
    for (int i = 0; i < 2; i++)
        if (ctx->side_ev[i]) (void)AT_HIP_TOLERATE(hipEventDestroy(ctx->side_ev[i]));
    for (int s = 0; s <= AT_FILTER_RING; s++) {
        at_filter_slot& fs = ctx->fring[s];
        if (fs.copied) (void)AT_HIP_TOLERATE(hipEventDestroy(fs.copied));
        for (int i = 0; i < 2; i++)
            if (fs.ev[i]) (void)AT_HIP_TOLERATE(hipEventDestroy(fs.ev[i]));
    }
    if (ctx->side_ev2) (void)AT_HIP_TOLERATE(hipEventDestroy(ctx->side_ev2));
    if (ctx->side_stream) (void)AT_HIP_TOLERATE(hipStreamDestroy(ctx->side_stream));
    if (ctx->background_stream) (void)AT_HIP_TOLERATE(hipStreamDestroy(ctx->background_stream));
    if (ctx->mt_ready) (void)AT_HIP_TOLERATE(hipEventDestroy(ctx->mt_ready));
    if (ctx->sum_ev) (void)AT_HIP_TOLERATE(hipEventDestroy(ctx->sum_ev));
    if (ctx->filter_host_misc) (void)AT_HIP_TOLERATE(hipHostFree(ctx->filter_host_misc));
    std::free(ctx->fb_user_copy);
    std::free(ctx->any_user_copy);
    (void)AT_HIP_TOLERATE(hipSetDevice(prev));
    delete ctx;
}

int at_background_stream(at_ctx* ctx, void** stream_out) {
    AT_REQUIRE(ctx && stream_out, "at_background_stream: null pointer");
    if (!ctx->background_stream) {
        AT_HIP(hipSetDevice(ctx->device));
        int least = 0, greatest = 0;   // numerically: least >= greatest (lower numbers are served first)
        AT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        AT_HIP(hipStreamCreateWithPriority(&ctx->background_stream, hipStreamNonBlocking, least));
    }
    *stream_out = ctx->background_stream;
    return AT_OK;
}

int64_t at_workspace_bytes(const at_ctx* ctx) {
    if (!ctx) return 0;
    size_t t = 0;
    for (int i = 0; i < WS_NSLOTS; i++) t += ctx->ws_bytes[i];
    return (int64_t)t;
}

}  // extern "C"

int at_raise_lds(at_ctx* ctx, const void* func, size_t bytes) {
    if (bytes <= 32 * 1024) return AT_OK;
    const int cap = (int)(sizeof ctx->lds_raised / sizeof ctx->lds_raised[0]);
    int slot = -1;
    for (int i = 0; i < ctx->n_lds_raised; i++)
        if (ctx->lds_raised[i].func == func) { slot = i; break; }
    if (slot >= 0 && ctx->lds_raised[slot].bytes >= bytes) return AT_OK;
    AT_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (slot < 0 && ctx->n_lds_raised < cap) slot = ctx->n_lds_raised++;
    if (slot >= 0) { ctx->lds_raised[slot].func = func; ctx->lds_raised[slot].bytes = bytes; }
    return AT_OK;   // (a full table only means the attribute is set again next time)
}

void* at_ws(at_ctx* ctx, int slot, size_t bytes, hipStream_t stream) {
    if (bytes == 0) bytes = 16;
    if (ctx->ws_bytes[slot] >= bytes) return ctx->ws[slot];
    // Work already queued may still read the old buffer -- on `stream`, or on the context's side stream / a
    // caller's second stream (the slots are per purpose, not per stream): drain the device before freeing.
    if (ctx->ws[slot]) {
        (void)stream;
        const hipError_t es = AT_HIP_TOLERATE(hipDeviceSynchronize());
        if (es != hipSuccess) {
            at_fail(AT_E_HIP, "workspace slot %d: hipDeviceSynchronize failed before regrowth: %s", slot, hipGetErrorString(es));
            return nullptr;
        }
        (void)AT_HIP_TOLERATE(hipFree(ctx->ws[slot]));
        ctx->ws[slot] = nullptr;
        ctx->ws_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;  // a little slack so slowly growing batches do not thrash
    want = (want + 255) & ~size_t(255);
    void* p = nullptr;
    if (AT_HIP_TOLERATE(hipMalloc(&p, want)) != hipSuccess) {
        if (AT_HIP_TOLERATE(hipMalloc(&p, bytes)) != hipSuccess) {
            at_fail(AT_E_NOMEM, "workspace slot %d: hipMalloc(%zu) failed", slot, bytes);
            return nullptr;
        }
        want = bytes;
    }
    ctx->ws[slot] = p;
    ctx->ws_bytes[slot] = want;
    if (slot == WS_LOGMEL_FB) { ctx->fb_sr = ctx->fb_nfft = ctx->fb_nmels = 0; ctx->fb_user = nullptr; }
    if (slot == WS_RESAMPLE_TAPS) ctx->rs_orig = ctx->rs_new = 0;
    if (slot == WS_MT_RAW) ctx->mt_have = 0;
    if (slot == WS_LONG_PRED) ctx->long_pred_k = 0;
    if (slot == WS_BUCKETS) ctx->buckets_k = 0;
    if (slot == WS_LOGMEL_ANY) ctx->any_nfft = 0;
    return p;
}

// ---------------------------------------------------------------------------------------------
// faiss utils/random.cpp: RandomGenerator wraps std::mt19937; rand_int(max) = mt() % max;
// rand_float() = mt() / float(mt.max()).
namespace {
struct FaissRng {
    std::mt19937 mt;
    explicit FaissRng(int64_t seed) : mt((unsigned int)seed) {}
    int rand_int(int max) { return (int)(mt() % (unsigned long)max); }
    float rand_float() { return mt() / float(mt.max()); }
};
}  // namespace

extern "C" {

int at_rand_perm_mt19937(int64_t n, int64_t seed, int32_t* perm) {
    AT_REQUIRE(n >= 0 && n <= INT32_MAX, "at_rand_perm_mt19937: n=%lld out of range", (long long)n);
    AT_REQUIRE(perm != nullptr || n == 0, "at_rand_perm_mt19937: perm is null");
    for (int64_t i = 0; i < n; i++) perm[i] = (int32_t)i;
    FaissRng rng(seed);
    for (int64_t i = 0; i + 1 < n; i++) {
        int64_t other = i + rng.rand_int((int)(n - i));
        std::swap(perm[i], perm[other]);
    }
    return AT_OK;
}

int at_rand_perm_prefix_mt19937(int64_t n, int64_t seed, int64_t m, int32_t* prefix) {
    AT_REQUIRE(n >= 0 && n <= INT32_MAX && m >= 0 && m <= n, "at_rand_perm_prefix_mt19937: bad sizes n=%lld m=%lld",
               (long long)n, (long long)m);
    AT_REQUIRE(prefix != nullptr || m == 0, "at_rand_perm_prefix_mt19937: prefix is null");
    // Step i of the shuffle swaps position i with a position >= i, so positions < m are final after
    // step m-1 and only the positions those m steps touch ever differ from the identity.  Positions
    // < m live in `prefix` itself; touched positions >= m in an open-addressing table of <= m entries
    // (no O(n) pool: at n = 17.2 M, m = 2.1 M that pool was 69 MB of cache misses, 240 ms).  The draws
    // are sequential by definition but the ADDRESSES they name are known as soon as they are drawn:
    // a block of draws is generated first and its slots prefetched before the swaps run.
    const int64_t steps = std::min<int64_t>(m, n - 1);
    for (int64_t i = 0; i < m; i++) prefix[i] = (int32_t)i;
    if (steps <= 0) return AT_OK;
    int bits = 4;
    while ((int64_t(1) << bits) < 2 * m) bits++;
    struct Slot { uint32_t key, val; };   // key 0 = empty (keys are positions >= m >= 1)
    const size_t tsize = size_t(1) << bits;
    Slot* tab = (Slot*)std::calloc(tsize, sizeof(Slot));
    if (!tab) return at_fail(AT_E_NOMEM, "at_rand_perm_prefix_mt19937: out of host memory");
    const uint32_t mask = (uint32_t)(tsize - 1), shift = 32u - (uint32_t)bits;
    auto home = [&](uint32_t pos) { return (pos * 2654435761u) >> shift; };
    FaissRng rng(seed);
    constexpr int B = 64;
    uint32_t other[B];
    const uint32_t um = (uint32_t)m;
    for (int64_t i0 = 0; i0 < steps; i0 += B) {
        const int nb = (int)std::min<int64_t>(B, steps - i0);
        for (int j = 0; j < nb; j++) {
            const int64_t i = i0 + j;
            other[j] = (uint32_t)(i + rng.rand_int((int)(n - i)));
            if (other[j] < um) __builtin_prefetch(&prefix[other[j]], 1);
            else __builtin_prefetch(&tab[home(other[j])], 1);
        }
        for (int j = 0; j < nb; j++) {
            const int64_t i = i0 + j;
            const uint32_t o = other[j];
            if (o < um) {
                std::swap(prefix[i], prefix[o]);
            } else {
                uint32_t h = home(o);
                while (tab[h].key != 0 && tab[h].key != o) h = (h + 1) & mask;
                const uint32_t v = tab[h].key ? tab[h].val : o;
                tab[h].key = o;
                tab[h].val = (uint32_t)prefix[i];
                prefix[i] = (int32_t)v;
            }
        }
    }
    std::free(tab);
    return AT_OK;
}

int64_t at_num_frames(int64_t L, int hop) { return hop > 0 ? 1 + L / hop : 0; }

int at_mel_filterbank_host(int sample_rate, int n_fft, int n_mels, float* fb) {
    AT_REQUIRE(fb && sample_rate > 0 && n_fft >= 2 && n_mels >= 1, "at_mel_filterbank_host: bad arguments");
    const int n_freqs = n_fft / 2 + 1;
    const double f_max = (double)(sample_rate / 2);  // torchaudio: float(sample_rate // 2)
    auto to_mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto to_hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> edge(n_mels + 2);
    const double m_lo = to_mel(0.0), m_hi = to_mel(f_max);
    for (int i = 0; i < n_mels + 2; i++) edge[i] = to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int f = 0; f < n_freqs; f++) {
        const double hz = f_max * f / (n_freqs - 1);
        for (int m = 0; m < n_mels; m++) {
            const double rise = (hz - edge[m]) / (edge[m + 1] - edge[m]);
            const double fall = (edge[m + 2] - hz) / (edge[m + 2] - edge[m + 1]);
            const double tri = std::fmin(rise, fall);
            fb[(size_t)f * n_mels + m] = tri > 0.0 ? (float)tri : 0.0f;
        }
    }
    return AT_OK;
}

int at_split_clusters_host(int d, int k, int64_t n, float* hassign, float* centroids, int* nsplit) {
    AT_REQUIRE(hassign && centroids && d > 0 && k > 0, "at_split_clusters_host: bad arguments");
    constexpr double kEps = 1.0 / 1024.0;
    FaissRng rng(1234);
    int count = 0;
    const float denom = (float)(n - k);
    // acceptance probability (size-1)/(n-k) of every cluster, kept up to date as sizes change;
    // one draw is consumed per candidate visited, exactly as in FAISS, so this loop is the
    // sequential heart of the repair and is kept as tight as possible
    auto prob = [&](int c) { return (float)((hassign[c] - 1.0) / denom); };
    std::vector<float> p((size_t)k);
    bool have_p = false;
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] != 0.0f) continue;
        if (!have_p) {
            for (int c = 0; c < k; c++) p[c] = prob(c);
            have_p = true;
        }
        // walk the clusters cyclically from 0 until one accepts
        int donor = 0;
        while (!(rng.rand_float() < p[donor])) {
            if (++donor == k) donor = 0;
        }
        float* dst = centroids + (size_t)ci * d;
        float* src = centroids + (size_t)donor * d;
        std::memcpy(dst, src, sizeof(float) * (size_t)d);
        for (int j = 0; j < d; j++) {
            const double up = 1.0 + kEps, down = 1.0 - kEps;
            if ((j & 1) == 0) {
                dst[j] = (float)(dst[j] * up);
                src[j] = (float)(src[j] * down);
            } else {
                dst[j] = (float)(dst[j] * down);
                src[j] = (float)(src[j] * up);
            }
        }
        hassign[ci] = hassign[donor] / 2;
        hassign[donor] -= hassign[ci];
        p[ci] = prob(ci);
        p[donor] = prob(donor);
        count++;
    }
    if (nsplit) *nsplit = count;
    return AT_OK;
}


}  // extern "C"
