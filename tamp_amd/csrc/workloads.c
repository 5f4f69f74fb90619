/*
 * workloads.c -- deterministic synthetic inputs for BASELINE.json's configs.
 *
 * Host-only data generators (no codec logic): bench.py, the tests and tests/golden/make_golden.py
 * all draw their inputs from here so that a fixture committed in this container names exactly the
 * bytes the GPU box regenerates.  Integer arithmetic only -> identical bytes on every host.
 *
 *   wl_synth_text : config 2 -- per-stream Zipf word sampler ("synthetic text")
 *   wl_telemetry  : config 5 -- 7-bit JSON-ish telemetry messages + their shared custom dictionary
 *   wl_lcg_runs   : run-heavy two-letter data (same recurrence as the reference's LCG fuzz corpus,
 *                   tests/test_compressor_decompressor.py:566-580 / ctests/test_compressor.c:704-723)
 *   wl_stress     : xorshift stress blocks (three shapes modelled on devices/common/tamp_bench.c:33-74)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t xs32(uint32_t *s) {
    uint32_t x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return *s = x;
}

/* ------------------------------------------------------------------ vocabulary */
enum { VOCAB = 512, WORD_MAX = 10 };
static uint8_t g_word[VOCAB][WORD_MAX];
static uint8_t g_wlen[VOCAB];
static uint32_t g_cdf[VOCAB]; /* cumulative integer Zipf(1.0) weights */
static uint32_t g_total;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void build_vocab(void) {
    uint32_t s = 0xC0FFEEu;
    for (int w = 0; w < VOCAB; w++) {
        g_wlen[w] = (uint8_t)(2 + xs32(&s) % 9);
        for (int k = 0; k < g_wlen[w]; k++) g_word[w][k] = (uint8_t)('a' + xs32(&s) % 26);
    }
    uint32_t acc = 0;
    for (int r = 0; r < VOCAB; r++) {
        acc += (1u << 20) / (uint32_t)(r + 1);
        g_cdf[r] = acc;
    }
    g_total = acc;
}

static inline int zipf_draw(uint32_t *s) {
    uint32_t u = xs32(s) % g_total;
    int lo = 0, hi = VOCAB - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (u < g_cdf[mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

static void synth_text_one(uint8_t *out, size_t len, uint64_t index) {
    uint32_t s = (uint32_t)(0x9E3779B9u * (uint32_t)(index + 1));
    if (s == 0) s = 1;
    size_t o = 0;
    int close_tag = 0;
    while (o < len) {
        int w = zipf_draw(&s);
        for (int k = 0; k < g_wlen[w] && o < len; k++) out[o++] = g_word[w][k];
        const char *sep;
        if (close_tag) {
            sep = "</b> ";
            close_tag = 0;
        } else {
            uint32_t r = xs32(&s) % 100;
            if (r < 85)
                sep = " ";
            else if (r < 90)
                sep = ", ";
            else if (r < 95)
                sep = ". ";
            else if (r < 98)
                sep = "\n";
            else {
                sep = " <b>";
                close_tag = 1;
            }
        }
        for (const char *c = sep; *c && o < len; c++) out[o++] = (uint8_t)*c;
    }
}

/* ------------------------------------------------------------------ telemetry (config 5) */
static void telemetry_one(uint8_t *out, size_t len, uint64_t index) {
    uint32_t s = (uint32_t)(0x10C0FFEEu + (uint32_t)index);
    if (s == 0) s = 1;
    char buf[320];
    uint32_t id = xs32(&s) % 100000u, ts = 1700000000u + xs32(&s) % 40000000u;
    int t_int = (int)(xs32(&s) % 70) - 20;
    uint32_t t_frac = xs32(&s) % 10, h = xs32(&s) % 101, v_int = 3 + xs32(&s) % 2, v_frac = xs32(&s) % 1000;
    int m = snprintf(buf, sizeof buf, "{\"id\":\"dev-%05u\",\"ts\":%u,\"t\":%d.%u,\"h\":%u,\"v\":%u.%03u,\"ok\":true}", id,
                     ts, t_int, t_frac, h, v_int, v_frac);
    size_t o = 0;
    for (; o < len && (int)o < m; o++) out[o] = (uint8_t)buf[o];
    for (; o < len; o++) out[o] = ' ';
}

/* Shared 256-byte custom dictionary for config 5: caller passes the seeded default
 * (initialize_dictionary(256, literal=7)); the tail is overwritten with the field skeleton, the
 * idiom of docs/source/custom_dictionary.rst:141-163. */
void wl_telemetry_dictionary(uint8_t *dict256) {
    static const char skel[] =
        "{\"id\":\"dev-0\",\"ts\":17,\"t\":2.5,\"h\":50,\"v\":3.300,\"ok\":true}          {\"id\":\"dev-1\",\"ts\":170,\"t\":-1.0,\"h\":4,\"v\":4.";
    size_t m = sizeof(skel) - 1;
    if (m > 200) m = 200;
    memcpy(dict256 + 256 - m, skel, m);
}

/* ------------------------------------------------------------------ LCG run-heavy */
static void lcg_runs_one(uint8_t *out, size_t len, uint64_t index) {
    uint32_t state = (uint32_t)index * 2u + 12345u;
    size_t o = 0;
    while (o < len) {
        state = (state * 1103515245u + 12345u) & 0x7FFFFFFFu;
        uint32_t r = state >> 7;
        uint8_t b = (uint8_t)('a' + (r & 1));
        if (r & 2) {
            uint32_t run = 1 + ((r >> 2) & 3);
            for (uint32_t j = 0; j < run && o < len; j++) out[o++] = b;
        } else if (r & 4) {
            out[o++] = b;
            if (o < len) out[o++] = (uint8_t)('a' + ((r >> 3) & 1));
        } else {
            out[o++] = b;
        }
    }
}

/* ------------------------------------------------------------------ xorshift stress shapes */
static void stress_one(uint8_t *out, size_t len, uint64_t index) {
    uint32_t s = 0x1234abcdu ^ (uint32_t)(index * 2654435761u);
    if (s == 0) s = 1;
    int shape = (int)(index % 3);
    size_t o = 0;
    if (shape == 0) { /* incompressible */
        while (o < len) out[o++] = (uint8_t)xs32(&s);
    } else if (shape == 1) { /* long runs + sparse noise: exercises RLE accumulation across many steps */
        while (o < len) {
            uint32_t r = xs32(&s);
            uint8_t b = (uint8_t)(r >> 24);
            uint32_t run = (r & 1) ? 1 + (r >> 1) % 700 : 1 + (r >> 1) % 5;
            for (uint32_t j = 0; j < run && o < len; j++) out[o++] = b;
        }
    } else { /* long repeats at varying distances: exercises extended matches and window-end truncation */
        while (o < len) {
            uint32_t r = xs32(&s);
            if (o > 40 && (r & 3)) {
                size_t dist = 1 + (r >> 2) % (o < 1500 ? o : 1500);
                size_t n = 3 + (r >> 16) % 200;
                for (size_t j = 0; j < n && o < len; j++, o++) out[o] = out[o - dist];
            } else {
                size_t n = 1 + (r >> 8) % 24;
                for (size_t j = 0; j < n && o < len; j++) out[o++] = (uint8_t)('A' + xs32(&s) % 20);
            }
        }
    }
}

/* ------------------------------------------------------------------ threaded fan-out */
typedef void (*gen_fn)(uint8_t *, size_t, uint64_t);
typedef struct {
    gen_fn fn;
    uint8_t *out;
    size_t stream_len;
    uint64_t first_index;
    size_t begin, end;
} Job;

static void *run_job(void *a) {
    Job *j = (Job *)a;
    for (size_t i = j->begin; i < j->end; i++) j->fn(j->out + i * j->stream_len, j->stream_len, j->first_index + i);
    return NULL;
}

static void fan_out(gen_fn fn, uint8_t *out, size_t n_streams, size_t stream_len, uint64_t first_index, int threads) {
    pthread_once(&g_once, build_vocab);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n_streams) threads = n_streams ? (int)n_streams : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    Job *jobs = (Job *)malloc(sizeof(Job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (Job){fn, out, stream_len, first_index, n_streams * (size_t)t / (size_t)threads,
                        n_streams * (size_t)(t + 1) / (size_t)threads};
        pthread_create(&th[t], NULL, run_job, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* Stream i (global index first_index+i) occupies out[i*stream_len .. (i+1)*stream_len). */
void wl_synth_text(uint8_t *out, size_t n_streams, size_t stream_len, uint64_t first_index, int threads) {
    fan_out(synth_text_one, out, n_streams, stream_len, first_index, threads);
}
void wl_telemetry(uint8_t *out, size_t n_streams, size_t stream_len, uint64_t first_index, int threads) {
    fan_out(telemetry_one, out, n_streams, stream_len, first_index, threads);
}
void wl_lcg_runs(uint8_t *out, size_t n_streams, size_t stream_len, uint64_t first_index, int threads) {
    fan_out(lcg_runs_one, out, n_streams, stream_len, first_index, threads);
}
void wl_stress(uint8_t *out, size_t n_streams, size_t stream_len, uint64_t first_index, int threads) {
    fan_out(stress_one, out, n_streams, stream_len, first_index, threads);
}
