/*
 * ref_harness.c -- thin driver around the REFERENCE tamp C library.  TEST / BENCH ONLY.
 *
 * Compiled by oracle/Makefile together with the reference's own sources, taken where
 * they lie under /root/reference/tamp/_c_src (nothing is copied into this repo), into
 * oracle/_ref/libtamp_ref.so.  The product never links this.
 *
 * It performs exactly the calls the reference's own benches time
 * (tools/c-profiler/main.c:52-54, devices/common/tamp_bench.c:118-121,159-162):
 *   tamp_compressor_init + tamp_compressor_compress_and_flush(write_token=false)
 *   tamp_decompressor_init(conf=NULL) + tamp_decompressor_decompress
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tamp/compressor.h"
#include "tamp/decompressor.h"

int ref_sizeof_compressor(void) { return (int)sizeof(TampCompressor); }
int ref_sizeof_decompressor(void) { return (int)sizeof(TampDecompressor); }
int ref_sizeof_conf(void) { return (int)sizeof(TampConf); }

void ref_initialize_dictionary(unsigned char *buf, size_t size, int literal) {
    tamp_initialize_dictionary(buf, size, (uint8_t)literal);
}

int ref_min_pattern_size(int window, int literal) {
    return tamp_compute_min_pattern_size((uint8_t)window, (uint8_t)literal);
}

int ref_compress(int window, int literal, int custom, int extended, int dictionary_reset, int lazy,
                 const unsigned char *dict, const unsigned char *in, size_t n, unsigned char *out, size_t cap,
                 size_t *out_len) {
    TampConf conf;
    memset(&conf, 0, sizeof conf);
    conf.window = (uint16_t)window;
    conf.literal = (uint16_t)literal;
    conf.use_custom_dictionary = custom != 0;
    conf.extended = extended != 0;
    conf.dictionary_reset = dictionary_reset != 0;
#if TAMP_LAZY_MATCHING
    conf.lazy_matching = lazy != 0;
#else
    (void)lazy;
#endif
    unsigned char window_buf[1 << 15];
    if (custom && dict) memcpy(window_buf, dict, (size_t)1 << window);
    TampCompressor c;
    size_t written = 0, consumed = 0;
    tamp_res res = tamp_compressor_init(&c, &conf, window_buf);
    if (res == TAMP_OK) res = tamp_compressor_compress_and_flush(&c, out, cap, &written, in, n, &consumed, false);
    if (out_len) *out_len = written;
    return res;
}

int ref_decompress(const unsigned char *in, size_t n, const unsigned char *dict, size_t dict_len, int max_window_bits,
                   unsigned char *out, size_t cap, size_t *out_len, size_t *in_consumed) {
    unsigned char window_buf[1 << 15];
    if (dict) memcpy(window_buf, dict, dict_len < ((size_t)1 << 15) ? dict_len : ((size_t)1 << 15));
    TampDecompressor d;
    size_t written = 0, consumed = 0;
    tamp_res res = tamp_decompressor_init(&d, NULL, window_buf, (uint8_t)max_window_bits);
    if (res == TAMP_OK) res = tamp_decompressor_decompress(&d, out, cap, &written, in, n, &consumed);
    if (out_len) *out_len = written;
    if (in_consumed) *in_consumed = consumed;
    return res;
}

/* ---- streaming calls, one reference object driven op by op (goldens for tests/golden/streaming.json) ---- */

typedef struct {
    TampCompressor c;
    unsigned char window[1 << 15];
} RefStream;

void *ref_stream_new(int window, int literal, int custom, int extended, int dictionary_reset, int append, int lazy,
                     const unsigned char *dict, int *res_out) {
    RefStream *s = (RefStream *)calloc(1, sizeof *s);
    TampConf conf;
    memset(&conf, 0, sizeof conf);
    conf.window = (uint16_t)window;
    conf.literal = (uint16_t)literal;
    conf.use_custom_dictionary = custom != 0;
    conf.extended = extended != 0;
    conf.dictionary_reset = dictionary_reset != 0;
    conf.append = append != 0;
#if TAMP_LAZY_MATCHING
    conf.lazy_matching = lazy != 0;
#else
    (void)lazy;
#endif
    if (custom && dict) memcpy(s->window, dict, (size_t)1 << window);
    int res = tamp_compressor_init(&s->c, &conf, s->window);
    if (res_out) *res_out = res;
    return s;
}

int ref_stream_write(void *h, const unsigned char *in, size_t n, unsigned char *out, size_t cap, size_t *written) {
    RefStream *s = (RefStream *)h;
    size_t consumed = 0;
    return tamp_compressor_compress(&s->c, out, cap, written, in, n, &consumed);
}

int ref_stream_flush(void *h, int write_token, unsigned char *out, size_t cap, size_t *written) {
    return tamp_compressor_flush(&((RefStream *)h)->c, out, cap, written, write_token != 0);
}

int ref_stream_reset_dictionary(void *h, unsigned char *out, size_t cap, size_t *written) {
    return tamp_compressor_reset_dictionary(&((RefStream *)h)->c, out, cap, written);
}

void ref_stream_free(void *h) { free(h); }

/* the same object below flush granularity: sink / full / poll / compress with the caller's output room */
int ref_stream_sink(void *h, const unsigned char *in, size_t n, size_t *consumed) {
    tamp_compressor_sink(&((RefStream *)h)->c, in, n, consumed);
    return 0;
}
int ref_stream_full(void *h) { return tamp_compressor_full(&((RefStream *)h)->c) ? 1 : 0; }
int ref_stream_poll(void *h, unsigned char *out, size_t cap, size_t *written) {
    return tamp_compressor_poll(&((RefStream *)h)->c, out, cap, written);
}
int ref_stream_compress(void *h, const unsigned char *in, size_t n, unsigned char *out, size_t cap, size_t *written,
                        size_t *consumed) {
    return tamp_compressor_compress(&((RefStream *)h)->c, out, cap, written, in, n, consumed);
}
int ref_stream_compress_and_flush(void *h, const unsigned char *in, size_t n, unsigned char *out, size_t cap,
                                  size_t *written, size_t *consumed, int write_token) {
    return tamp_compressor_compress_and_flush(&((RefStream *)h)->c, out, cap, written, in, n, consumed, write_token != 0);
}

/* ---- resumable decoder object, one tamp_decompressor_decompress call at a time ---- */

typedef struct {
    TampDecompressor d;
    unsigned char window[1 << 15];
} RefDecoder;

/* window < 0: conf = NULL (the header comes from the stream); dict, when given, pre-fills the window buffer */
void *ref_decoder_new(int window, int literal, int custom, int extended, int dictionary_reset, int window_bits,
                      const unsigned char *dict, size_t dict_len, int *res_out) {
    RefDecoder *s = (RefDecoder *)calloc(1, sizeof *s);
    if (dict) memcpy(s->window, dict, dict_len < sizeof s->window ? dict_len : sizeof s->window);
    TampConf conf;
    memset(&conf, 0, sizeof conf);
    if (window >= 0) {
        conf.window = (uint16_t)window;
        conf.literal = (uint16_t)literal;
        conf.use_custom_dictionary = custom != 0;
        conf.extended = extended != 0;
        conf.dictionary_reset = dictionary_reset != 0;
    }
    int res = tamp_decompressor_init(&s->d, window >= 0 ? &conf : NULL, s->window, (uint8_t)window_bits);
    if (res_out) *res_out = res;
    return s;
}

int ref_decoder_call(void *h, const unsigned char *in, size_t n, unsigned char *out, size_t cap, size_t *written,
                     size_t *consumed) {
    return tamp_decompressor_decompress(&((RefDecoder *)h)->d, out, cap, written, in, n, consumed);
}

void ref_decoder_free(void *h) { free(h); }

/* ---- multi-threaded batch drivers for the cpu_baseline leg of bench.py ---- */

typedef struct {
    int window, literal, custom, extended, lazy, decompress;
    const unsigned char *dict;
    size_t dict_len;
    const unsigned char *in;
    const uint64_t *in_off;
    const uint32_t *in_len;
    unsigned char *out;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    uint32_t *out_len;
    int8_t *status;
    size_t begin, end;
} Job;

static void *worker(void *arg) {
    Job *j = (Job *)arg;
    for (size_t i = j->begin; i < j->end; i++) {
        size_t w = 0;
        int r;
        if (j->decompress)
            r = ref_decompress(j->in + j->in_off[i], j->in_len[i], j->dict, j->dict_len, 15,
                               j->out + j->out_off[i], j->out_cap[i], &w, NULL);
        else
            r = ref_compress(j->window, j->literal, j->custom, j->extended, 0, j->lazy, j->dict, j->in + j->in_off[i],
                             j->in_len[i], j->out + j->out_off[i], j->out_cap[i], &w);
        j->out_len[i] = (uint32_t)w;
        j->status[i] = (int8_t)r;
    }
    return NULL;
}

/* Streams are cut into `n_threads` contiguous index ranges, one pthread each.  Returns wall seconds. */
double ref_batch(int decompress, int window, int literal, int custom, int extended, int lazy,
                 const unsigned char *dict, size_t dict_len, const unsigned char *in, const uint64_t *in_off,
                 const uint32_t *in_len, unsigned char *out, const uint64_t *out_off, const uint32_t *out_cap,
                 uint32_t *out_len, int8_t *status, size_t n_streams, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    Job *jobs = (Job *)malloc(sizeof(Job) * (size_t)n_threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) {
        Job *j = &jobs[t];
        j->window = window, j->literal = literal, j->custom = custom, j->extended = extended, j->lazy = lazy;
        j->decompress = decompress;
        j->dict = dict, j->dict_len = dict_len, j->in = in, j->in_off = in_off, j->in_len = in_len;
        j->out = out, j->out_off = out_off, j->out_cap = out_cap, j->out_len = out_len, j->status = status;
        j->begin = n_streams * (size_t)t / (size_t)n_threads;
        j->end = n_streams * (size_t)(t + 1) / (size_t)n_threads;
        pthread_create(&th[t], NULL, worker, j);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
