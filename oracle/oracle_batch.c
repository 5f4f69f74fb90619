/*
 * oracle_batch.c -- multi-threaded batch driver over the CPU restatement (tamp_oracle.c).
 * TEST / BENCH ONLY: used by tests as the per-stream checker for large batches and by
 * bench.py's cpu_baseline leg (kind "port") when oracle/_ref is not available.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include "tamp_oracle.h"

typedef struct {
    int decompress;
    OracleConf conf;
    const uint8_t *dict;
    size_t dict_len;
    const uint8_t *in;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out;
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
            r = oracle_decompress(j->in + j->in_off[i], j->in_len[i], j->dict, j->dict_len, 15, j->out + j->out_off[i],
                                  j->out_cap[i], &w, NULL);
        else
            r = oracle_compress(&j->conf, j->dict, j->in + j->in_off[i], j->in_len[i], j->out + j->out_off[i],
                                j->out_cap[i], &w);
        j->out_len[i] = (uint32_t)w;
        j->status[i] = (int8_t)r;
    }
    return NULL;
}

/* Same CSR contract as tamp_batch_* in include/tamp_amd.h; returns wall seconds. */
double oracle_batch(int decompress, int window, int literal, int custom, int extended, int lazy, const uint8_t *dict,
                    size_t dict_len, const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint8_t *out,
                    const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, int8_t *status,
                    size_t n_streams, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    Job *jobs = (Job *)malloc(sizeof(Job) * (size_t)n_threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) {
        Job *j = &jobs[t];
        j->decompress = decompress;
        j->conf.window = (uint8_t)window, j->conf.literal = (uint8_t)literal;
        j->conf.use_custom_dictionary = (uint8_t)custom, j->conf.extended = (uint8_t)extended;
        j->conf.dictionary_reset = 0, j->conf.lazy_matching = (uint8_t)lazy;
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
