/*
 * tamp_oracle.h -- CPU restatement of the tamp hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this.  The product (tamp_amd/, libtamp_amd.so) never does: it fails loudly
 * when its HIP library is missing.
 *
 * Parity status: PINNED.  Checked (tests/test_oracle_golden.py) against
 *   - every known-answer vector in the reference's own tests
 *     (tests/test_compressor.py, tests/test_decompressor.py, tests/test_bug_regressions.py,
 *      ctests/test_compressor.c, ctests/test_decompressor.c, tests/test_pseudorandom.py),
 *   - outputs of the reference C itself (oracle/_ref/libtamp_ref.so, built in place from
 *     /root/reference/tamp/_c_src by oracle/Makefile) committed under tests/golden/.
 *
 * Everything here is a one-shot, position-indexed restatement: a stream is a byte
 * array, "the 16-byte input ring" of the reference is the slice in[p .. p+R) with
 * R = min(16, n-p), and the emitter appends to a flat MSb-first bit string.
 */
#ifndef TAMP_ORACLE_H
#define TAMP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numeric values as the reference's tamp_res (tamp/_c_src/tamp/common.h:145-168). */
enum {
    ORACLE_OK = 0,
    ORACLE_OUTPUT_FULL = 1,
    ORACLE_INPUT_EXHAUSTED = 2,
    ORACLE_ERROR = -1,
    ORACLE_EXCESS_BITS = -2,
    ORACLE_INVALID_CONF = -3,
    ORACLE_OOB = -4,
};

/* Unpacked twin of TampConf (tamp/_c_src/tamp/common.h:170-182). */
typedef struct OracleConf {
    uint8_t window;                /* 8..15 */
    uint8_t literal;               /* 5..8  */
    uint8_t use_custom_dictionary; /* 0/1   */
    uint8_t extended;              /* 0/1   */
    uint8_t dictionary_reset;      /* 0/1 : header bit0 + zero second header byte */
    uint8_t lazy_matching;         /* 0/1   */
} OracleConf;

/* tamp_initialize_dictionary  (common.c:37-52) */
void oracle_initialize_dictionary(uint8_t *buf, size_t size, uint8_t literal);

/* tamp_compute_min_pattern_size  (common.c:54-56) */
int oracle_min_pattern_size(uint8_t window, uint8_t literal);

/*
 * One stream, one shot: bytes identical to
 *   tamp_compressor_init(&c, &conf, window)                       (compressor.c:191-245)
 *   tamp_compressor_compress_and_flush(&c, out, cap, &w, in, n, &consumed, false)
 *                                                                 (compressor.c:815-845)
 * `dict` must hold 1<<window bytes when conf->use_custom_dictionary, else may be NULL.
 * Returns ORACLE_OK / ORACLE_OUTPUT_FULL / ORACLE_EXCESS_BITS / ORACLE_INVALID_CONF.
 * On EXCESS_BITS *out_len is the count of whole bytes emitted before the offending
 * literal (what the reference has written when poll returns, compressor.c:550,629-631).
 */
int oracle_compress(const OracleConf *conf, const uint8_t *dict, const uint8_t *in, size_t n, uint8_t *out,
                    size_t cap, size_t *out_len);

/*
 * One SEGMENT of a stream (streaming surface, tamp/_c_compressor.pyx:70-170): the bytes the reference emits for
 *   [tamp_compressor_init: header (emit_header) or append marker (append_marker), compressor.c:227-241]
 *   tamp_compressor_compress(in, n) ... tamp_compressor_flush(write_token = flush_token)   (compressor.c:728-810)
 * with the window carried in window_state / *window_pos (ring order, like TampCompressor.window/window_pos).
 * resume = 0 seeds a fresh window (or takes the custom dictionary already in window_state).  The caller folds the
 * reference's last_was_flush rule (compressor.c:784) into flush_token; *token_written reports the outcome.
 */
int oracle_compress_segment(const OracleConf *conf, int emit_header, int append_marker, int resume, int flush_token,
                            uint8_t *window_state, uint16_t *window_pos, const uint8_t *in, size_t n, uint8_t *out,
                            size_t cap, size_t *out_len, int *token_written);

/*
 * One stream, one shot: same bytes and same status as
 *   tamp_decompressor_init(&d, NULL, window, max_window_bits)     (decompressor.c:331-347)
 *   tamp_decompressor_decompress(&d, out, cap, &w, in, n, &consumed)   (decompressor.c:371-578)
 * Normal completion is ORACLE_INPUT_EXHAUSTED (2), exactly like the reference
 * (decompressor.h:125-126); ORACLE_OUTPUT_FULL when `cap` is reached with work left.
 * `dict` (>= 1<<window bytes) seeds the window when the header's custom bit is set;
 * a custom-bit stream with dict == NULL yields ORACLE_INVALID_CONF (the Python surface
 * raises ValueError there, tamp/_c_decompressor.pyx:63-64).
 */
int oracle_decompress(const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_len, uint8_t max_window_bits,
                      uint8_t *out, size_t cap, size_t *out_len, size_t *in_consumed);

/* Resumable decoder = the reference's TampDecompressor object by value (decompressor.h:13-57; 16 bytes of state next
 * to the caller's window buffer).  oracle_decoder_init follows tamp_decompressor_init (decompressor.c:331-347; conf
 * NULL = read the header from the stream), oracle_decoder_call one tamp_decompressor_decompress call
 * (decompressor.c:371-578): any input size, any output room, OUTPUT_FULL / INPUT_EXHAUSTED and pick up later. */
typedef struct OracleDecoder {
    uint32_t bit_buffer;
    uint16_t window_pos;
    uint8_t bit_buffer_pos;
    uint8_t token_state;
    uint16_t pending_window_offset;
    uint16_t pending_match_size;
    uint8_t conf;       /* header byte 0 once configured */
    uint8_t skip_bytes; /* before the header is complete: its stashed first byte */
    uint8_t flags;      /* 1 configured, 2 header byte stashed, 4 last token was FLUSH */
    uint8_t window_bits_max;
} OracleDecoder;
int oracle_decoder_init(OracleDecoder *d, uint8_t *window, const OracleConf *conf, uint8_t window_bits_max);
int oracle_decoder_call(OracleDecoder *d, uint8_t *window, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                        size_t *written, size_t *consumed);

/* Per-token trace hook for debugging the HIP path (kind: 0 literal, 1 match, 2 rle, 3 ext). */
typedef void (*oracle_token_cb)(void *user, int kind, size_t in_pos, unsigned len, unsigned index);
void oracle_set_token_cb(oracle_token_cb cb, void *user);

#ifdef __cplusplus
}
#endif
#endif
