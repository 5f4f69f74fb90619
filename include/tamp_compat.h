/*
 * tamp_compat.h -- the reference's OWN C symbol names for the one-shot path, served by libtamp_amd.so.
 *
 * Struct layouts and prototypes follow the reference headers (paths relative to its repository root):
 *   TampConf .................................. tamp/_c_src/tamp/common.h:170-182 (lazy_matching field present, as in
 *                                               the Python wheel build, setup.py:49)
 *   TampCompressor (48 B), TampDecompressor (24 B)  compressor.h:13-66, decompressor.h:13-57 -- caller-allocated;
 *                                               only the `window` pointer is public, the rest is private here too
 *   tamp_compressor_init ...................... compressor.h:84
 *   tamp_compressor_compress_and_flush[_cb] ... compressor.h:259,280-286
 *   tamp_decompressor_read_header ............. decompressor.h:67
 *   tamp_decompressor_init .................... decompressor.h:83
 *   tamp_decompressor_decompress[_cb] ......... decompressor.h:93,128
 *   tamp_initialize_dictionary, tamp_compute_min_pattern_size .... common.h:395,405 (declared in tamp_amd.h)
 *
 * Each call is a batch of one stream on HIP device $TAMP_AMD_DEVICE (default 0); there is no CPU code path.
 * What a compressor/decompressor object supports in this release is ONE whole-stream call after init:
 *   - tamp_compressor_compress_and_flush on a freshly initialised compressor, write_token = false
 *     (exactly what tamp.compress() and every reference benchmark do);
 *   - tamp_decompressor_decompress with the complete stream (conf read from the header, or passed to init with the
 *     input starting after the header, as tamp/_c_decompressor.pyx:50-75 does).
 * Anything that needs state carried between calls (sink / poll / flush with a FLUSH token / reset_dictionary / a
 * second compress or decompress call / the callback stream API) returns TAMP_ERROR instead of computing on the
 * host; that is SURVEY.md section 8(f) rows 2-3.  Progress callbacks are invoked once, at completion.
 */
#ifndef TAMP_COMPAT_H
#define TAMP_COMPAT_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "tamp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct TampConf {
    uint16_t window : 4;
    uint16_t literal : 4;
    uint16_t use_custom_dictionary : 1;
    uint16_t extended : 1;
    uint16_t dictionary_reset : 1;
    uint16_t append : 1;
    uint16_t lazy_matching : 1;
} TampConf;

typedef struct TampCompressor {
    unsigned char *window; /* caller's 1<<window byte buffer (seeded by init unless use_custom_dictionary) */
    unsigned char private_[40];
} TampCompressor;

typedef struct TampDecompressor {
    unsigned char *window; /* caller's window buffer: holds the custom dictionary when the stream uses one */
    unsigned char private_[16];
} TampDecompressor;

typedef int (*tamp_callback_t)(void *user_data, size_t bytes_processed, size_t total_bytes);

tamp_res tamp_compressor_init(TampCompressor *compressor, const TampConf *conf, unsigned char *window);

tamp_res tamp_compressor_compress_and_flush_cb(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                               size_t *output_written_size, const unsigned char *input,
                                               size_t input_size, size_t *input_consumed_size, bool write_token,
                                               tamp_callback_t callback, void *user_data);

tamp_res tamp_compressor_compress_and_flush(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                            size_t *output_written_size, const unsigned char *input, size_t input_size,
                                            size_t *input_consumed_size, bool write_token);

tamp_res tamp_decompressor_read_header(TampConf *conf, const unsigned char *input, size_t input_size,
                                       size_t *input_consumed_size);

tamp_res tamp_decompressor_init(TampDecompressor *decompressor, const TampConf *conf, unsigned char *window,
                                uint8_t window_bits);

tamp_res tamp_decompressor_decompress_cb(TampDecompressor *decompressor, unsigned char *output, size_t output_size,
                                         size_t *output_written_size, const unsigned char *input, size_t input_size,
                                         size_t *input_consumed_size, tamp_callback_t callback, void *user_data);

tamp_res tamp_decompressor_decompress(TampDecompressor *decompressor, unsigned char *output, size_t output_size,
                                      size_t *output_written_size, const unsigned char *input, size_t input_size,
                                      size_t *input_consumed_size);

#ifdef __cplusplus
}
#endif
#endif /* TAMP_COMPAT_H */
