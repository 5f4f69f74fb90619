/*
 * tamp_compat.h -- the reference's OWN C symbol names, served by libtamp_amd.so.
 *
 * Struct layouts and prototypes follow the reference headers (paths relative to its repository root):
 *   TampConf .................................. tamp/_c_src/tamp/common.h:170-182 (lazy_matching field present, as in
 *                                               the Python wheel build, setup.py:49)
 *   TampCompressor (48 B), TampDecompressor (24 B)  compressor.h:13-66, decompressor.h:13-57 -- caller-allocated;
 *                                               only the `window` pointer is public, the rest is private here too
 *   tamp_compressor_init ...................... compressor.h:84
 *   tamp_compressor_sink / _full / _poll ...... compressor.h:101,154,142
 *   tamp_compressor_compress[_cb] ............. compressor.h:227,244
 *   tamp_compressor_compress_and_flush[_cb] ... compressor.h:259,280-286
 *   tamp_compressor_flush ..................... compressor.h:193
 *   tamp_compressor_reset_dictionary .......... compressor.h:217
 *   tamp_compress_stream ...................... compressor.h:338
 *   tamp_decompressor_read_header ............. decompressor.h:67
 *   tamp_decompressor_init .................... decompressor.h:83
 *   tamp_decompressor_decompress[_cb] ......... decompressor.h:93,128
 *   tamp_decompress_stream .................... decompressor.h:190
 *   tamp_callback_t / tamp_read_t / tamp_write_t, TampMemReader/Writer, tamp_stream_{mem,stdio}_{read,write}
 *                                               common.h:210-242,255-290 (host-only adaptors, common.c:92-132)
 *   tamp_initialize_dictionary, tamp_compute_min_pattern_size .... common.h:395,405 (declared in tamp_amd.h)
 *
 * Every codec call is a launch on HIP device $TAMP_AMD_DEVICE (default 0); there is no CPU code path.
 *
 * Compressor objects hold the reference's own state (TampAmdEncoderState in tamp_amd.h: bit buffer, 16-byte input
 * ring, pending RLE run / extended match, lazy cache, last_was_flush) in their 40 private bytes and the window in the
 * caller's buffer, so every call means what it means in the reference at any granularity: tamp_compressor_sink /
 * _full are ring operations on the host, tamp_compressor_poll / _compress[_cb] / _flush / _compress_and_flush[_cb] /
 * _reset_dictionary run on the device, with the reference's status, written and consumed counts also when the
 * output buffer is too small.  Calls below flush granularity are parsed token by token by one wavefront
 * (tamp_compress_resume_kernel.hpp); tamp_compressor_compress_and_flush of 2 KiB or more on an object that is
 * between segments, and tamp_compress_stream, go through the batch kernel's segment mode.
 *
 * Decompressor objects resume like the reference's: the 16 private bytes hold the same fields (TampAmdDecoderState in
 * tamp_amd.h), the window lives in the caller's buffer, and each tamp_decompressor_decompress[_cb] call is one step of
 * the device decoder -- any split of the input, any output room, TAMP_OUTPUT_FULL / TAMP_INPUT_EXHAUSTED and the
 * written / consumed counts exactly as decompressor.c:371-578 returns them.  tamp_decompress_stream is the reference's
 * loop over that call (decompressor.c:585-640) with larger work buffers.
 * The progress callback is invoked once per call (stream API: once per pulled chunk).
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
typedef int (*tamp_read_t)(void *handle, unsigned char *buffer, size_t size);        /* fread-like, <0 = error */
typedef int (*tamp_write_t)(void *handle, const unsigned char *buffer, size_t size); /* fwrite-like, <0 = error */

typedef struct TampMemReader {
    const unsigned char *data;
    size_t size;
    size_t pos; /* initialise to 0 */
} TampMemReader;

typedef struct TampMemWriter {
    unsigned char *data;
    size_t capacity;
    size_t pos; /* initialise to 0 */
} TampMemWriter;

int tamp_stream_mem_read(void *handle, unsigned char *buffer, size_t size);
int tamp_stream_mem_write(void *handle, const unsigned char *buffer, size_t size);
int tamp_stream_stdio_read(void *handle, unsigned char *buffer, size_t size);
int tamp_stream_stdio_write(void *handle, const unsigned char *buffer, size_t size);

tamp_res tamp_compressor_init(TampCompressor *compressor, const TampConf *conf, unsigned char *window);
void tamp_compressor_sink(TampCompressor *compressor, const unsigned char *input, size_t input_size,
                          size_t *consumed_size);
bool tamp_compressor_full(const TampCompressor *compressor);
tamp_res tamp_compressor_poll(TampCompressor *compressor, unsigned char *output, size_t output_size,
                              size_t *output_written_size);
tamp_res tamp_compressor_compress_cb(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                     size_t *output_written_size, const unsigned char *input, size_t input_size,
                                     size_t *input_consumed_size, tamp_callback_t callback, void *user_data);
tamp_res tamp_compressor_compress(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                  size_t *output_written_size, const unsigned char *input, size_t input_size,
                                  size_t *input_consumed_size);

tamp_res tamp_compressor_compress_and_flush_cb(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                               size_t *output_written_size, const unsigned char *input,
                                               size_t input_size, size_t *input_consumed_size, bool write_token,
                                               tamp_callback_t callback, void *user_data);

tamp_res tamp_compressor_compress_and_flush(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                            size_t *output_written_size, const unsigned char *input, size_t input_size,
                                            size_t *input_consumed_size, bool write_token);

tamp_res tamp_compressor_flush(TampCompressor *compressor, unsigned char *output, size_t output_size,
                               size_t *output_written_size, bool write_token);

tamp_res tamp_compressor_reset_dictionary(TampCompressor *compressor, unsigned char *output, size_t output_size,
                                          size_t *output_written_size);

tamp_res tamp_compress_stream(TampCompressor *compressor, tamp_read_t read_cb, void *read_handle,
                              tamp_write_t write_cb, void *write_handle, size_t *input_consumed_size,
                              size_t *output_written_size, tamp_callback_t callback, void *user_data);

tamp_res tamp_decompress_stream(TampDecompressor *decompressor, tamp_read_t read_cb, void *read_handle,
                                tamp_write_t write_cb, void *write_handle, size_t *input_consumed_size,
                                size_t *output_written_size, tamp_callback_t callback, void *user_data);

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
