/*
 * tamp_amd.h -- C ABI of libtamp_amd.so: the MI355X-native batch codec for the Tamp `.tamp` format.
 *
 * This is the drop-in boundary for the reference's hot path.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repository root).  Plain pointers and
 * sizes only; no torch / C++ types cross this boundary.  Every codec call runs hand-written HIP
 * kernels on a gfx950 device: there is no CPU fallback -- without a device the calls return
 * TAMP_AMD_NO_DEVICE (-20) and compute nothing.
 *
 * Status codes are the reference's tamp_res values (tamp/_c_src/tamp/common.h:145-168).
 */
#ifndef TAMP_AMD_H
#define TAMP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: identical numbering to tamp_res (common.h:145-168) ---- */
enum {
    TAMP_OK = 0,
    TAMP_OUTPUT_FULL = 1,
    TAMP_INPUT_EXHAUSTED = 2,
    TAMP_ERROR = -1,
    TAMP_EXCESS_BITS = -2,
    TAMP_INVALID_CONF = -3,
    TAMP_OOB = -4,
    TAMP_IO_ERROR = -10, /* stream API, common.h:160-163 */
    TAMP_READ_ERROR = -11,
    TAMP_WRITE_ERROR = -12,
    /* library-level failures of this implementation (outside the reference's range) */
    TAMP_AMD_NO_DEVICE = -20, /* no HIP device / HIP runtime error: nothing was computed */
    TAMP_AMD_BAD_ARGUMENT = -21,
};
typedef int8_t tamp_res;

#define TAMP_AMD_WINDOW_BITS_EXACT 0x80 /* flag for tamp_batch_decompress's max_window_bits: no header pre-pass */

/* Unpacked twin of TampConf (common.h:170-182).  One configuration per batch launch. */
typedef struct TampAmdConf {
    uint8_t window;                /* 8..15 */
    uint8_t literal;               /* 5..8  */
    uint8_t use_custom_dictionary; /* `dictionary` argument holds 1<<window bytes shared by all streams */
    uint8_t extended;              /* library default of the reference is 1 (compressor.c:193-203) */
    uint8_t dictionary_reset;      /* sets header bit0 and emits the zero second header byte */
    uint8_t lazy_matching;         /* compressor.c:576-619; a bit over half the default mode's speed */
    uint8_t input_hint;            /* TAMP_AMD_HINT_*: which build of the compress kernel parses a batch of SHORT messages
                                      (same bytes either way).  Streams of 1 KiB and more -- max_in_len given, or computed
                                      from in_len for host memory, 0 = unknown -- always take the run-aware build (round 3:
                                      it was the faster one on every kind of text, so the lean 256-thread builds are gone);
                                      shorter ones take the lean one-wavefront build unless the hint says RUNS. */
    uint8_t reserved;
} TampAmdConf;

enum {
    TAMP_AMD_HINT_AUTO = 0,
    TAMP_AMD_HINT_PLAIN = 1, /* lean build: every buffer position in the bigram index, every extended match searched;
                                the faster one for short messages (256-byte telemetry) and for data that is mostly runs */
    TAMP_AMD_HINT_RUNS = 2,  /* run-aware build: long runs of one byte (indentation, rulers, padding) are listed once
                                instead of being indexed byte by byte, extended matches without a rival are settled
                                without a window search: faster on streams of 1 KiB and more (config 2 by 5 %, source
                                code by 40 %) */
};

/* Where the data pointers of a batch call live. */
enum {
    TAMP_AMD_MEM_HOST = 0,   /* host pointers: the library stages H2D / D2H itself, in overlapping chunks on its
                                own streams; pinned memory (tamp_amd_host_alloc) lets both copy directions run
                                asynchronously at link speed, pageable memory works at the runtime's staging speed */
    TAMP_AMD_MEM_DEVICE = 1, /* device pointers on `device`: zero-copy, kernels only */
};

/* `device` argument of tamp_batch_compress / tamp_batch_decompress with host memory: cut the batch into one contiguous
 * range of streams per visible HIP device (balanced by input bytes) and run the ranges concurrently, one host thread
 * and one device each.  Streams are independent: no data moves between devices. */
#define TAMP_AMD_ALL_DEVICES (-1)

/* Page-locked host memory for TAMP_AMD_MEM_HOST calls, for callers that do not link the HIP runtime themselves
 * (cgo / JNI / ctypes).  Not in the reference: its buffers never leave the CPU.  NULL when the allocation fails. */
void *tamp_amd_host_alloc(size_t bytes);
void tamp_amd_host_free(void *p);

/* ---- host helpers (no device needed) -------------------------------------------------------- */

/* Replaces tamp_initialize_dictionary (common.h:395, common.c:37-52). */
void tamp_initialize_dictionary(unsigned char *buffer, size_t size, uint8_t literal);

/* Replaces tamp_compute_min_pattern_size (common.h:405, common.c:54-56). */
int8_t tamp_compute_min_pattern_size(uint8_t window, uint8_t literal);

/* Replaces tamp_window_copy (common.h:424, common.c:58-86): window[pos..pos+n) (wrapping with window_mask) <-
 * window[offset..offset+n) with memmove semantics; *window_pos advances.  Caller validates offset + n. */
void tamp_window_copy(unsigned char *window, uint16_t *window_pos, uint16_t window_offset, uint8_t match_size,
                      uint16_t window_mask);

/* Worst-case compressed size of an n-byte stream: header byte(s) + every byte a literal
 * (compressor.h flush table / SURVEY.md H7). */
size_t tamp_amd_compress_bound(size_t n, uint8_t literal, int dictionary_reset);

/* Number of visible HIP devices (0 when there is none), and the library version string. */
int tamp_amd_device_count(void);
const char *tamp_amd_version(void);

/* Text of the last HIP runtime failure seen by the calling thread ("" if none): what TAMP_AMD_NO_DEVICE meant. */
const char *tamp_amd_last_error(void);

/* How tamp_batch_compress would launch streams of up to `max_in_len` bytes (0 = unknown) at this window: positions matched
 * per epoch, LDS bytes per workgroup, threads per workgroup and the workgroups per CU its registers aim at (7 for the
 * run-aware builds, 6 lean, 5 lazy).  Pure host arithmetic -- no device needed; a diagnostics / capacity-planning call
 * (the reference has no counterpart: its state is the 16-byte ring + window of compressor.h:13-66).  Returns TAMP_OK or
 * TAMP_AMD_BAD_ARGUMENT. */
int tamp_amd_compress_plan(uint8_t window_bits, uint32_t max_in_len, int lazy_matching, uint32_t *block_positions,
                           uint32_t *lds_bytes, uint32_t *threads, uint32_t *workgroups_per_cu);

/* ---- batch codec ---------------------------------------------------------------------------- */

/*
 * Compress n_streams independent streams.  Stream i's bytes are identical to what
 *     tamp_compressor_init(&c, &conf, window)                             (compressor.h:84)
 *     tamp_compressor_compress_and_flush(&c, out, cap, &w, in, n, &consumed, false)
 *                                                                         (compressor.h:280-286)
 * produce in the reference on a freshly initialised compressor -- the call every reference
 * benchmark times (tools/c-profiler/main.c:52-54, devices/common/tamp_bench.c:118-121) and that
 * tamp.compress() performs (tamp/_c_compressor.pyx:189-199).
 *
 *   in[in_off[i] .. in_off[i]+in_len[i])        input bytes of stream i
 *   out[out_off[i] .. out_off[i]+out_cap[i])    output slab of stream i
 *   out_len[i]                                  bytes produced.  Slab bytes behind out_len[i] are unspecified after
 *                                               the call when the slabs tile the output buffer back to back (host
 *                                               memory: the results come back in one transfer per chunk); with gaps
 *                                               between slabs or permuted offsets only the out_len[i] produced bytes
 *                                               of each slab are written, nothing else in the caller's buffer
 *   status[i]                                   TAMP_OK, TAMP_OUTPUT_FULL (slab too small; out_len[i] <= out_cap[i]
 *                                               bytes of valid prefix), TAMP_EXCESS_BITS (a literal does not fit
 *                                               conf->literal bits, compressor.c:629-631; out_len[i] = whole bytes
 *                                               emitted before it), or TAMP_INVALID_CONF
 *   dictionary                                  1<<window bytes shared by every stream when
 *                                               conf->use_custom_dictionary, else NULL (each stream starts from
 *                                               its own pristine copy)
 *   max_in_len                                  upper bound on in_len[] (sizes the per-workgroup LDS block); 0 = unknown:
 *                                               computed from in_len for host memory, 4096-position blocks for
 *                                               device memory (any length still works, in several epochs)
 *   mem                                         TAMP_AMD_MEM_HOST or TAMP_AMD_MEM_DEVICE: applies to ALL pointer
 *                                               arguments except conf
 *   device                                      HIP device ordinal
 *   stream                                      hipStream_t (as void*) to enqueue on, or NULL for the default
 *                                               stream.  With device memory the call is asynchronous on `stream`;
 *                                               with host memory the call is synchronous and runs on the
 *                                               library's own streams (`stream` is not used).
 *                                               ONE EXCEPTION to "asynchronous": a device-memory call of at most 64
 *                                               streams with max_in_len >= 256 KiB in the v1 format (extended = 0,
 *                                               literal = 8, default parse) reads the streams' table rows back and
 *                                               WAITS for `stream` once before it launches (block mode: one long
 *                                               stream over all workgroups) -- such a call cannot be captured in a
 *                                               hipGraph; the environment variable TAMP_AMD_BLOCK_MIN=0 turns block
 *                                               mode off and restores the asynchronous batch kernel for them.
 * Returns TAMP_OK when the batch was launched (per-stream results are in status[]), or a negative
 * library-level code.
 */
int tamp_batch_compress(const TampAmdConf *conf, const uint8_t *dictionary, const uint8_t *in, const uint64_t *in_off,
                        const uint32_t *in_len, uint8_t *out, const uint64_t *out_off, const uint32_t *out_cap,
                        uint32_t *out_len, int8_t *status, size_t n_streams, uint32_t max_in_len, int mem, int device,
                        void *stream);

/*
 * Decompress n_streams independent `.tamp` streams.  Stream i's bytes and status are identical to
 *     tamp_decompressor_init(&d, NULL, window, max_window_bits)           (decompressor.h:83)
 *     tamp_decompressor_decompress(&d, out, cap, &w, in, n, &consumed)    (decompressor.h:128)
 * i.e. the configuration is read from each stream's own header (window 8..15 may differ per stream),
 * normal completion is TAMP_INPUT_EXHAUSTED (2) (decompressor.h:125-126), a full output slab with
 * work left is TAMP_OUTPUT_FULL (1), and malformed input yields TAMP_OOB / TAMP_INVALID_CONF
 * (decompressor.c:232-236,284,311,540-544) -- never a crash.
 *
 *   dictionary / dictionary_len   custom dictionary for streams whose header has the custom bit
 *                                 (>= 1<<window bytes, prefix used; such a stream with dictionary == NULL gets
 *                                 TAMP_INVALID_CONF, where the Python surface raises ValueError,
 *                                 tamp/_c_decompressor.pyx:63-64)
 *   max_window_bits               streams whose header asks for more get TAMP_INVALID_CONF (decompressor.c:311).  When it
 *                                 is above 8 the call first reads the headers of the batch (one tiny kernel and a
 *                                 4-byte copy, which waits for `stream`) to size the on-chip windows for the largest
 *                                 one actually present; OR in TAMP_AMD_WINDOW_BITS_EXACT to skip that and stay fully
 *                                 asynchronous (windows are then sized for max_window_bits itself)
 *   in_consumed                   optional (may be NULL): compressed bytes consumed per stream
 *   in_len[i]                     below 2^29 bytes (512 MiB) per stream: the decoders count bits in 32-bit registers;
 *                                 a longer stream gets status TAMP_AMD_BAD_ARGUMENT and produces nothing.  Feed longer
 *                                 streams in pieces through tamp_batch_decompress_resume, which looks at 2^28 bytes per
 *                                 call at most and reports what it consumed.
 */
int tamp_batch_decompress(const uint8_t *dictionary, size_t dictionary_len, uint8_t max_window_bits, const uint8_t *in,
                          const uint64_t *in_off, const uint32_t *in_len, uint8_t *out, const uint64_t *out_off,
                          const uint32_t *out_cap, uint32_t *out_len, int8_t *status, uint32_t *in_consumed,
                          size_t n_streams, int mem, int device, void *stream);

/* ---- resumable decoding: decoder OBJECTS that survive between calls ------------------------------
 *
 * What TampDecompressor is in the reference (decompressor.h:13-57): 16 bytes of state next to a window buffer,
 * advanced by tamp_decompressor_decompress (decompressor.h:93-128, decompressor.c:371-578) with whatever input and
 * output room one call has -- TAMP_OUTPUT_FULL / TAMP_INPUT_EXHAUSTED and pick up later, mid-token if need be.
 * Here the objects live in one array (host or device memory, like every other pointer of the call):
 *   object i = states + i * state_stride:  TampAmdDecoderState (16 B), then its window of 1 << window_bits_max bytes
 * state_stride >= tamp_amd_decoder_state_size(window_bits_max), a multiple of 16.  One call advances every object by
 * one step: object i sees in[in_off[i] .. +in_len[i]) and out_cap[i] bytes of room, and status / out_len /
 * in_consumed come back as from the reference's call.  Input that was not consumed must be offered again.
 */
typedef struct TampAmdDecoderState {
    uint32_t bit_buffer;            /* bits pulled from the input and not yet decoded, left aligned */
    uint16_t window_pos;
    uint8_t bit_buffer_pos;         /* how many */
    uint8_t token_state;            /* 0 none, 1 RLE, 2 extended match, 3 extended match with its size known */
    uint16_t pending_window_offset; /* token cut short by a full output buffer: where it copies from (RLE: its count) */
    uint16_t pending_match_size;
    uint8_t conf;                   /* header byte 0 once the header has been seen */
    uint8_t skip_bytes;             /* bytes of the pending token already delivered (before the header is
                                       complete: its stashed first byte) */
    uint8_t flags;                  /* 1 configured, 2 first header byte stashed, 4 last token was FLUSH */
    uint8_t window_bits_max;        /* capacity of the window buffer behind this state */
} TampAmdDecoderState;

size_t tamp_amd_decoder_state_size(uint8_t window_bits_max); /* 16 + (1 << window_bits_max) */

/* Replaces tamp_decompressor_init (decompressor.h:83, decompressor.c:331-347) for an object in HOST memory (copy it
 * to the device afterwards if the array lives there).  conf == NULL: the header comes from the stream.  With a conf
 * the window is seeded here unless conf->use_custom_dictionary, in which case the caller fills the window bytes
 * (at state + 16) with the dictionary, as with the reference's window buffer. */
tamp_res tamp_amd_decoder_state_init(void *state, const TampAmdConf *conf, uint8_t window_bits_max);

int tamp_batch_decompress_resume(void *states, size_t state_stride, uint8_t window_bits_max, const uint8_t *in,
                                 const uint64_t *in_off, const uint32_t *in_len, uint8_t *out, const uint64_t *out_off,
                                 const uint32_t *out_cap, uint32_t *out_len, int8_t *status, uint32_t *in_consumed,
                                 size_t n_streams, int mem, int device, void *stream);

/* ---- compressor objects below flush granularity -----------------------------------------------
 *
 * What TampCompressor is in the reference (compressor.h:13-66): a 16-byte input ring that is parsed only while it is
 * full, an RLE run / extended match that may still be growing, a lazily cached match, up to 31 pending output bits --
 * next to the window.  tamp_batch_compress_resume runs one of the reference's calls on every object of an array
 * (same layout rule as the decoder objects: state, then the window of 1 << window_bits_max bytes; state_stride a
 * multiple of 16 and >= tamp_amd_encoder_state_size):
 *   TAMP_AMD_OP_POLL                tamp_compressor_poll (compressor.h:142)            one parse step
 *   TAMP_AMD_OP_COMPRESS            tamp_compressor_compress_cb (compressor.h:227)     sink + poll while the ring fills
 *   TAMP_AMD_OP_FLUSH               tamp_compressor_flush (compressor.h:193)
 *   TAMP_AMD_OP_COMPRESS_AND_FLUSH  tamp_compressor_compress_and_flush_cb (compressor.h:259)
 * with the reference's results per object: status (TAMP_OK / TAMP_OUTPUT_FULL / TAMP_EXCESS_BITS), bytes written,
 * input bytes consumed -- including calls whose output buffer fills up.  Whole segments (everything between two
 * flush points, known up front) are the batch kernel's job: tamp_batch_compress / tamp_amd_compress_segment.
 */
typedef struct TampAmdEncoderState {
    uint32_t bit_buffer;      /* pending output bits, left aligned (the header sits here after init) */
    uint16_t window_pos;
    uint8_t bit_buffer_pos;
    uint8_t input_size;       /* bytes in the ring (0..16) */
    uint8_t input_pos;        /* ring read position */
    uint8_t window;           /* conf: window bits */
    uint8_t literal;          /* conf: literal bits */
    uint8_t flags;            /* conf: 1 custom dictionary, 2 extended, 4 dictionary_reset, 8 append, 16 lazy_matching */
    uint8_t input[16];        /* the ring */
    int16_t cached_match_index; /* lazy matching: match found for the next position, -1 = none */
    uint16_t extended_match_position;
    uint8_t cached_match_size;
    uint8_t rle_count;
    uint8_t extended_match_count;
    uint8_t last_was_flush;
    uint32_t reserved;
} TampAmdEncoderState;

enum {
    TAMP_AMD_OP_POLL = 1,
    TAMP_AMD_OP_COMPRESS = 2,
    TAMP_AMD_OP_FLUSH = 3,
    TAMP_AMD_OP_COMPRESS_AND_FLUSH = 4,
};

size_t tamp_amd_encoder_state_size(uint8_t window_bits_max); /* 40 + (1 << window_bits_max) */

/* Replaces tamp_compressor_init (compressor.h:84, compressor.c:191-244) for an object in HOST memory: header (or the
 * append marker) into the bit buffer, window seeded unless conf->use_custom_dictionary (then the caller fills the
 * window bytes at state + 40).  `append` as TampConf.append (compressor.c:227-235). */
tamp_res tamp_amd_encoder_state_init(void *state, const TampAmdConf *conf, int append, uint8_t window_bits_max);

int tamp_batch_compress_resume(void *states, size_t state_stride, uint8_t window_bits_max, int op, int write_token,
                               const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint8_t *out,
                               const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, int8_t *status,
                               uint32_t *in_consumed, size_t n_objects, int mem, int device, void *stream);

/* ---- single-stream one-shot entry points (the reference's own call shapes) ------------------- */

/*
 * One stream, host buffers: tamp_compressor_init + tamp_compressor_compress_and_flush(write_token=false)
 * (compressor.h:84,280-286) as a batch of one on `device`.  *output_written_size may be NULL.
 */
tamp_res tamp_amd_compress(const TampAmdConf *conf, const unsigned char *dictionary, unsigned char *output,
                           size_t output_size, size_t *output_written_size, const unsigned char *input,
                           size_t input_size, int device);

/*
 * One stream, host buffers: tamp_decompressor_init(conf=NULL) + tamp_decompressor_decompress
 * (decompressor.h:83,128) as a batch of one on `device`.
 */
tamp_res tamp_amd_decompress(const unsigned char *dictionary, size_t dictionary_len, unsigned char *output,
                             size_t output_size, size_t *output_written_size, const unsigned char *input,
                             size_t input_size, size_t *input_consumed_size, int device);

/*
 * One SEGMENT of a stream: the bytes between two flush points, with the window carried in and out.  It is what the
 * reference's streaming calls add up to between flushes -- tamp_compressor_init (compressor.h:84; header, or with
 * conf.append the FLUSH marker of compressor.c:227-235), any number of tamp_compressor_compress (compressor.h:244)
 * and one tamp_compressor_flush(write_token) (compressor.h:193; FLUSH rule compressor.c:776-794) -- and is what
 * tamp_amd.Compressor.write()/flush()/reset_dictionary() are built from.  A segment always ends byte aligned, so the
 * carried state is just the window (1<<window bytes, ring order) and window_pos.
 *   emit_header / append_marker   how the segment opens: header byte(s), the 2-byte FLUSH marker, or nothing
 *   resume                        0: fresh stream (window_state holds the custom dictionary if conf says so, else it is
 *                                 ignored on input); 1: continue from window_state / *window_pos
 *   flush_token                   write_token of tamp_compressor_flush; *token_written tells whether one was emitted
 * window_state / *window_pos are updated on TAMP_OK.  Host buffers, one stream, device `device`.
 */
tamp_res tamp_amd_compress_segment(const TampAmdConf *conf, int emit_header, int append_marker, int resume,
                                   int flush_token, unsigned char *window_state, uint16_t *window_pos,
                                   unsigned char *output, size_t output_size, size_t *output_written_size,
                                   const unsigned char *input, size_t input_size, int *token_written, int device);

/*
 * One PIECE of a stream -- a tamp_compressor_compress call on an object that lives on the host as (window_state,
 * *window_pos, *carry).  With finish = 0 the piece ends exactly as the reference's call does (compressor.c:681-722: every
 * byte is taken, parse steps run only while the 16-byte ring is full, whole output bytes leave, nothing is drained) and
 * *carry receives what the reference's object still holds: a run or extended match that is still growing
 * (compressor.h rle_count, extended_match_count / _position), up to 7 pending output bits, up to 15 unparsed input
 * bytes.  The next piece continues from it (resume = 1).  finish = 1 ends the segment like tamp_amd_compress_segment
 * (tamp_compressor_flush with write_token = flush_token) and clears the carry.  This is what bounded-memory writers are
 * built from: tamp_amd.Compressor.write() sends a piece whenever it has gathered enough and returns the bytes written,
 * as tamp/_c_compressor.pyx:74-118 does.  Not offered with conf->lazy_matching (the cached match of compressor.c:576-619
 * is not carried): TAMP_AMD_BAD_ARGUMENT.  Give a piece tamp_amd_compress_bound(input_size + 271, ...) bytes of room;
 * TAMP_OUTPUT_FULL leaves window_state / carry as they were.
 */
typedef struct TampAmdCarry {
    uint8_t rle_count;   /* bytes consumed by a run that has not ended yet */
    uint8_t ext_count;   /* bytes consumed by an extended match that has not ended yet ... */
    uint16_t ext_pos;    /* ... and its window position */
    uint8_t bit_count;   /* output bits not written yet: 0..7 on return; up to 31 accepted on entry (a reference object
                            sits on its last token's bits until the next poll, compressor.c:549-551) */
    uint8_t tail_len;    /* 0..16 input bytes taken but not parsed yet (0..15 on return) */
    uint16_t reserved;
    uint32_t bits;       /* the pending bits, left aligned: first one in bit 31 */
    uint8_t tail[16];
} TampAmdCarry;
tamp_res tamp_amd_compress_piece(const TampAmdConf *conf, int emit_header, int append_marker, int resume, int finish,
                                 int flush_token, unsigned char *window_state, uint16_t *window_pos, TampAmdCarry *carry,
                                 unsigned char *output, size_t output_size, size_t *output_written_size,
                                 const unsigned char *input, size_t input_size, int *token_written, int device);

/* Replaces tamp_decompressor_read_header (decompressor.h:67, decompressor.c:276-297): host-side header parse. */
tamp_res tamp_amd_read_header(TampAmdConf *conf, const unsigned char *input, size_t input_size,
                              size_t *input_consumed_size);

/* Timing hook for bench.py: duration in milliseconds of the most recent codec kernel launched by this
 * thread on `device`, measured with hipEvents on the stream the kernel ran on; < 0 if none recorded.
 * Enabled by tamp_amd_set_timing(1). */
void tamp_amd_set_timing(int enabled);
float tamp_amd_last_kernel_ms(void);

/* Releases the device scratch the library keeps between calls on `device` -- decoder window slabs and the split
 * decoder's token-record slab, one set per HIP stream that ever decoded (up to a quarter of the free device memory, 8 GiB
 * at most, per stream) -- and the staging buffers of the host-memory batch calls (pinned host memory sized by the largest
 * output extent ever staged, device chunk buffers).  Synchronises those streams first.  Returns the bytes released or a negative TAMP_AMD_* code.
 * (The library itself falls back to decoders without scratch when an allocation fails; this call is for callers that
 * want the memory back.) */
long long tamp_amd_trim(int device);

#ifdef __cplusplus
}
#endif
#endif /* TAMP_AMD_H */
