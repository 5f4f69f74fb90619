// tamp_decompress_long_kernel.hpp -- ONE long v1 stream decoded by the whole device (round 5; the decode side of block mode,
// tamp_compress_kernel.hpp BLOCKM).
//
// Replaces, for one stream, tamp_decompressor_decompress (tamp/_c_src/tamp/decompressor.c:371-578) where its token loop is
// position-independent: the v1 format without a dictionary reset -- a literal is 1 + literal bits, a match is the flag, a
// prefix code (decompressor.c:52-104) and window bits, FLUSH pads to the byte boundary (:501-514), and none of it depends
// on the window.  A wavefront per stream decodes such a stream at ~700 cycles per token (7 MB/s); here
//
//   1. the compressed bits are cut into chunks of kLongChunkBits and a lane per chunk parses from a GUESSED start; the
//      position where it leaves its chunk is the next chunk's start for the next round (tamp_long_sync_kernel).  Chunk 0
//      starts behind the header, so after round k the first k chunks are certainly right, and a prefix code re-synchronises
//      within a few tokens: the guesses stop changing after two to four rounds.  Unchanged guesses everywhere = all right;
//   2. a lane per chunk counts its tokens and bytes, the host cuts the chunks into groups of at most kSplitMaxOut output
//      bytes, a lane per chunk writes the split decoder's 32-bit records (tamp_long_parse_kernel);
//   3. the groups go through the split decoder's RESOLVE (tamp_decompress_split_kernel.hpp) IN ORDER, each with the W output
//      bytes in front of it as its "dictionary": a group is a stream whose window is rotated so that its write cursor starts
//      at 0 (window offsets are rotated with it when the records are written).
//
// Anything else -- extended format, dictionary reset, an out-of-bounds offset, an output buffer that is too small, a sync
// that does not settle -- is left to the exact decoders: the launcher falls back before anything has been written.
#pragma once
#include "tamp_common.hpp"
#include "tamp_decompress_split_kernel.hpp"

namespace tamp_amd {

constexpr uint32_t kLongChunkBits = 4096;        // 512 compressed bytes per lane: at most 341 tokens, 5,115 output bytes
constexpr uint32_t kLongEnd = 0xFFFFFFFFu;       // "the stream ended in front of this chunk"

struct LongArgs {
    const uint8_t* in;     // the stream (header included)
    uint32_t n;            // its bytes
    uint32_t first_bit;    // 8 x header bytes
    uint32_t n_chunks;
    uint32_t wbits, lbits;
    const uint32_t* g;     // n_chunks + 1 start positions (bits)
    uint32_t* g_next;      // sync: the next round's
    uint32_t* flags;       // [0] a guess changed  [1] an out-of-bounds offset was seen (parse)
    uint32_t* ntok;        // parse, counting: per chunk
    uint32_t* outb;
    const uint32_t* tokbase;  // parse, writing: first record of the chunk, rotation of its group
    const uint32_t* rot;
    uint32_t* recs;
    uint32_t write;        // parse: 0 = count, 1 = write records
};

// 32 bits of the stream from bit position t, MSb first (bytes behind the end read as zero)
__device__ __forceinline__ uint32_t long_bits(const uint8_t* in, uint32_t n, uint32_t t) {
    const uint32_t b = t >> 3;
    uint64_t v = 0;
    if (b + 5 <= n) {
        uint32_t lo;
        __builtin_memcpy(&lo, in + b, 4);
        v = ((uint64_t)__builtin_bswap32(lo) << 8) | in[b + 4];
    } else {
        for (uint32_t k = 0; k < 5; k++) v = (v << 8) | (b + k < n ? in[b + k] : 0u);
    }
    return (uint32_t)(v >> (8 - (t & 7)));
}

// One token at bit position t.  -> bits it takes (0: the stream ends here -- fewer bits left than the token needs, the
// reference returns TAMP_INPUT_EXHAUSTED, decompressor.c:431-445); rec = the split decoder's record, or 0xFFFFFFFF for FLUSH
// (no output), bad = an offset that runs out of the window (TAMP_OOB, :231-236,540-544).
__device__ __forceinline__ uint32_t long_token(const uint8_t* in, uint32_t n, uint32_t t, const uint8_t* lut, uint32_t wbits,
                                               uint32_t lbits, uint32_t minp, uint32_t& rec, bool& bad) {
    const uint32_t avail = 8 * n - t;
    rec = 0xFFFFFFFFu;
    if (avail == 0) return 0;
    const uint32_t w = long_bits(in, n, t);
    if (w >> 31) {  // literal, decompressor.c:466-482
        if (avail < 1 + lbits) return 0;
        rec = kRecLit | (1u << 2) | (((w << 1) >> (32 - lbits)) << 10);
        return 1 + lbits;
    }
    if (avail < 2) return 0;
    const bool coded = ((w >> 30) & 1u) != 0;
    const uint32_t e0 = lut[(w >> 23) & 0x7F];
    const uint32_t sym = coded ? (e0 & 15u) : 0u;
    const uint32_t used = coded ? 2 + (e0 >> 4) : 2u;
    if (avail < used) return 0;
    if (sym == (uint32_t)kSymFlush) return used + ((8 - ((t + used) & 7)) & 7);  // decompressor.c:501-514
    if (avail < used + wbits) return 0;
    const uint32_t off = (w << used) >> (32 - wbits), len = sym + minp;
    if (off + len > (1u << wbits)) bad = true;
    rec = kRecCopy | (len << 2) | (off << 10);
    return used + wbits;
}

__device__ __forceinline__ void long_lut(uint8_t* lut) {  // (the table of tamp_decode_parse_kernel)
    for (uint32_t v = threadIdx.x; v < 128; v += blockDim.x) {
        const uint64_t codes_lo = 0x2b2624140b080300ull, codes_hi = 0x00ab27aa9594544bull, nbits = 0x979998877765532ull;
        uint32_t entry = 0;
        for (int sy = 1; sy < 15; sy++) {
            const uint32_t l = (uint32_t)((nbits >> (4 * sy)) & 15) - 1u;
            const uint32_t code = (uint32_t)((sy < 8 ? codes_lo >> (8 * sy) : codes_hi >> (8 * (sy - 8))) & 0xFF);
            if ((code & ((1u << (l - 1)) - 1)) == (v >> (7 - (l - 1)))) entry = ((l - 1) << 4) | (uint32_t)sy;
        }
        lut[v] = (uint8_t)entry;
    }
    __syncthreads();
}

// Round of the start-position search: g_next[i + 1] = where the parse that starts at g[i] leaves chunk i.  The 64 chunks of a
// workgroup settle among themselves first (their starts travel through LDS, up to 64 inner rounds -- a parse that starts at the
// wrong phase of a PERIODIC bit stream, a run of equal tokens, never falls back into step, and the right start then moves on
// one chunk per round): a round of the host's loop carries it over 64 chunks at least.
__global__ void __launch_bounds__(64) tamp_long_sync_kernel(LongArgs a) {
    __shared__ uint8_t lut[128];
    __shared__ uint32_t gs[65];
    __shared__ uint32_t moved;
    long_lut(lut);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < a.n_chunks;
    const uint32_t minp = (uint32_t)min_pattern_size((int)a.wbits, (int)a.lbits);
    const uint32_t end = (i + 1) * kLongChunkBits;
    if (live) gs[threadIdx.x] = a.g[i];
    if (threadIdx.x == 63 || i + 1 == a.n_chunks) gs[threadIdx.x + 1] = live ? a.g[i + 1] : kLongEnd;
    __syncthreads();
    uint32_t seen = 0xFFFFFFFEu, t = 0;  // the start this lane parsed from last time, and where that parse left the chunk
    for (uint32_t inner = 0; inner < 64; inner++) {
        if (threadIdx.x == 0) moved = 0;
        __syncthreads();
        const uint32_t from = live ? gs[threadIdx.x] : kLongEnd;
        if (live && from != seen) {
            seen = from, t = from;
            while (t != kLongEnd && t < end) {
                uint32_t rec;
                bool bad = false;
                const uint32_t k = long_token(a.in, a.n, t, lut, a.wbits, a.lbits, minp, rec, bad);
                t = k ? t + k : kLongEnd;
            }
        }
        __syncthreads();
        if (live && gs[threadIdx.x + 1] != t) {
            gs[threadIdx.x + 1] = t;
            moved = 1;
        }
        __syncthreads();
        if (!moved) break;
    }
    if (live) {
        if (i == 0) a.g_next[0] = a.first_bit;  // (every other start is written by the chunk in front of it, possibly in another workgroup)
        a.g_next[i + 1] = gs[threadIdx.x + 1];
        if (gs[threadIdx.x + 1] != a.g[i + 1]) a.flags[0] = 1;
    }
}

// The tokens that START in chunk i, from its settled start: counted (write = 0) or written as records (write = 1).
__global__ void __launch_bounds__(64) tamp_long_parse_kernel(LongArgs a) {
    __shared__ uint8_t lut[128];
    long_lut(lut);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_chunks) return;
    const uint32_t minp = (uint32_t)min_pattern_size((int)a.wbits, (int)a.lbits);
    const uint32_t end = (i + 1) * kLongChunkBits, mask = (1u << a.wbits) - 1;
    uint32_t t = a.g[i], nt = 0, nb = 0;
    uint32_t* const out = a.write ? a.recs + a.tokbase[i] : nullptr;
    const uint32_t rot = a.write ? a.rot[i] : 0u;
    bool bad = false;
    while (t != kLongEnd && t < end) {
        uint32_t rec;
        const uint32_t k = long_token(a.in, a.n, t, lut, a.wbits, a.lbits, minp, rec, bad);
        if (!k) break;
        t += k;
        if (rec == 0xFFFFFFFFu) continue;  // FLUSH
        if (a.write) {
            // a group starts with its write cursor at window index `rot`: seen from there the offsets are (off - rot) mod W
            if ((rec & 3u) == kRecCopy) rec = (rec & 0x3FFu) | ((((rec >> 10) - rot) & mask) << 10);
            out[nt] = rec;
        }
        nt++;
        nb += (rec >> 2) & 0xFFu;
    }
    if (!a.write) a.ntok[i] = nt, a.outb[i] = nb;
    if (bad) a.flags[1] = 1;
}

// The window in front of a group that starts inside the first W output bytes, oldest byte first: output so far, and the
// initial dictionary where nothing has been written yet (window_pos of a fresh decoder = bytes written mod W).
__global__ void tamp_long_window_kernel(uint8_t* d, const uint8_t* out, const uint8_t* dict0, uint32_t v0, uint32_t W) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    const uint32_t r = (j + v0) & (W - 1);  // ring index of the j-th oldest byte
    d[j] = r < v0 ? out[r] : dict0[r];
}

__global__ void tamp_long_finish_kernel(uint32_t* out_len, int8_t* status, uint32_t* consumed, uint32_t total, uint32_t n) {
    out_len[0] = total;
    status[0] = (int8_t)kInputExhausted;
    if (consumed) consumed[0] = n;
}

}  // namespace tamp_amd
