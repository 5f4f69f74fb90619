// tamp_decompress_long_kernel.hpp -- ONE long stream decoded by the whole device (v1: round 5, the decode side of block mode,
// tamp_compress_kernel.hpp BLOCKM; the extended format: round 6).
//
// Replaces, for one stream, tamp_decompressor_decompress (tamp/_c_src/tamp/decompressor.c:371-578) where its token loop is
// position-independent: the v1 format without a dictionary reset -- a literal is 1 + literal bits, a match is the flag, a
// prefix code (decompressor.c:52-104) and window bits, FLUSH pads to the byte boundary (:501-514), and none of it depends
// on the window.  A wavefront per stream decodes such a stream at ~700 cycles per token (7 MB/s; this path: 10 GB/s); here
//
//   1. the compressed bits are cut into chunks of kLongChunkBits and a lane per chunk parses from a GUESSED start; the
//      position where it leaves its chunk is the next chunk's start for the next round (tamp_long_sync_kernel; the 64 chunks
//      of a workgroup settle among themselves first).  Chunk 0 starts behind the header, so after round k the first k chunk
//      rows are certainly right, and a prefix code re-synchronises within a few tokens: text settles in two rounds of the
//      host's loop.  Unchanged guesses everywhere = all right (tests/test_host_logic.py restates this on the CPU);
//   2. a lane per chunk counts its tokens and bytes, the host cuts the chunks into groups of at most kLongGroupOut output
//      bytes, a lane per chunk writes the split decoder's 32-bit records (tamp_long_parse_kernel);
//   3. the groups are resolved by a workgroup each, all at once, and none waits for another: what a group takes from the
//      stream in front of it is the WINDOW as it stood when the group began, and those windows follow from the groups' tail
//      maps by one cheap pass (tails / scan / finish, see tamp_long_resolve_kernel below).  A group is a stream whose window is
//      rotated so that its write cursor starts at 0 (window offsets are rotated with it when the records are written).
//      (TAMP_AMD_LONGDEC_CHAIN=0: groups of kSplitMaxOut bytes through the split decoder's RESOLVE, one launch after the
//      other, each with the W output bytes in front of it as its "dictionary".)
//
// The extended format (round 6) adds RLE / extended-match tokens, which write fewer bytes to the window than they produce
// (decompressor.c:162-170, 266-268): window_pos at those tokens comes from tamp_long_wp_kernel, the groups get the split decoder's
// lag lists.  Scalar model of steps 3 and of the window_pos pass: tests/test_host_logic.py
// (test_long_stream_groups_tail_maps_and_window_pos_blocks_model).
// Anything else -- dictionary reset, an out-of-bounds offset, an output buffer that is too small, a sync that does not settle,
// more lagging tokens in one chunk than a group lists -- is left to the exact decoders: the launcher falls back before anything
// has been written.
#pragma once
#include "tamp_common.hpp"
#include "tamp_decompress_split_kernel.hpp"

namespace tamp_amd {

constexpr uint32_t kLongChunkBits = 4096;        // v1: 512 compressed bytes per lane: at most 341 tokens, 5,115 output bytes
// extended format: an RLE token is 12 bits for up to 241 bytes -- 128 compressed bytes per lane keep a chunk below 21 KB of output,
// so that whole chunks still fit a group (kLongGroupOut)
constexpr uint32_t kLongChunkBitsExt = 1024;
constexpr uint32_t kLongEnd = 0xFFFFFFFFu;       // "the stream ended in front of this chunk"
constexpr uint32_t kLongMaxLag = 64;             // slots of a group's lag list in LDS (a power of two: six-step binary searches) ...
constexpr uint32_t kLongLagCap = kLongMaxLag - 1;  // ... of which a group uses at most 63: the searches count up to 32 + 16 + .. + 1
                                                   // (more lagging tokens in ONE chunk: the exact decoder takes the stream)

struct LongArgs {
    const uint8_t* in;     // the stream (header included)
    uint32_t n;            // its bytes
    uint32_t first_bit;    // 8 x header bytes
    uint32_t n_chunks;
    uint32_t chunk_bits;   // kLongChunkBits or kLongChunkBitsExt
    uint32_t wbits, lbits;
    uint32_t extended;     // RLE / extended-match tokens (decompressor.c:114-273)
    const uint32_t* g;     // n_chunks + 1 start positions (bits)
    uint32_t* g_next;      // sync: the next round's
    uint32_t* flags;       // [0] a guess changed  [1] an out-of-bounds offset was seen (parse)
    uint32_t* ntok;        // parse, counting: per chunk
    uint32_t* outb;
    uint32_t* nspec;       // ... RLE / extended-match tokens of the chunk (extended format)
    const uint32_t* tokbase;  // parse, writing: first record of the chunk, rotation of its group
    const uint32_t* rot;
    uint32_t* recs;
    uint32_t write;        // parse: 0 = count, 1 = write records, 2 = list the chunk's RLE / extended-match tokens
    // extended format: the list of tokens that may write fewer bytes to the window than they produce -- per chunk its RLE /
    // extended-match tokens and one end marker -- walked in stream order by tamp_long_wp_kernel
    const uint32_t* specbase;  // per chunk: index of its first list entry (nspec + 1 entries per chunk)
    uint32_t* spec_gap;        // bytes produced by plain tokens in front of the entry (since the previous one / the chunk's start)
    uint32_t* spec_kl;         // bytes the token produces | RLE << 31; 0 = the chunk's end marker
    uint32_t* spec_written;    // tamp_long_wp_kernel: bytes it writes to the window (decompressor.c:162-170,266-268)
    uint32_t* chunk_lag;       // tamp_long_wp_kernel, per chunk: cumulative lag behind it, lagging tokens in it (n_chunks x 2)
    // records pass, extended format: where the chunk stands inside its group
    const uint32_t* chunk_o0;    // output bytes of the group in front of the chunk
    const uint32_t* chunk_lag0;  // lag of the group in front of the chunk
    const uint32_t* lagbase;     // index of the chunk's first entry in the lag lists
    uint32_t* lag;               // lag lists, two words per lagging token (the split decoder's: Oend | Vend << 16, cumulative lag)
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
                                               uint32_t lbits, uint32_t minp, bool extended, uint32_t& rec, bool& bad) {
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
    if (extended && sym >= (uint32_t)kSymRle) {
        // RLE: a second prefix code + 4 bits -> count - 2 (decompressor.c:114-174); extended match: code + 3 bits -> size - min - 12,
        // then the window offset (:187-273).  How many bits the token takes depends on nothing but its own bits.
        if (avail < used + 1) return 0;
        const uint32_t w2 = long_bits(in, n, t + used);
        const bool coded2 = (w2 >> 31) != 0;
        const uint32_t e2 = lut[(w2 >> 24) & 0x7F];
        const uint32_t h = coded2 ? (e2 & 15u) : 0u, trailing = sym == (uint32_t)kSymRle ? 4u : 3u;
        uint32_t u = coded2 ? 1 + (e2 >> 4) : 1u;
        if (avail < used + u + trailing) return 0;
        const uint32_t value = (h << trailing) + ((w2 << u) >> (32 - trailing));
        u += trailing;
        if (sym == (uint32_t)kSymRle) {
            rec = kRecFill | ((value + 2) << 2);
            return used + u;
        }
        if (avail < used + u + wbits) return 0;
        const uint32_t off = long_bits(in, n, t + used + u) >> (32 - wbits), len = value + minp + 12;
        if (off + len > (1u << wbits)) bad = true;
        rec = kRecCopyExt | (len << 2) | (off << 10);
        return used + u + wbits;
    }
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
    const uint32_t end = (i + 1) * a.chunk_bits;
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
                const uint32_t k = long_token(a.in, a.n, t, lut, a.wbits, a.lbits, minp, a.extended != 0, rec, bad);
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

// The tokens that START in chunk i, from its settled start: counted (write = 0), listed where they can lag (write = 2, extended
// format), or written as records (write = 1).
__global__ void __launch_bounds__(64) tamp_long_parse_kernel(LongArgs a) {
    __shared__ uint8_t lut[128];
    long_lut(lut);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_chunks) return;
    const uint32_t minp = (uint32_t)min_pattern_size((int)a.wbits, (int)a.lbits);
    const uint32_t end = (i + 1) * a.chunk_bits, mask = (1u << a.wbits) - 1;
    const bool ext = a.extended != 0;
    uint32_t t = a.g[i], nt = 0, nb = 0, ns = 0;
    uint32_t* const out = a.write == 1 ? a.recs + a.tokbase[i] : nullptr;
    const uint32_t rot = a.write == 1 ? a.rot[i] : 0u;
    // extended format: this chunk's entries of the special-token list (write = 2: filled in; write = 1: read back with what
    // tamp_long_wp_kernel found), and where the chunk stands inside its group (write = 1)
    uint32_t sb = 0, gap = 0, O = 0, cl = 0, li = 0;
    if (ext && a.write) sb = a.specbase[i];
    if (ext && a.write == 1) O = a.chunk_o0[i], cl = a.chunk_lag0[i], li = a.lagbase[i];
    bool bad = false;
    while (t != kLongEnd && t < end) {
        uint32_t rec;
        const uint32_t k = long_token(a.in, a.n, t, lut, a.wbits, a.lbits, minp, ext, rec, bad);
        if (!k) break;
        t += k;
        if (rec == 0xFFFFFFFFu) continue;  // FLUSH
        const uint32_t kind = rec & 3u, olen = (rec >> 2) & 0xFFu;
        const bool special = kind == kRecFill || kind == kRecCopyExt;
        if (a.write == 2) {
            if (special) {
                a.spec_gap[sb + ns] = gap, a.spec_kl[sb + ns] = olen | (kind == kRecFill ? 0x80000000u : 0u);
                gap = 0;
            } else {
                gap += olen;
            }
        }
        if (a.write == 1) {
            // a group starts with its write cursor at window index `rot`: seen from there the offsets are (off - rot) mod W
            if (kind == kRecCopy || kind == kRecCopyExt) rec = (rec & 0x3FFu) | ((((rec >> 10) - rot) & mask) << 10);
            out[nt] = rec;
            O += olen;
            if (special) {
                const uint32_t wr = a.spec_written[sb + ns];
                if (wr < olen) {  // a lag: the split decoder's list entry, positions counted from the group's start
                    cl += olen - wr;
                    a.lag[2 * li] = (O & 0xFFFFu) | (((O - cl) & 0xFFFFu) << 16);
                    a.lag[2 * li + 1] = cl;
                    li++;
                }
            }
        }
        if (special) ns++;
        nt++;
        nb += olen;
    }
    if (a.write == 0) {
        a.ntok[i] = nt, a.outb[i] = nb;
        if (ext) a.nspec[i] = ns;
    }
    if (a.write == 2) a.spec_gap[sb + ns] = gap, a.spec_kl[sb + ns] = 0;  // the chunk's end marker
    if (bad) a.flags[1] = 1;
}

// Extended format: window_pos at every token that can lag.  An RLE token writes min(count, 8, W - window_pos) bytes to the window
// and an extended match min(size, W - window_pos) (decompressor.c:162-170, 266-268: clipped at the ring's end, no wrap) -- the one
// place where the token stream depends on the window, and only through the scalar window_pos: a chain over the RLE / extended-
// match tokens (a few per cent of all tokens), with the byte counts of the plain tokens between them.  One lane walking the whole
// list took 167 ms for a 100 MB stream (560 cycles an entry: dependent LDS reads in a lone wavefront); the chain is cut into blocks
// of kLongWpBlock entries instead:
//   MODE 0  a workgroup per block runs the block from EVERY possible window_pos at once (a thread per start value: W of them, the
//           entries broadcast from LDS): where window_pos ends up, the lag added, the lagging tokens behind the block's last chunk
//           marker -- three tables of W entries per block;
//   MODE 1  one lane walks the blocks' tables (a look-up per block): the state in front of every block;
//   MODE 2  a lane per block walks its entries from that state and writes what every token wrote and the chunks' totals.
constexpr uint32_t kLongWpBlock = 2048;
struct LongWpArgs {
    LongArgs a;
    uint32_t n_entries, n_blocks;
    uint16_t* f_wp;        // n_blocks x W: window_pos behind the block, by window_pos in front of it
    uint32_t* f_cum;       // ... lag the block adds
    uint16_t* f_nl;        // ... lagging tokens behind its last chunk marker (the whole block's when it has none)
    uint32_t* markers;     // n_blocks: chunk markers in the block
    uint32_t* state;       // n_blocks x 4: window_pos, lag, chunk, lagging tokens of the open chunk -- in front of the block
};
template <int MODE>
__global__ void __launch_bounds__(1024) tamp_long_wp_kernel(LongWpArgs x) {
    const LongArgs& a = x.a;
    __shared__ uint32_t s_gap[kLongWpBlock], s_kl[kLongWpBlock], s_wr[MODE == 2 ? kLongWpBlock : 1];
    const uint32_t W = 1u << a.wbits, mask = W - 1;
    if constexpr (MODE == 1) {
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            uint32_t wp = 0, cum = 0, chunk = 0, nl = 0;
            for (uint32_t b = 0; b < x.n_blocks; b++) {
                x.state[4 * b] = wp, x.state[4 * b + 1] = cum, x.state[4 * b + 2] = chunk, x.state[4 * b + 3] = nl;
                const size_t at = (size_t)b * W + wp;
                const uint32_t m = x.markers[b], tail = x.f_nl[at];
                cum += x.f_cum[at];
                nl = m ? tail : nl + tail;
                chunk += m;
                wp = x.f_wp[at];
            }
        }
        return;
    } else {
        const uint32_t base = blockIdx.x * kLongWpBlock;
        const uint32_t cnt = min(kLongWpBlock, x.n_entries - base);
        for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) s_gap[k] = a.spec_gap[base + k], s_kl[k] = a.spec_kl[base + k];
        __syncthreads();
        auto run = [&](uint32_t wp, uint32_t cum, uint32_t chunk, uint32_t nl, uint32_t& wp_out, uint32_t& cum_out, uint32_t& nl_out,
                       uint32_t& nmark) {
            nmark = 0;
            for (uint32_t k = 0; k < cnt; k++) {
                const uint32_t kl = s_kl[k];
                wp = (wp + s_gap[k]) & mask;
                if (kl == 0) {  // the end of a chunk
                    if constexpr (MODE == 2) a.chunk_lag[2 * (size_t)chunk] = cum, a.chunk_lag[2 * (size_t)chunk + 1] = nl;
                    chunk++, nl = 0, nmark++;
                    continue;
                }
                const uint32_t L = kl & 0xFFFFu;
                uint32_t w = (kl >> 31) ? min(L, kRleWindowMax) : L;
                w = min(w, W - wp);
                if constexpr (MODE == 2) s_wr[k] = w;
                if (w < L) cum += L - w, nl++;
                wp = (wp + w) & mask;
            }
            wp_out = wp, cum_out = cum, nl_out = nl;
        };
        if constexpr (MODE == 0) {
            for (uint32_t wp_in = threadIdx.x; wp_in < W; wp_in += blockDim.x) {
                uint32_t wp, cum, nl, nm;
                run(wp_in, 0, 0, 0, wp, cum, nl, nm);
                const size_t at = (size_t)blockIdx.x * W + wp_in;
                x.f_wp[at] = (uint16_t)wp, x.f_cum[at] = cum, x.f_nl[at] = (uint16_t)nl;
                if (wp_in == 0) x.markers[blockIdx.x] = nm;
            }
        } else {
            if (threadIdx.x == 0) {
                uint32_t wp, cum, nl, nm;
                run(x.state[4 * blockIdx.x], x.state[4 * blockIdx.x + 1], x.state[4 * blockIdx.x + 2], x.state[4 * blockIdx.x + 3], wp, cum, nl, nm);
            }
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) a.spec_written[base + k] = s_wr[k];
        }
    }
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

// ---------------------------------------------------------------------------------------------------------------
// Step 3: every group of at most kLongGroupOut output bytes by a workgroup of its own, and NO workgroup waits for another
// (round 6; round 5 chained the groups through one flag each inside one launch -- ~13 us per link, 39 ms of a 100 MB stream
// was that chain, and its freedom from deadlock rested on the order in which workgroups are dispatched).
// What a group needs from the stream in front of it is only the WINDOW as it stood when the group began: the last W bytes
// written.  Everything inside the group -- the split decoder's token pass, byte pass and pointer jumping
// (tamp_decode_resolve_kernel, without lags: the v1 format has none) -- is independent of it: a byte whose source lies in front
// of the group becomes EXTERNAL (bit 15 of its pointer, the index of the byte in that window, oldest first, below it) and
// pointer jumping carries the mark to every byte that ends there.  Three launches:
//   tails   (tamp_long_resolve_kernel<1>)  a workgroup per group resolves the group and writes its TAIL MAP: for each of the W
//           bytes of the window as the group leaves it, the byte itself or "byte j of the window in front" (a group shorter
//           than W passes the rest of that window on the same way);
//   scan    (tamp_long_tail_scan_kernel)   the maps compose: window after group g = map g applied to the window after group
//           g-1, starting from the fresh decoder's window.  One workgroup walks the groups, W gathers in LDS per step, and
//           stores every group's last bytes where they belong in the output (~0.3 us a group);
//   finish  (tamp_long_resolve_kernel<2>)  a workgroup per group resolves it again, takes its external bytes from the output
//           in front of it -- written by the scan launch, or the fresh window where the stream is younger than W -- and stores.
// The resolve runs twice (compute, ~50 us a group, all groups at once) so that nothing of it has to travel through HBM.
// ---------------------------------------------------------------------------------------------------------------
struct LongGroup {
    unsigned long long v0;  // output position of the group's first byte
    uint32_t tok0, ntok, nout;
    uint32_t lag0, nlag, pad;  // extended format: the group's entries of the lag lists (at most kLongMaxLag)
};
struct LongResolveArgs {
    const uint32_t* recs;
    const LongGroup* groups;
    uint8_t* out;            // the stream's output
    const uint8_t* dict0;    // the window of a fresh decoder (custom dictionary or the seeded default)
    uint16_t* tailmap;       // n_groups x W: byte k of the window behind group g = the byte itself, or kLongExt | j (see above)
    const uint32_t* lag;     // extended format: lag lists (two words per lagging token, positions counted from the group's start)
    uint8_t* groupwin;       // extended format: n_groups x W, the window in front of every group (the scan launch writes it, finish
                             // reads it: with lags the window is no longer "the last W output bytes"); nullptr for v1
    uint32_t wbits;
    uint32_t n_groups;
};
constexpr uint32_t kLongExt = 0x8000u;
// output bytes per group of the one-launch resolve: twice RESOLVE's -- the groups' chain is serial, fewer and larger links are
// faster (100 MB: 92 -> see DESIGN.md 4), and 96 KB of LDS still leaves a workgroup per CU waiting on every CU
constexpr uint32_t kLongGroupOut = 32768;
__host__ __device__ constexpr uint32_t long_resolve_lds() { return (kLongGroupOut + 16) + 2 * kLongGroupOut + 128 + kLongMaxLag * 8; }

template <int PHASE, bool EXT>  // PHASE: 1 = tail maps, 2 = finish; EXT: RLE / extended-match records and lags (extended format)
__global__ void __launch_bounds__(256) tamp_long_resolve_kernel(LongResolveArgs ra) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t nt = 256, BPT = 4;
    const uint32_t g = blockIdx.x;
    const LongGroup gr = ra.groups[g];
    const uint32_t n_out = gr.nout, ntok = gr.ntok;
    const uint32_t W = 1u << ra.wbits, mask = W - 1;
    const uint32_t* const rec = ra.recs + gr.tok0;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t capa = kLongGroupOut;
    uint8_t* const outb = smem;
    uint16_t* const src = reinterpret_cast<uint16_t*>(smem + capa + 16);
    typedef __attribute__((address_space(3))) volatile uint32_t LdsCtl;
    LdsCtl* const ctl = (LdsCtl*)(smem + capa + 16 + 2 * capa);
    // Extended format: tokens that wrote fewer bytes to the window than they produced (tamp_decompress_split_kernel.hpp: every byte
    // written to the window has a virtual position; output position = virtual position + the lag of the lagging tokens in front).
    // The group's list, padded to kLongMaxLag entries that no position reaches: both look-ups are six-step binary searches.
    uint32_t* const lagl = reinterpret_cast<uint32_t*>(smem + capa + 16 + 2 * capa + 128);
    const uint32_t nlag = EXT ? gr.nlag : 0u;
    if constexpr (EXT) {
        for (uint32_t i = tid; i < kLongMaxLag; i += nt) {
            lagl[2 * i] = i < nlag ? ra.lag[2 * (size_t)(gr.lag0 + i)] : 0xFFFFFFFFu;
            lagl[2 * i + 1] = i < nlag ? ra.lag[2 * (size_t)(gr.lag0 + i) + 1] : 0u;
        }
    }
    auto lag_before_out = [&](uint32_t O) -> uint32_t {  // lag of the lagging tokens that END at or before output position O
        uint32_t idx = 0;
#pragma unroll
        for (uint32_t st = kLongMaxLag / 2; st; st >>= 1)
            if ((lagl[2 * (idx + st - 1)] & 0xFFFFu) <= O) idx += st;
        return idx ? lagl[2 * idx - 1] : 0u;
    };
    auto out_of_virtual = [&](uint32_t v) -> uint32_t {  // output position of the byte with virtual position v
        uint32_t idx = 0;
#pragma unroll
        for (uint32_t st = kLongMaxLag / 2; st; st >>= 1)
            if ((lagl[2 * (idx + st - 1)] >> 16) <= v) idx += st;
        return v + (idx ? lagl[2 * idx - 1] : 0u);
    };
    if (n_out) {
        for (uint32_t i = tid; i < capa / 2; i += nt) reinterpret_cast<uint32_t*>(src)[i] = 0;
        __syncthreads();
        {   // token pass: every token leaves its number + 1 at the output position where it starts
            const uint32_t K = (ntok + nt - 1) / nt;
            const uint32_t j0 = min(tid * K, ntok), j1 = min(j0 + K, ntok);
            uint32_t sum = 0;
            for (uint32_t j = j0; j < j1; j++) sum += (rec[j] >> 2) & 0xFFu;
            const uint32_t incl = wave_scan_add(sum);
            if (lane == 63) ctl[wave] = incl;
            __syncthreads();
            uint32_t O = incl - sum;
            for (uint32_t w2 = 0; w2 < wave; w2++) O += ctl[w2];
            for (uint32_t j = j0; j < j1; j++) {
                const uint32_t olen = (rec[j] >> 2) & 0xFFu;
                if (olen) src[O] = (uint16_t)(j + 1);
                O += olen;
            }
        }
        __syncthreads();
        // byte pass: a final byte (literal), a pointer to an earlier byte of the group, or an external mark
        for (uint32_t r0 = 0, carry = 0; r0 < n_out; r0 += BPT * nt) {
            const uint32_t p0 = r0 + BPT * tid;
            uint2 lo = make_uint2(0, 0);
            if (p0 < capa) lo = *reinterpret_cast<const uint2*>(src + p0);
            const uint32_t h[2] = {lo.x, lo.y};
            uint32_t last = 0;
#pragma unroll
            for (uint32_t i = 0; i < BPT; i++)
                if ((h[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) last = p0 + i + 1;
            const uint32_t inc = wave_scan_max(last);
            if (lane == 63) ctl[8 + wave] = inc;
            __syncthreads();
            uint32_t head = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xF, 0xF, false);
            for (uint32_t w2 = 0; w2 < wave; w2++) head = max(head, (uint32_t)ctl[8 + w2]);
            const uint32_t carry_in = carry;
            head = max(head, carry);
            uint32_t wgmax = carry;
            for (uint32_t w2 = 0; w2 < 4; w2++) wgmax = max(wgmax, (uint32_t)ctl[8 + w2]);
            carry = wgmax;
            uint32_t jcur = 0;
            if (head) jcur = (head == carry_in && r0) ? (uint32_t)ctl[12] : (uint32_t)src[head - 1] - 1u;
            uint32_t hpos = head ? head - 1 : 0u;
            __syncthreads();  // every mark has been read: `src` may be overwritten with pointers now
            if (p0 < n_out) {
                uint32_t kind = 0, arg = 0, Vj = 0;
                const uint32_t pend = min(p0 + BPT, n_out);
                if (head) {
                    const uint32_t r = rec[jcur];
                    kind = r & 3u, arg = r >> 10, Vj = (EXT && nlag) ? hpos - lag_before_out(hpos) : hpos;
                }
#pragma unroll
                for (uint32_t i = 0; i < BPT; i++) {
                    const uint32_t p = p0 + i;
                    if (p >= pend) break;
                    const uint32_t m = (h[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                    if (m) {
                        jcur = m - 1, hpos = p;
                        const uint32_t r = rec[jcur];
                        kind = r & 3u, arg = r >> 10, Vj = (EXT && nlag) ? hpos - lag_before_out(hpos) : hpos;
                    }
                    const bool lit = kind == kRecLit;
                    // (rotated) ring index read; an RLE token repeats the byte in front of it (decompressor.c:136-138)
                    const uint32_t idx = (EXT && kind == kRecFill) ? ((Vj - 1) & mask) : ((arg + (p - hpos)) & mask);
                    const uint32_t back = (Vj - 1 - idx) & mask;      // 0 = newest ... W-1 = oldest
                    const bool ext = !lit && back >= Vj;              // not written by this group: the window in front of it
                    uint32_t v = Vj - 1 - back;                       // virtual position inside the group ...
                    if (EXT && nlag && !lit && !ext) v = out_of_virtual(v);  // ... and where that byte is in the output
                    src[p] = (uint16_t)(lit ? p : (ext ? (kLongExt | idx) : v));
                    outb[p] = (uint8_t)(lit ? arg : 0u);
                }
                if (tid == nt - 1) ctl[12] = jcur;
            }
            __syncthreads();
        }
        // pointer jumping; final = points at itself, external = bit 15 (both end a chain)
        constexpr uint32_t NQ = kLongGroupOut / (16 * nt);  // masks of 16 positions per thread
        uint32_t um[NQ] = {};
#pragma unroll
        for (uint32_t q = 0; q < NQ; q++) {
            if (q * 16 * nt < n_out) {
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t p = q * 16 * nt + i * nt + tid;
                    if (p < n_out) {
                        const uint32_t sp = src[p];
                        um[q] |= ((sp != p && !(sp & kLongExt)) ? 1u : 0u) << i;
                    }
                }
            }
        }
        for (uint32_t round = 0; round < 20; round++) {
#pragma unroll
            for (uint32_t q = 0; q < NQ; q++) {
                const uint32_t pq = q * 16 * nt + tid;
                for (uint32_t m = um[q]; m;) {
                    const uint32_t i = (uint32_t)__builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t p = pq + i * nt;
                    const uint32_t s1 = src[p];          // (a position: p is unresolved)
                    const uint32_t s2 = src[s1];
                    asm volatile("" ::: "memory");
                    if (s2 & kLongExt) {                 // s1 is external: so is p
                        src[p] = (uint16_t)s2;
                        um[q] &= ~(1u << i);
                    } else if (s2 == s1) {               // s1 is final
                        outb[p] = outb[s1];
                        asm volatile("" ::: "memory");
                        src[p] = (uint16_t)p;
                        um[q] &= ~(1u << i);
                    } else {
                        src[p] = (uint16_t)s2;           // one hop closer
                    }
                }
            }
            uint32_t any = 0;
#pragma unroll
            for (uint32_t q = 0; q < NQ; q++) any |= um[q];
            if (!__syncthreads_or((int)any)) break;
        }
    }
    if constexpr (PHASE == 1) {
        // the window as this group leaves it, oldest byte first: its own last bytes, and in front of them -- a group shorter
        // than W -- what is left of the window it found
        // (extended format: the window holds what was WRITTEN -- the group's last W virtual positions)
        uint16_t* const tm = ra.tailmap + (size_t)g * W;
        const uint32_t n_written = n_out - ((EXT && nlag) ? lagl[2 * nlag - 1] : 0u);
        for (uint32_t k = tid; k < W; k += nt) {
            uint32_t e;
            if (n_written + k >= W) {
                const uint32_t v = n_written + k - W;
                const uint32_t p = (EXT && nlag) ? out_of_virtual(v) : v;
                const uint32_t sp = src[p];
                e = (sp & kLongExt) ? sp : (uint32_t)outb[p];
            } else {
                e = kLongExt | (n_written + k);
            }
            tm[k] = (uint16_t)e;
        }
        return;
    } else {
        if (!n_out) return;
        // external bytes: the window in front of the group -- the output behind v0 - W (the scan launch stored every group's last
        // bytes), or, where the stream is younger than W, the fresh decoder's window at that ring index
        const unsigned long long v0 = gr.v0;
        const uint32_t wp0 = (uint32_t)(v0 & mask);
        uint8_t* const out = ra.out + v0;
        for (uint32_t p = tid; p < n_out; p += nt) {
            const uint32_t sp = src[p];
            if (sp & kLongExt) {
                const uint32_t j = sp & 0x7FFFu;  // j-th oldest byte of the window in front of the group
                uint32_t b;
                if (EXT) {
                    b = ra.groupwin[(size_t)g * W + j];
                } else if (v0 >= W) {
                    b = ra.out[v0 - W + j];
                } else {
                    const uint32_t r = (j + wp0) & mask;
                    b = r < (uint32_t)v0 ? ra.out[r] : ra.dict0[r];
                }
                outb[p] = (uint8_t)b;
            }
        }
        __syncthreads();
        const uint32_t head = min((uint32_t)((4 - (reinterpret_cast<uintptr_t>(out) & 3)) & 3), n_out);
        if (tid < head) out[tid] = outb[tid];
        const uint32_t ndw = (n_out - head) >> 2;
        uint32_t* const out32 = reinterpret_cast<uint32_t*>(out + head);
        for (uint32_t i = tid; i < ndw; i += nt) out32[i] = lds_u32_unaligned(outb, head + 4 * i);
        for (uint32_t i = head + 4 * ndw + tid; i < n_out; i += nt) out[i] = outb[i];
    }
}

// The tail maps composed.  window after group g = map g applied to the window after group g - 1, starting from the fresh
// decoder's window -- a chain over the groups, made short by blocks of kLongScanBlock groups (round 6; one workgroup walking all
// 3,050 groups of a 100 MB stream took 4.9 ms of the call's 9.4, 1.6 us a step):
//   MODE 0  a workgroup per block composes its groups' maps into ONE map of the block (the identity pushed through them: an
//           entry stays "byte j of the window in front of the block" or becomes a byte), all blocks at once;
//   MODE 1  ONE workgroup walks the block maps: the window in front of every block (n_blocks x W bytes);
//   MODE 2  a workgroup per block walks its groups again from that window and stores every group's last bytes where they
//           belong in the output: the finish launch reads its external bytes from there.
// A step is W gathers in LDS (two buffers), the next map prefetched into registers in front of the barrier.
constexpr uint32_t kLongScanThreads = 1024;
constexpr uint32_t kLongScanBlock = 64;
struct LongScanArgs {
    LongResolveArgs r;
    uint16_t* blockmap;   // n_blocks x W
    uint8_t* blockwin;    // n_blocks x W: the window in front of block b
    uint32_t n_blocks;
};
template <int MODE>
__global__ void __launch_bounds__(kLongScanThreads) tamp_long_tail_scan_kernel(LongScanArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const LongResolveArgs& ra = sa.r;
    const uint32_t W = 1u << ra.wbits;
    const uint32_t tid = threadIdx.x, nt = kLongScanThreads;
    constexpr uint32_t kPer = (1u << 15) / kLongScanThreads;  // map entries a thread holds at the largest window
    const uint32_t per = (W + nt - 1) / nt;                   // 1 .. kPer
    // two buffers of W entries: bytes (MODE 1, 2) or map entries (MODE 0)
    typedef typename std::conditional<MODE == 0, uint16_t, uint8_t>::type Ent;
    Ent* const buf0 = reinterpret_cast<Ent*>(smem);
    Ent* const buf1 = buf0 + W;
    // what is walked: the block maps (MODE 1: one workgroup, all of them) or this block's group maps
    const uint32_t first = MODE == 1 ? 0u : blockIdx.x * kLongScanBlock;
    const uint32_t count = MODE == 1 ? sa.n_blocks : min(kLongScanBlock, ra.n_groups - first);
    const uint16_t* const maps = MODE == 1 ? sa.blockmap : ra.tailmap + (size_t)first * W;
    __shared__ unsigned long long s_vend[kLongScanBlock];
    __shared__ uint32_t s_nout[kLongScanBlock];
    if constexpr (MODE == 2) {
        if (tid < count) {
            const LongGroup gr = ra.groups[first + tid];
            s_vend[tid] = gr.v0 + gr.nout, s_nout[tid] = gr.nout;
        }
    }
    for (uint32_t k = tid; k < W; k += nt) {
        if constexpr (MODE == 0) buf0[k] = (Ent)(kLongExt | k);  // the identity
        else if constexpr (MODE == 1) buf0[k] = ra.dict0[k];     // (group 0 starts at v0 = 0: byte j of its window is ring index j)
        else buf0[k] = sa.blockwin[(size_t)blockIdx.x * W + k];
    }
    uint16_t cur[kPer], nxt[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) {
        const uint32_t k = i * nt + tid;
        cur[i] = (i < per && k < W && count) ? maps[k] : 0;
        nxt[i] = 0;
    }
    __syncthreads();
    for (uint32_t g = 0; g < count; g++) {
        const Ent* const src = (g & 1) ? buf1 : buf0;
        Ent* const dst = (g & 1) ? buf0 : buf1;
        if (g + 1 < count) {
            const uint16_t* const tm = maps + (size_t)(g + 1) * W;
#pragma unroll
            for (uint32_t i = 0; i < kPer; i++) {
                const uint32_t k = i * nt + tid;
                if (i < per && k < W) nxt[i] = tm[k];
            }
        }
        if constexpr (MODE == 1) {  // the window in front of block g
            for (uint32_t k = tid; k < W; k += nt) sa.blockwin[(size_t)g * W + k] = src[k];
        }
        if constexpr (MODE == 2) {  // extended format: the window in front of group first + g, for the finish launch
            if (ra.groupwin)
                for (uint32_t k = tid; k < W; k += nt) ra.groupwin[(size_t)(first + g) * W + k] = src[k];
        }
#pragma unroll
        for (uint32_t i = 0; i < kPer; i++) {
            const uint32_t k = i * nt + tid;
            if (i < per && k < W) {
                const uint32_t e = cur[i];
                const Ent b = (e & kLongExt) ? src[e & 0x7FFFu] : (Ent)e;
                dst[k] = b;
                if constexpr (MODE == 2) {
                    // window byte k is output byte vend - W + k -- when the stream is that old, and when this group wrote any of
                    // it (an empty group leaves the bytes where the group in front stored them)
                    const unsigned long long vend = s_vend[g];
                    if (!ra.groupwin && s_nout[g] && vend + k >= W) ra.out[vend + k - W] = (uint8_t)b;
                }
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < kPer; i++) cur[i] = nxt[i];
        __syncthreads();
    }
    if constexpr (MODE == 0) {
        const Ent* const fin = (count & 1) ? buf1 : buf0;
        for (uint32_t k = tid; k < W; k += nt) sa.blockmap[(size_t)blockIdx.x * W + k] = fin[k];
    }
}

}  // namespace tamp_amd
