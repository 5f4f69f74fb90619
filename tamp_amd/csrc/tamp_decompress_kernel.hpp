// tamp_decompress_kernel.hpp -- batch `.tamp` decoder for gfx950: one lane per stream.
//
// Replaces, per stream, tamp_decompressor_init(conf=NULL) + tamp_decompressor_decompress
// (tamp/_c_src/tamp/decompressor.c:331-347,371-578): header parse (:276-297), bit refill (:357-365),
// prefix-code decode (:52-104), RLE / extended-match tokens (:114-273), the out-of-bounds rule
// (:232-236,540-544) and the window update (common.c:58-86).  Decoding is bit-serial inside a stream
// (SURVEY.md H5), so parallelism is across streams: 64 independent streams per wavefront, each lane
// running the reference's token loop on its own bit buffer.
//
// Template parameters pick one of three builds:
//   <true,  false>  windows in LDS rows, only the reference-shaped ("exact") token loop: batches of short messages;
//   <true,  true>   windows in LDS rows (<= 2^10 bytes) + the straight-line bulk path in front of the exact loop;
//   <false, true>   the same bulk path with the window in a per-lane slot of a global scratch slab (any window size,
//                   any number of resident waves);  <false, false> is the slab variant without the bulk path.
//
//   * Exact loop: the reference's token loop, byte-wise refill included (it decides status and consumed counts on
//     truncated input); a window slot that has not been written yet reads straight from the shared seed / custom
//     dictionary; input is prefetched a dword at a time, output leaves as aligned dwords.
//   * Bulk path: one token (or one 16-byte piece of a long one) per lane and step, straight-line code; 16-byte window
//     and output moves; input through a 64-byte LDS ring; global loads / stores only at I/O points common to the wave.
//     It ends at a token boundary whenever something needs the exact loop, which then resumes from the bit position.
#pragma once
#include "tamp_common.hpp"

namespace tamp_amd {

struct DecompressArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    const uint32_t* in_len;
    uint8_t* out;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    uint32_t* out_len;
    int8_t* status;
    uint32_t* in_consumed;      // may be null
    const uint8_t* dict;        // custom dictionary (>= 1<<window bytes) or null
    uint32_t dict_len;
    const uint8_t* seed_dicts;  // 3 tables of 1<<15 bytes: literal<=5, literal==6, literal>=7 (common.c:18-25)
    uint8_t* scratch;           // global variant: one window slot of (1 << max_wbits) bytes per resident lane
    const uint8_t* only_flagged;  // null, or one byte per stream: decode only the streams whose byte is set (what the
                                  // split decoder, tamp_decompress_split_kernel.hpp, left over)
    const uint32_t* flagged_count;  // with only_flagged: how many bytes are set (0: the kernel returns at once)
    uint32_t n_streams;
    uint32_t lds_row;           // LDS variant: bytes per lane row = (1 << max_wbits) + 4
    uint8_t max_wbits;
};

constexpr uint32_t kLdsWinBits = 10;  // largest window kept in LDS: 64 lanes x (2^10 + 36) B = 66 KiB per workgroup
// LDS variant, per 64-lane workgroup: [128 B prefix-code LUT][64 x 36 B output staging][64 window rows].  A row is
// (1 << max_wbits) + 36 bytes: 32 bytes of slack behind the window (16 that mirror its first bytes when a write runs
// over the end, 16 so that 16-byte reads near the end stay inside the row) + 4 so that the row stride in dwords is odd
// (lanes touching the same index hit different banks).
constexpr uint32_t kLaneRowPad = 36;
// per-lane staging in bulk builds: 80 B of output (4 pieces of 16 B + 15 carried bytes) + a 64 B ring of input; 148 B
// = 37 dwords, an odd stride
constexpr uint32_t kLaneStagePad = 148, kLaneInRing = 64, kLaneInOff = 84;
__host__ __device__ constexpr uint32_t lane_decoder_lds(uint32_t max_wbits) {
    return 128u + kWave * kLaneStagePad + kWave * ((1u << max_wbits) + kLaneRowPad);
}

struct B16 {
    uint32_t w[4];
};
// 16 bytes at any address (gfx950 runs with unaligned DS / global access enabled: one ds_read_b128 / global_load_dwordx4)
__device__ __forceinline__ B16 ld16(const uint8_t* p) {
    B16 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void st16(uint8_t* p, const B16& v) { __builtin_memcpy(p, &v, 16); }
// first `k` bytes (0..16) from `a`, the rest from `b`
__device__ __forceinline__ B16 blend16(const B16& a, const B16& b, uint32_t k) {
    B16 r;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t lo = 4u * (uint32_t)j;
        const uint32_t m = k >= lo + 4 ? 0xFFFFFFFFu : (k <= lo ? 0u : (1u << (8 * (k - lo))) - 1u);
        r.w[j] = (a.w[j] & m) | (b.w[j] & ~m);
    }
    return r;
}

// Prefix-code reader for the symbol that follows the 0 flag (decompressor.c:52-104).  `b` holds the
// upcoming bits left-aligned; returns the symbol and its code length, or -1 when `avail` is too small.
__device__ __forceinline__ int read_symbol(uint32_t b, uint32_t avail, uint32_t& used) {
    if (avail < 1) return -1;
    if ((b >> 31) == 0) {
        used = 1;
        return 0;
    }
    // code words (without the flag) are 2..8 bits; walk them from the packed tables
    const uint64_t codes_lo = 0x2b2624140b080300ull, codes_hi = 0x00ab27aa9594544bull, nbits = 0x979998877765532ull;
    int sym = -1;
    uint32_t nb = 0;
#pragma unroll
    for (int s = 1; s < 15; s++) {
        const uint32_t l = (uint32_t)((nbits >> (4 * s)) & 15) - 1u;
        const uint32_t code = (uint32_t)((s < 8 ? codes_lo >> (8 * s) : codes_hi >> (8 * (s - 8))) & 0xFF);
        if (sym < 0 && (b >> (32 - l)) == code) {
            sym = s;
            nb = l;
        }
    }
    if (avail < nb) return -1;
    used = nb;
    return sym;
}

// LDSWIN: window rows in LDS (else per-lane slots of a global scratch slab).  BULK: the straight-line
// bulk path with its LUT / output stage / row slack is compiled in; batches of short messages take the lean build,
// which fits more workgroups on a CU.
template <bool LDSWIN, bool BULK = false>
__global__ void __launch_bounds__(LDSWIN ? 64 : 256) tamp_decompress_kernel(DecompressArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    uint8_t* const lut = smem;                                         // bulk builds only
    uint8_t* const stg = smem + 128 + threadIdx.x * kLaneStagePad;      // bulk builds only
    // window: an LDS row per lane, or a slot of the global scratch slab (bulk builds: slot stride a.lds_row, padded)
    uint8_t* const win = LDSWIN ? smem + (BULK ? 128 + kWave * kLaneStagePad : 0u) + threadIdx.x * a.lds_row
                                : a.scratch + (BULK ? (size_t)gtid * a.lds_row : ((size_t)gtid << a.max_wbits));
    if constexpr (BULK) {
        // prefix-code LUT: index = the 7 bits after the leading 1 of a code word -> (extra bits << 4) | symbol
        // (decompressor.c:52-57 restated from the code table, compressor.c:33-36)
        for (uint32_t v = threadIdx.x; v < 128; v += blockDim.x) {
            const uint64_t codes_lo = 0x2b2624140b080300ull, codes_hi = 0x00ab27aa9594544bull, nbits = 0x979998877765532ull;
            uint32_t entry = 0;
            for (int s = 1; s < 15; s++) {
                const uint32_t l = (uint32_t)((nbits >> (4 * s)) & 15) - 1u;  // code length without the flag: 2..8
                const uint32_t code = (uint32_t)((s < 8 ? codes_lo >> (8 * s) : codes_hi >> (8 * (s - 8))) & 0xFF);
                if ((code & ((1u << (l - 1)) - 1)) == (v >> (7 - (l - 1)))) entry = ((l - 1) << 4) | (uint32_t)s;
            }
            lut[v] = (uint8_t)entry;
        }
        __syncthreads();
    }

    for (uint32_t s = gtid; s < a.n_streams; s += nthreads) {
        if (a.only_flagged && !a.only_flagged[s]) continue;
        const uint8_t* const in = a.in + a.in_off[s];
        const uint32_t n = a.in_len[s];
        uint8_t* const out = a.out + a.out_off[s];
        const uint32_t cap = a.out_cap[s];
        uint32_t ip = 0, op = 0;
        int res = kInputExhausted;

        // output staging: bytes collect in `oacc` and leave as one aligned dword
        uint32_t oacc = 0, on = 0;
        bool out_aligned = (reinterpret_cast<uintptr_t>(out) & 3) == 0;
        auto emit = [&](uint32_t b) {
            if (out_aligned) {
                oacc |= b << (8 * on);
                if (++on == 4) {
                    *reinterpret_cast<uint32_t*>(out + op - 3) = oacc;
                    oacc = 0;
                    on = 0;
                }
            } else {
                out[op] = (uint8_t)b;
            }
            op++;
        };
        auto emit_flush = [&]() {
            for (uint32_t k = 0; k < on; k++) out[op - on + k] = (uint8_t)(oacc >> (8 * k));
            on = 0;
            oacc = 0;
        };

        do {  // single pass; `break` = finished with `res`
            if (n > kMaxDecodeIn) { res = kBadArgument; break; }  // 32-bit bit counters (tamp_common.hpp)
            if (a.max_wbits < 8 || a.max_wbits > 15) { res = kInvalidConf; break; }  // decompressor.c:336
            // ---- header, decompressor.c:276-297 ----
            if (n == 0) break;
            const uint32_t h0 = in[0];
            const uint32_t hs = 1 + (h0 & 1);
            if (n < hs) { ip = 1; break; }  // first byte stashed, waiting for the second (decompressor.c:405-410)
            if (hs == 2 && in[1]) { res = kInvalidConf; break; }
            ip = hs;
            const uint32_t wbits = ((h0 >> 5) & 7) + 8, lbits = ((h0 >> 3) & 3) + 5;
            const bool custom = (h0 >> 2) & 1, extended = (h0 >> 1) & 1, dreset = h0 & 1;
            if (wbits > a.max_wbits) { res = kInvalidConf; break; }  // decompressor.c:311
            const uint32_t W = 1u << wbits, mask = W - 1;
            const uint32_t minp = (uint32_t)min_pattern_size((int)wbits, (int)lbits);
            const uint32_t table = (!extended || lbits >= 7) ? 2u : (lbits == 6 ? 1u : 0u);  // decompressor.c:318-319
            const uint8_t* const seed_default = a.seed_dicts + ((size_t)table << 15);
            const uint8_t* seed = seed_default;
            if (custom) {
                if (!a.dict || a.dict_len < W) { res = kInvalidConf; break; }
                seed = a.dict;
            }
            // window[i] is the private copy once it has been written, the shared dictionary before that
            uint32_t filled = 0;  // slots [0, filled) are private (saturates at W)
            auto wread = [&](uint32_t i) -> uint32_t { return i < filled ? win[i] : seed[i]; };
            uint32_t wp = 0;
            auto wwrite = [&](uint32_t b) {
                win[wp] = (uint8_t)b;
                wp = (wp + 1) & mask;
                if (filled < W) filled = wp ? (filled > wp ? filled : wp) : W;
            };

            // bit reader: `bb`/`nb` behave exactly like the reference's 32-bit buffer (decompressor.c:357-365);
            // bytes are fetched from HBM a dword at a time into `stage`
            uint32_t bb = 0, nb = 0;
            uint32_t stage = 0, ns = 0;  // ns prefetched bytes, next one in the low byte
            bool last_flush = false;
            auto refill = [&]() {
                while (ip < n && nb <= 24) {
                    if (ns == 0) {
                        const uint8_t* p = in + ip;
                        if ((reinterpret_cast<uintptr_t>(p) & 3) == 0 && ip + 4 <= n) {
                            stage = *reinterpret_cast<const uint32_t*>(p);
                            ns = 4;
                        } else {
                            stage = *p;
                            ns = 1;
                        }
                    }
                    nb += 8;
                    bb |= (stage & 0xFFu) << (32 - nb);
                    stage >>= 8;
                    ns--;
                    ip++;
                }
            };

            bool first_pass = true;
            for (;;) {  // LDS variant: bulk path and exact loop alternate; otherwise a single pass of the exact loop
            bool resume = false;
            uint32_t budget = 0xFFFFFFFFu;  // tokens the exact loop may take before the bulk path is tried again
            // short streams (telemetry messages) are over before the bulk path has paid for seeding its window row
            const bool use_bulk = BULK && n >= hs + 160;
            if constexpr (BULK) if (use_bulk) {
                // ================= bulk path: one token (or one 16-byte piece of a long one) per lane and iteration =================
                // Every lane keeps its own stream going in lock step with the other 63.  A step is straight-line code:
                // decode from a 64-bit bit window (input arrives 16 bytes at a time, prefetched 16 bytes ahead), read the
                // 16 source bytes with one unaligned LDS read, merge the first `wlen` of them into the window row with a
                // read-modify-write, append them to a 32-byte output stage that leaves as 16-byte stores.  Anything the
                // straight-line code does not cover -- the last ~48 input bytes, output about to fill up, FLUSH, an
                // out-of-bounds or self-overlapping extended match -- ends the bulk path for that lane at a token
                // boundary; the reference-exact loop below finishes the stream from the same bit position.
                if (first_pass) {
                    for (uint32_t k = 0; k < W; k += 16) st16(win + k, ld16(seed + k));  // window <- dictionary
                } else {
                    for (uint32_t k = filled; k < W; k++) win[k] = seed[k];  // after a dictionary reset in the exact loop
                    emit_flush();
                }
                first_pass = false;
                filled = W;
                uint32_t T = 8 * ip - nb;  // bits consumed from the start of the stream
                const uint32_t sp = T >> 3;  // stream offset of the byte holding the next bit
                bool fast = sp + 32 <= n && cap - op >= 16;
                // Input pipeline: compressed bytes pass through a 64-byte ring per lane in LDS.  A 16-byte chunk is fetched
                // from HBM into registers at one I/O point (every 4th step, the same step for all 64 lanes), stored to the
                // ring at the next one, and read from there a dword at a time (one dword prefetched).  Global loads and
                // stores are thus issued, and their results first touched, only at I/O points: stores share the
                // vector-memory counter with loads on gfx9, and a wait inside the step code would sit out every store.
                uint8_t* const inr = stg + kLaneInOff;
                const uint32_t sp0 = sp;  // stream byte x lives at inr[(x - sp0) & 63]
                auto ring_u32 = [&](uint32_t x) { return *reinterpret_cast<const uint32_t*>(inr + ((x - sp0) & (kLaneInRing - 1))); };
                B16 cb = {{0, 0, 0, 0}};  // chunk in flight
                bool cb_valid = false;
                uint32_t rp = sp;      // stream offset of the dword held in `wnext`
                uint32_t fill = sp;    // the ring holds stream bytes [.., fill)
                uint32_t ld_off = sp;  // next chunk to fetch
                uint32_t wnext = 0;
                uint64_t fb = 0;  // upcoming bits, left aligned
                uint32_t fn = 0;  // valid bits in fb
                if (fast) {
                    st16(inr, ld16(in + sp));
                    st16(inr + 16, ld16(in + sp + 16));
                    fill = ld_off = sp + 32;
                    fb = (uint64_t)__builtin_bswap32(ring_u32(sp)) << (32 + (T & 7));  // first dword, bits before T dropped
                    fn = 32 - (T & 7);
                    rp = sp + 4;
                    wnext = ring_u32(rp);
                    budget = 2;
                }
                const uint32_t T_in = T;
                uint32_t T_mark = T;  // bit position at the reference's most recent refill (it decides the consumed count)
                uint32_t on16 = 0, obase = op;  // bytes in the output stage; bytes already stored
                uint32_t pend = 0, p_w = 0, p_off = 0, p_kind = 0, p_byte = 0;  // rest of a token longer than 16 bytes
                auto refill32 = [&]() -> bool {
                    if (rp + 4 > fill) return false;  // the ring ran dry (a streak of long tokens, or the end of the input)
                    fb |= (uint64_t)__builtin_bswap32(wnext) << (32 - fn);
                    fn += 32;
                    rp += 4;
                    wnext = ring_u32(rp);  // (may be a stale slot: it is not used before `fill` has passed it)
                    return true;
                };
                auto flush_blocks = [&]() {  // whole 16-byte blocks of the stage -> HBM, the rest moves to the front
                    const uint32_t nblk = on16 >> 4;
                    for (uint32_t b = 0; __ballot(b < nblk); b++)
                        if (b < nblk) st16(out + obase + 16 * b, ld16(stg + 16 * b));
                    if (nblk) {
                        st16(stg, ld16(stg + 16 * nblk));
                        obase += 16 * nblk;
                        on16 &= 15;
                    }
                };
                enum { kLit = 0, kCopy = 1, kFill = 2 };
                while (__ballot(fast)) {
                    {  // ---- I/O point: every fourth step (the inner loop below), the same step for all 64 lanes ----
                        flush_blocks();
                        if (fast) {
                            if (cb_valid && fill + 16 - (rp - 4) <= kLaneInRing) {  // room: the oldest live dword is at rp - 4
                                st16(inr + ((fill - sp0) & (kLaneInRing - 1)), cb);
                                if (fill == rp) wnext = cb.w[0];  // the prefetched dword was read before this chunk arrived
                                fill += 16;
                                cb_valid = false;
                            }
                            if (!cb_valid && ld_off + 16 <= n) {
                                cb = ld16(in + ld_off);
                                ld_off += 16;
                                cb_valid = true;
                            }
                        }
                    }
                    // (four steps as an inner loop: with the I/O point tested in every step the compiler copied the
                    // loop-carried state twice per step, a third of the step's vector instructions)
#pragma unroll 1
                    for (uint32_t quad = 0; quad < 4; quad++) {
                    if (!fast) continue;
                    if (pend == 0) {  // ---- next token: peek, then commit or leave ----
                        const uint32_t T0 = T;
                        bool ok = fn >= 32 || refill32();
                        uint32_t used = 0, tok = 0, wl = 0;
                        uint32_t mark = T0;  // the reference refills at the top of every token (decompressor.c:357-365,431-445)
                        if (ok) {
                            if (fb >> 63) {  // literal, decompressor.c:466-482
                                p_byte = (uint32_t)((fb << 1) >> (64 - lbits));
                                used = 1 + lbits, tok = 1, wl = 1, p_kind = kLit;
                                ok = op < cap;
                            } else {
                                uint32_t sym = 0;
                                used = 2;
                                if ((fb >> 62) & 1) {
                                    const uint32_t e = lut[(uint32_t)(fb >> 55) & 0x7F];
                                    sym = e & 15, used = 2 + (e >> 4);
                                }
                                if (sym == kSymFlush) {
                                    ok = false;
                                } else if (!extended || sym < kSymRle) {  // plain match, decompressor.c:529-572
                                    tok = sym + minp;
                                    p_off = (uint32_t)((fb << used) >> (64 - wbits));
                                    used += wbits;
                                    wl = tok, p_kind = kCopy;
                                    ok = p_off + tok <= W && tok <= cap - op;
                                } else {  // RLE / extended match, decompressor.c:114-273
                                    fb <<= used, fn -= used, T += used;
                                    ok = fn >= 32 || refill32();
                                    if (ok) {
                                        const uint32_t trailing = sym == kSymRle ? 4u : 3u;
                                        uint32_t h = 0, u = 1;
                                        if (fb >> 63) {
                                            const uint32_t e = lut[(uint32_t)(fb >> 56) & 0x7F];
                                            h = e & 15, u = 1 + (e >> 4);
                                        }
                                        const uint32_t value = (h << trailing) + (uint32_t)((fb << u) >> (64 - trailing));
                                        u += trailing;
                                        if (sym == kSymRle) {
                                            tok = value + 2;
                                            wl = min(min(tok, kRleWindowMax), W - wp);
                                            p_kind = kFill, p_byte = win[(wp - 1) & mask];
                                            ok = tok <= cap - op;
                                        } else {
                                            tok = value + minp + 12;
                                            p_off = (uint32_t)((fb << u) >> (64 - wbits));
                                            // ... and once more in front of the offset if its buffer (25..32 bits after the
                                            // top-of-token refill) no longer holds `wbits` bits (decompressor.c:447-456)
                                            const uint32_t nb_top = 8 * (((T0 + 24) >> 3) + 1) - T0;
                                            if (nb_top - (T - T0) - u < wbits) mark = T + u;
                                            u += wbits;
                                            wl = min(tok, W - wp);
                                            p_kind = kCopy;
                                            // pieces are copied 16 bytes at a time: exact only if the window bytes written by an
                                            // earlier piece are not a later piece's source
                                            const bool overlap = p_off < wp + wl && wp < p_off + tok;
                                            ok = p_off + tok <= W && tok <= cap - op && !overlap;
                                        }
                                        used = u;
                                    }
                                }
                            }
                        }
                        if (!ok) {
                            T = T0;
                            fast = false;
                            continue;
                        }
                        fb <<= used, fn -= used, T += used;
                        pend = tok, p_w = wl;
                        T_mark = mark;
                    }
                    // ---- one piece of at most 16 bytes ----
                    const uint32_t olen = pend < 16 ? pend : 16u;
                    const uint32_t wlen = p_w < olen ? p_w : olen;
                    B16 src;
                    if (p_kind == kCopy) {
                        src = ld16(win + p_off);
                    } else {
                        const uint32_t r = p_kind == kFill ? p_byte * 0x01010101u : p_byte;
                        src.w[0] = r, src.w[1] = r, src.w[2] = r, src.w[3] = r;
                    }
                    if (wlen) {  // window <- first wlen bytes (memmove semantics: the source was read first, common.c:58-86)
                        st16(win + wp, blend16(src, ld16(win + wp), wlen));
                        if (wp + wlen > W) {  // ran over the end: the overhang sits in the mirror, bring it to the front
                            st16(win, blend16(ld16(win + W), ld16(win), wp + wlen - W));
                        }
                        wp = (wp + wlen) & mask;
                    }
                    st16(stg + on16, src);  // on16 <= 15 + 3 * 16 here: the stage is emptied at every I/O point
                    on16 += olen, op += olen;
                    pend -= olen, p_w -= wlen;
                    if (p_kind == kCopy) p_off += olen;
                    }
                }
                // ---- hand over to the exact loop: staged output out, bit reader rebuilt from the bit position ----
                flush_blocks();
                for (uint32_t k = 0; k < on16; k++) out[obase + k] = stg[k];
                if (T != T_in) {
                    // the reference's buffer at this token boundary: everything its last refill pulled in
                    last_flush = false;
                    const uint32_t ip_ref = min(n, ((T_mark + 24) >> 3) + 1);
                    bb = 0, nb = 0, stage = 0, ns = 0;
                    for (uint32_t b = T >> 3; b < ip_ref; b++) {
                        uint32_t byte = in[b], width = 8;
                        if (b == (T >> 3)) byte &= 0xFFu >> (T & 7), width = 8 - (T & 7);
                        bb |= byte << (32 - nb - width);
                        nb += width;
                    }
                    ip = ip_ref;
                }
                // the dword staging of the exact loop assumes it starts on a dword of the output
                out_aligned = ((reinterpret_cast<uintptr_t>(out) | op) & 3) == 0;
            }

            for (;;) {  // decompressor.c:431-575
                if (use_bulk && budget-- == 0) {  // back to the bulk path (it declines by itself near the end of the input)
                    resume = true;
                    break;
                }
                if (!(ip < n || nb)) break;
                if (op == cap) { res = kOutputFull; break; }
                refill();
                if (nb == 0) break;

                if (bb >> 31) {  // literal, decompressor.c:466-482
                    last_flush = false;
                    if (nb < 1 + lbits) break;
                    const uint32_t c = (bb << 1) >> (32 - lbits);
                    bb <<= 1 + lbits;
                    nb -= 1 + lbits;
                    emit(c);
                    wwrite(c);
                    continue;
                }

                uint32_t b2 = bb << 1, n2 = nb - 1, used = 0;
                const int sym = read_symbol(b2, n2, used);
                if (sym < 0) break;
                b2 <<= used;
                n2 -= used;

                if (sym == kSymFlush) {  // decompressor.c:501-514
                    bb = b2 << (n2 & 7);
                    nb = n2 & ~7u;
                    if (dreset && last_flush) {  // double FLUSH: back to the pristine seeded dictionary
                        wp = 0;
                        filled = 0;
                        seed = seed_default;
                    }
                    last_flush = true;
                    continue;
                }
                last_flush = false;

                if (extended && sym >= kSymRle) {
                    bb = b2;  // symbol bits are committed before the payload is read (decompressor.c:521-526)
                    nb = n2;
                    const uint32_t trailing = (sym == kSymRle) ? 4u : 3u;
                    uint32_t value = 0, match_len = 0, off = 0;
                    int got = 0;
                    bool starved = false;
                    for (;;) {  // decode_rle / decode_extended_match with the loop's refill-and-retry (:114-273,447-456)
                        if (got == 0) {
                            uint32_t u3 = 0;
                            int hsym = (nb >= 1 + trailing) ? read_symbol(bb, nb, u3) : -1;
                            if (hsym >= 0 && nb - u3 < trailing) hsym = -1;
                            if (hsym >= 0) {
                                uint32_t b3 = bb << u3;
                                value = ((uint32_t)hsym << trailing) + (b3 >> (32 - trailing));
                                bb = b3 << trailing;
                                nb -= u3 + trailing;
                                got = (sym == kSymRle) ? 2 : 1;
                                if (sym == kSymExt) match_len = value + minp + 12;
                            }
                        }
                        if (got == 1 && nb >= wbits) {
                            off = bb >> (32 - wbits);
                            bb <<= wbits;
                            nb -= wbits;
                            got = 2;
                        }
                        if (got == 2) break;
                        const uint32_t before = nb;
                        refill();
                        if (nb == before && ip == n) { starved = true; break; }
                    }
                    if (starved) break;
                    if (sym == kSymRle) {  // decompressor.c:140-173
                        const uint32_t count = value + 2;
                        const uint32_t c = wread((wp - 1) & mask);
                        const uint32_t room = cap - op;
                        const uint32_t w = count <= room ? count : room;
                        for (uint32_t k = 0; k < w; k++) emit(c);
                        const uint32_t ww = min(min(count, kRleWindowMax), W - wp);
                        for (uint32_t k = 0; k < ww; k++) wwrite(c);
                        if (w < count) { res = kOutputFull; break; }
                    } else {  // decompressor.c:229-272
                        if (off >= W || off + match_len > W) { res = kOob; break; }
                        const uint32_t room = cap - op;
                        const uint32_t w = match_len <= room ? match_len : room;
                        for (uint32_t k = 0; k < w; k++) emit(wread(off + k));
                        if (w < match_len) { res = kOutputFull; break; }
                        // window <- the same bytes up to the end of the buffer, no wrap, memmove semantics
                        // (tamp_window_copy, common.c:58-86): backwards when the destination runs into the source
                        const uint32_t ww = min(match_len, W - wp);
                        const uint32_t dist = (wp - off) & mask;
                        if (dist > 0 && dist < ww) {
                            for (uint32_t k = ww; k-- > 0;) win[wp + k] = (uint8_t)wread(off + k);
                            if (filled < W) filled = max(filled, wp + ww);
                            wp = (wp + ww) & mask;
                        } else {
                            for (uint32_t k = 0; k < ww; k++) wwrite(wread(off + k));
                        }
                    }
                    continue;
                }

                // plain match, decompressor.c:529-572
                if (n2 < wbits) break;
                const uint32_t match_len = (uint32_t)sym + minp;
                const uint32_t off = b2 >> (32 - wbits);
                if (off >= W || off + match_len > W) { res = kOob; break; }
                const uint32_t room = cap - op;
                if (match_len > room) {  // partial copy, token not consumed (decompressor.c:553-557)
                    for (uint32_t k = 0; k < room; k++) emit(wread(off + k));
                    res = kOutputFull;
                    break;
                }
                bb = b2 << wbits;
                nb = n2 - wbits;
                for (uint32_t k = 0; k < match_len; k++) emit(wread(off + k));
                {  // tamp_window_copy: destination wraps, memmove semantics
                    const uint32_t dist = (wp - off) & mask;
                    if (dist > 0 && dist < match_len) {
                        // reverse copy reads every source byte before it is overwritten; the private-copy
                        // watermark must already cover the whole destination for wread() to stay consistent
                        const uint32_t wp0 = wp;
                        uint32_t tmp[4] = {0, 0, 0, 0};
                        for (uint32_t k = 0; k < match_len; k++) tmp[k >> 2] |= wread(off + k) << (8 * (k & 3));
                        for (uint32_t k = 0; k < match_len; k++) wwrite((tmp[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                        (void)wp0;
                    } else {
                        for (uint32_t k = 0; k < match_len; k++) wwrite(wread(off + k));
                    }
                }
            }
            if (!resume) break;
            }  // bulk / exact alternation
        } while (false);

        emit_flush();
        a.out_len[s] = op;
        a.status[s] = (int8_t)res;
        if (a.in_consumed) a.in_consumed[s] = ip;
    }
}

}  // namespace tamp_amd
