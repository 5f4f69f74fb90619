// tamp_decompress_kernel.hpp -- batch `.tamp` decoder for gfx950: one lane per stream.
//
// Replaces, per stream, tamp_decompressor_init(conf=NULL) + tamp_decompressor_decompress
// (tamp/_c_src/tamp/decompressor.c:331-347,371-578): header parse (:276-297), bit refill (:357-365),
// prefix-code decode (:52-104), RLE / extended-match tokens (:114-273), the out-of-bounds rule
// (:232-236,540-544) and the window update (common.c:58-86).  Decoding is bit-serial inside a stream
// (SURVEY.md H5), so parallelism is across streams: 64 independent streams per wavefront, each lane
// running the reference's token loop on its own bit buffer.
//
//   * Windows up to 2^10 bytes live in LDS (one padded 1028-byte row per lane, conflict-free when the lanes
//     touch the same index and spread over the banks otherwise): every back-reference byte is an LDS access
//     instead of a divergent global one.  Larger windows (or mixed batches up to 2^15) use a per-lane slot of
//     a global scratch slab that stays L2 / MALL resident.
//   * A window slot that has not been written yet reads straight from the shared seed (or custom) dictionary,
//     so no per-stream dictionary copy is made.
//   * Compressed input is prefetched a dword at a time and output is written as aligned dwords; the logical
//     byte-by-byte refill of the reference (which decides status and consumed counts on truncated input) is
//     reproduced on top of that.
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
    uint32_t n_streams;
    uint32_t lds_row;           // LDS variant: bytes per lane row = (1 << max_wbits) + 4
    uint8_t max_wbits;
};

constexpr uint32_t kLdsWinBits = 10;  // largest window kept in LDS: 64 lanes x (2^10 + 4) B = 64.25 KiB per workgroup
// padded row of (1 << max_wbits) + 4 bytes per lane: lane l, index i -> bank (l * (row/4) + i/4) % 64 = (l + i/4) % 64

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

template <bool LDSWIN>
__global__ void __launch_bounds__(LDSWIN ? 64 : 256) tamp_decompress_kernel(DecompressArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    uint8_t* const win = LDSWIN ? smem + threadIdx.x * a.lds_row : a.scratch + ((size_t)gtid << a.max_wbits);

    for (uint32_t s = gtid; s < a.n_streams; s += nthreads) {
        const uint8_t* const in = a.in + a.in_off[s];
        const uint32_t n = a.in_len[s];
        uint8_t* const out = a.out + a.out_off[s];
        const uint32_t cap = a.out_cap[s];
        uint32_t ip = 0, op = 0;
        int res = kInputExhausted;

        // output staging: bytes collect in `oacc` and leave as one aligned dword
        uint32_t oacc = 0, on = 0;
        const bool out_aligned = (reinterpret_cast<uintptr_t>(out) & 3) == 0;
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

            for (;;) {  // decompressor.c:431-575
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
        } while (false);

        emit_flush();
        a.out_len[s] = op;
        a.status[s] = (int8_t)res;
        if (a.in_consumed) a.in_consumed[s] = ip;
    }
}

}  // namespace tamp_amd
