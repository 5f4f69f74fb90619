// tamp_decompress_kernel.hpp -- batch `.tamp` decoder for gfx950: one lane per stream.
//
// Replaces, per stream, tamp_decompressor_init(conf=NULL) + tamp_decompressor_decompress
// (tamp/_c_src/tamp/decompressor.c:331-347,371-578): header parse (:276-297), bit refill (:357-365),
// prefix-code decode (:52-104), RLE / extended-match tokens (:114-273), the out-of-bounds rule
// (:232-236,540-544) and the window update (common.c:58-86).  Decoding is bit-serial inside a stream
// (SURVEY.md H5), so parallelism is across streams: 64 independent streams per wavefront, each lane
// running the reference's token loop on its own bit buffer.  Windows of mixed sizes (8..15 bits, read
// from each header) live in a per-lane slot of a global scratch slab that stays L2 / MALL resident.
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
    uint32_t* in_consumed;     // may be null
    const uint8_t* dict;       // custom dictionary (>= 1<<window bytes) or null
    uint32_t dict_len;
    const uint8_t* seed_dicts; // 3 tables of 1<<15 bytes: literal<=5, literal==6, literal>=7 (common.c:18-25)
    uint8_t* scratch;          // one window slot of (1 << max_wbits) bytes per resident lane
    uint32_t n_streams;
    uint8_t max_wbits;
};

// Prefix-code reader for the symbol that follows the 0 flag (decompressor.c:52-104).  `b` holds the
// upcoming bits left-aligned; returns the symbol and its code length, or -1 when `avail` is too small.
__device__ __forceinline__ int read_symbol(uint32_t b, uint32_t avail, uint32_t& used) {
    if (avail < 1) return -1;
    if ((b >> 31) == 0) {
        used = 1;
        return 0;
    }
    int sym = -1;
    uint32_t nb = 0;
#pragma unroll
    for (int s = 1; s < 15; s++) {
        const uint32_t l = d_nbits[s] - 1u;
        if (sym < 0 && (b >> (32 - l)) == d_code[s]) {
            sym = s;
            nb = l;
        }
    }
    if (avail < nb) return -1;
    used = nb;
    return sym;
}

__global__ void __launch_bounds__(256) tamp_decompress_kernel(DecompressArgs a) {
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    uint8_t* const win = a.scratch + ((size_t)gtid << a.max_wbits);

    for (uint32_t s = gtid; s < a.n_streams; s += nthreads) {
        const uint8_t* const in = a.in + a.in_off[s];
        const uint32_t n = a.in_len[s];
        uint8_t* const out = a.out + a.out_off[s];
        const uint32_t cap = a.out_cap[s];
        uint32_t ip = 0, op = 0;
        int res = kInputExhausted;

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
            const uint8_t* seed = a.seed_dicts + ((size_t)table << 15);
            if (custom) {
                if (!a.dict || a.dict_len < W) { res = kInvalidConf; break; }
                if ((reinterpret_cast<uintptr_t>(a.dict) & 3) == 0) {
                    for (uint32_t k = 0; k < W; k += 4)
                        *reinterpret_cast<uint32_t*>(win + k) = *reinterpret_cast<const uint32_t*>(a.dict + k);
                } else {
                    for (uint32_t k = 0; k < W; k++) win[k] = a.dict[k];
                }
            } else {
                for (uint32_t k = 0; k < W; k += 4)
                    *reinterpret_cast<uint32_t*>(win + k) = *reinterpret_cast<const uint32_t*>(seed + k);
            }

            uint32_t bb = 0, nb = 0, wp = 0;
            bool last_flush = false;
            auto refill = [&]() {  // decompressor.c:357-365
                while (ip < n && nb <= 24) {
                    nb += 8;
                    bb |= (uint32_t)in[ip++] << (32 - nb);
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
                    const uint8_t c = (uint8_t)((bb << 1) >> (32 - lbits));
                    bb <<= 1 + lbits;
                    nb -= 1 + lbits;
                    out[op++] = c;
                    win[wp] = c;
                    wp = (wp + 1) & mask;
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
                    if (dreset && last_flush) {
                        wp = 0;
                        for (uint32_t k = 0; k < W; k += 4)
                            *reinterpret_cast<uint32_t*>(win + k) = *reinterpret_cast<const uint32_t*>(seed + k);
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
                        const uint8_t c = win[(wp - 1) & mask];
                        const uint32_t room = cap - op;
                        const uint32_t w = count <= room ? count : room;
                        for (uint32_t k = 0; k < w; k++) out[op + k] = c;
                        op += w;
                        const uint32_t ww = min(min(count, kRleWindowMax), W - wp);
                        for (uint32_t k = 0; k < ww; k++) win[wp + k] = c;
                        wp = (wp + ww) & mask;
                        if (w < count) { res = kOutputFull; break; }
                    } else {  // decompressor.c:229-272
                        if (off >= W || off + match_len > W) { res = kOob; break; }
                        const uint32_t room = cap - op;
                        const uint32_t w = match_len <= room ? match_len : room;
                        for (uint32_t k = 0; k < w; k++) out[op + k] = win[off + k];
                        if (w < match_len) { op += w; res = kOutputFull; break; }
                        // window <- the same bytes, up to the end of the buffer, no wrap; sources are read from the
                        // output just written, which gives tamp_window_copy's memmove semantics (common.c:58-86)
                        const uint32_t ww = min(match_len, W - wp);
                        for (uint32_t k = 0; k < ww; k++) win[wp + k] = out[op + k];
                        wp = (wp + ww) & mask;
                        op += w;
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
                    for (uint32_t k = 0; k < room; k++) out[op + k] = win[off + k];
                    op += room;
                    res = kOutputFull;
                    break;
                }
                bb = b2 << wbits;
                nb = n2 - wbits;
                for (uint32_t k = 0; k < match_len; k++) out[op + k] = win[off + k];
                for (uint32_t k = 0; k < match_len; k++) win[(wp + k) & mask] = out[op + k];
                wp = (wp + match_len) & mask;
                op += match_len;
            }
        } while (false);

        a.out_len[s] = op;
        a.status[s] = (int8_t)res;
        if (a.in_consumed) a.in_consumed[s] = ip;
    }
}

}  // namespace tamp_amd
