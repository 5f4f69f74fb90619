// tamp_decompress_wave_kernel.hpp -- `.tamp` decoder, one WAVEFRONT per stream.
//
// Same contract as tamp_decompress_kernel.hpp (the reference's tamp_decompressor_init(conf=NULL) +
// tamp_decompressor_decompress, tamp/_c_src/tamp/decompressor.c:331-347,371-578); used when the batch is too
// small to fill the chip with one lane per stream, and generally faster for multi-KiB streams:
//   * the token loop runs on the scalar unit (bit buffer, prefix-code LUT, state) -- no divergence;
//   * the window lives in LDS (2^w bytes per wave, dictionary copied in with coalesced loads);
//   * back-references are copied by up to 64 lanes at once: read the source bytes, then write output staging and
//     window -- SIMT order gives tamp_window_copy's memmove semantics (common.c:58-86) for free;
//   * compressed input arrives 256 B per wave load, output leaves as coalesced 256 B stores from an LDS stage.
// Status / consumed counts follow the reference's byte-wise refill rule (decompressor.c:357-365): before each
// token it has pulled bytes until more than 24 bits are buffered, i.e. ip = min(n, floor((T + 24) / 8) + 1) for T
// bits consumed so far.
#pragma once
#include "tamp_common.hpp"
#include "tamp_decompress_kernel.hpp"

namespace tamp_amd {

constexpr uint32_t kStage = 512;  // output staging bytes per wave (ring); flushed 256 B at a time

__device__ __forceinline__ uint32_t uni32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// LDS per wave: window (1 << max_wbits) + stage.  Shared by the block: 128-byte prefix-code LUT.
__host__ __device__ inline uint32_t decode_wave_lds(uint32_t max_wbits, uint32_t waves) {
    return 128 + waves * ((1u << max_wbits) + kStage);
}

__global__ void __launch_bounds__(256) tamp_decompress_wave_kernel(DecompressArgs a) {
    if (a.only_flagged && a.flagged_count && *a.flagged_count == 0) return;  // (the split decoder left nothing over: the usual case)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const uint32_t wave = uni32(threadIdx.x >> 6);  // tell the compiler it is wave-uniform: the token loop goes scalar
    uint8_t* const lut = smem;  // index: 7 bits after the leading 1 of a code word -> (extra bits << 4) | symbol
    uint8_t* const win = smem + 128 + wave * ((1u << a.max_wbits) + kStage);
    uint8_t* const stage = win + (1u << a.max_wbits);

    // prefix-code LUT (decompressor.c:52-57 restated from the code table, compressor.c:33-36)
    for (uint32_t v = threadIdx.x; v < 128; v += blockDim.x) {  // v = the 7 bits following the leading 1
        const uint64_t codes_lo = 0x2b2624140b080300ull, codes_hi = 0x00ab27aa9594544bull, nbits = 0x979998877765532ull;
        uint32_t entry = 0;
        for (int s = 1; s < 15; s++) {
            const uint32_t l = (uint32_t)((nbits >> (4 * s)) & 15) - 1u;  // code length without the flag: 2..8
            const uint32_t code = (uint32_t)((s < 8 ? codes_lo >> (8 * s) : codes_hi >> (8 * (s - 8))) & 0xFF);
            // code = 1 followed by (l-1) bits; compare those with the top (l-1) bits of v
            if ((code & ((1u << (l - 1)) - 1)) == (v >> (7 - (l - 1)))) entry = ((l - 1) << 4) | (uint32_t)s;
        }
        lut[v] = (uint8_t)entry;
    }
    __syncthreads();

    const uint32_t gw = blockIdx.x * nwaves + wave, tw = gridDim.x * nwaves;
    for (uint32_t s = gw; s < a.n_streams; s += tw) {
        if (a.only_flagged && !a.only_flagged[s]) continue;  // (wave-uniform)
        const uint8_t* const in = a.in + a.in_off[s];
        const uint32_t n = a.in_len[s];
        uint8_t* const out = a.out + a.out_off[s];
        const uint32_t cap = a.out_cap[s];
        uint32_t op = 0;       // output bytes produced (staged or stored)
        uint32_t flushed = 0;  // output bytes already stored to HBM
        uint32_t ip_ref = 0;   // the reference's input cursor
        int res = kInputExhausted;

        // ---- input: 256-byte aligned chunks, one dword per lane ----
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3);
        const uint32_t* const in32 = reinterpret_cast<const uint32_t*>(in - mis);
        const uint32_t nwords = (mis + n + 3) >> 2;  // dwords covering the stream
        uint32_t chunk = 0xFFFFFFFFu, inreg = 0;
        uint64_t acc = 0;   // upcoming bits, left aligned
        uint32_t have = 0;  // valid bits in acc
        uint32_t wnext = 0; // next dword index (relative to in32) to append; bytes before `mis` are skipped
        uint32_t T = 0;     // bits consumed so far (from the start of the stream)
        auto fetch = [&]() {  // top up acc to >= 32 bits while input remains
            while (have <= 32 && wnext < nwords) {
                if ((wnext >> 6) != chunk) {
                    chunk = wnext >> 6;
                    const uint32_t idx = (chunk << 6) + lane;
                    inreg = __builtin_bswap32(idx < nwords ? in32[idx] : 0);  // first stream byte in the top bits
                }
                uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)inreg, (int)(wnext & 63));
                uint32_t nbv = 32;
                if (wnext == 0 && mis) {  // skip the bytes in front of the stream
                    w <<= 8 * mis;
                    nbv = 32 - 8 * mis;
                }
                acc |= (uint64_t)w << (32 - have);
                have += nbv;
                wnext++;
            }
        };
        auto take = [&](uint32_t k) {
            acc <<= k;
            have -= k;
            T += k;
        };
        uint32_t t_mark = 0;
        bool marked = false;
        auto mark_refill = [&]() {  // the reference refills here; its cursor is derived from T at the last such point
            t_mark = T;
            marked = true;
        };
        auto settle_mark = [&]() {
            if (marked) {
                const uint32_t r = ((t_mark + 24) >> 3) + 1;
                const uint32_t v = r < n ? r : n;
                ip_ref = v > ip_ref ? v : ip_ref;
                marked = false;
            }
        };
        const uint32_t total_bits = 8 * n;
        auto bits_left = [&]() { return total_bits - T; };

        // ---- output staging ----
        auto flush_stage = [&](bool all) {
            __builtin_amdgcn_wave_barrier();
            while (op - flushed >= 256 || (all && op > flushed)) {
                const uint32_t nbytes = min(op - flushed, 256u);
                uint8_t* dst = out + flushed;
                if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0 && nbytes == 256) {
                    reinterpret_cast<uint32_t*>(dst)[lane] =
                        *reinterpret_cast<const uint32_t*>(stage + ((flushed + 4 * lane) & (kStage - 1)));
                } else {
                    for (uint32_t k = lane; k < nbytes; k += 64) dst[k] = stage[(flushed + k) & (kStage - 1)];
                }
                flushed += nbytes;
            }
            __builtin_amdgcn_wave_barrier();
        };

        do {
            if (n > kMaxDecodeIn) { res = kBadArgument; break; }  // 32-bit bit counters (tamp_common.hpp)
            if (a.max_wbits < 8 || a.max_wbits > 15) { res = kInvalidConf; break; }
            if (n == 0) break;
            fetch();
            const uint32_t h0 = (uint32_t)(acc >> 56);
            const uint32_t hs = 1 + (h0 & 1);
            if (n < hs) { ip_ref = 1; break; }
            if (hs == 2 && ((acc >> 48) & 0xFF)) { res = kInvalidConf; break; }
            take(8 * hs);
            ip_ref = hs;
            const uint32_t wbits = ((h0 >> 5) & 7) + 8, lbits = ((h0 >> 3) & 3) + 5;
            const bool custom = (h0 >> 2) & 1, extended = (h0 >> 1) & 1, dreset = h0 & 1;
            if (wbits > a.max_wbits) { res = kInvalidConf; break; }
            const uint32_t W = 1u << wbits, mask = W - 1;
            const uint32_t minp = (uint32_t)min_pattern_size((int)wbits, (int)lbits);
            const uint32_t table = (!extended || lbits >= 7) ? 2u : (lbits == 6 ? 1u : 0u);
            const uint8_t* const seed_default = a.seed_dicts + ((size_t)table << 15);
            const uint8_t* seed = seed_default;
            if (custom) {
                if (!a.dict || a.dict_len < W) { res = kInvalidConf; break; }
                seed = a.dict;
            }
            auto load_window = [&](const uint8_t* src) {
                if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
                    for (uint32_t k = lane * 4; k < W; k += 256)
                        *reinterpret_cast<uint32_t*>(win + k) = *reinterpret_cast<const uint32_t*>(src + k);
                } else {
                    for (uint32_t k = lane; k < W; k += 64) win[k] = src[k];
                }
                __builtin_amdgcn_wave_barrier();
            };
            load_window(seed);
            uint32_t wp = 0;
            bool last_flush = false;

            for (;;) {
                if (bits_left() == 0) break;                     // nothing buffered, nothing left (decompressor.c:431)
                if (op == cap) { res = kOutputFull; break; }
                fetch();
                mark_refill();
                const uint32_t avail = bits_left();

                if (avail >= 32 && cap - op >= 16) {
                    // FAST PATH (same decisions, no availability checks): >= 32 real bits are buffered, which covers a
                    // literal (<= 9) or a plain match token (<= 9 + 15), and any plain match (<= 16 bytes) fits the output
                    const uint32_t top = (uint32_t)(acc >> 32);
                    if (top >> 31) {
                        last_flush = false;
                        const uint32_t c = (top << 1) >> (32 - lbits);
                        take(1 + lbits);
                        if (lane == 0) {
                            stage[op & (kStage - 1)] = (uint8_t)c;
                            win[wp] = (uint8_t)c;
                        }
                        op++;
                        wp = (wp + 1) & mask;
                        if (op - flushed >= 256) flush_stage(false);
                        continue;
                    }
                    const uint32_t t1 = top << 1;  // code word, left aligned
                    uint32_t fsym, fused;
                    if ((t1 >> 31) == 0) {
                        fsym = 0, fused = 2;
                    } else if ((t1 >> 30) == 3) {
                        fsym = 1, fused = 3;
                    } else {
                        const uint32_t e = uni32(lut[(t1 >> 24) & 0x7F]);
                        fsym = e & 15, fused = 2 + (e >> 4);
                    }
                    if (fsym < (extended ? (uint32_t)kSymRle : (uint32_t)kSymFlush)) {
                        last_flush = false;
                        const uint32_t match_len = fsym + minp;
                        const uint32_t off = (top << fused) >> (32 - wbits);
                        if (off >= W || off + match_len > W) { res = kOob; break; }
                        uint32_t b = 0;
                        if (lane < match_len) {
                            b = win[off + lane];
                            stage[(op + lane) & (kStage - 1)] = (uint8_t)b;
                        }
                        op += match_len;
                        take(fused + wbits);
                        if (lane < match_len) win[(wp + lane) & mask] = (uint8_t)b;  // sources all read above: memmove semantics
                        wp = (wp + match_len) & mask;
                        if (op - flushed >= 256) flush_stage(false);
                        continue;
                    }
                    // FLUSH / RLE / extended match: fall through to the general path
                }

                if (acc >> 63) {  // literal, decompressor.c:466-482
                    last_flush = false;
                    if (avail < 1 + lbits) break;
                    const uint32_t c = (uint32_t)((acc << 1) >> (64 - lbits));
                    take(1 + lbits);
                    if (lane == 0) {
                        stage[op & (kStage - 1)] = (uint8_t)c;
                        win[wp] = (uint8_t)c;
                    }
                    op++;
                    wp = (wp + 1) & mask;
                    if (op - flushed >= 256) flush_stage(false);
                    continue;
                }

                // token: prefix code after the 0 flag
                uint32_t used;  // bits of flag + symbol
                int sym;
                {
                    if (avail < 2) break;
                    if (((acc >> 62) & 1) == 0) {
                        sym = 0;
                        used = 2;
                    } else {
                        const uint32_t e = uni32(lut[(uint32_t)(acc >> 55) & 0x7F]);
                        sym = (int)(e & 15);
                        used = 2 + (e >> 4);
                        if (avail < used) break;
                    }
                }

                if (sym == kSymFlush) {  // decompressor.c:501-514: drop to the byte boundary
                    take(used);
                    take((8 - (T & 7)) & 7);
                    if (dreset && last_flush) {
                        wp = 0;
                        seed = seed_default;
                        load_window(seed);
                    }
                    last_flush = true;
                    continue;
                }
                last_flush = false;

                if (extended && sym >= kSymRle) {
                    take(used);  // the symbol is committed before its payload is read (decompressor.c:521-526)
                    const uint32_t trailing = (sym == kSymRle) ? 4u : 3u;
                    fetch();
                    // payload: prefix code (no flag) + trailing bits [+ window offset]
                    uint32_t pl;  // payload symbol bits
                    int hsym;
                    {
                        const uint32_t av = bits_left();
                        bool ok = av >= 1 + trailing;
                        if (ok && (acc >> 63) == 0) {
                            hsym = 0;
                            pl = 1;
                        } else if (ok) {
                            const uint32_t e = uni32(lut[(uint32_t)(acc >> 56) & 0x7F]);
                            hsym = (int)(e & 15);
                            pl = 1 + (e >> 4);
                        } else {
                            hsym = -1;
                            pl = 0;
                        }
                        if (hsym >= 0 && av < pl + trailing) hsym = -1;
                    }
                    if (hsym < 0) {  // starved: the reference refills once more and gives up (decompressor.c:447-456)
                        marked = false;
                        ip_ref = n;
                        break;
                    }
                    const uint32_t value = ((uint32_t)hsym << trailing) + (uint32_t)((acc << pl) >> (64 - trailing));
                    take(pl + trailing);
                    if (sym == kSymRle) {  // decompressor.c:140-173
                        const uint32_t count = value + 2;
                        const uint32_t c = uni32(win[(wp - 1) & mask]);
                        const uint32_t room = cap - op;
                        const uint32_t w = count <= room ? count : room;
                        for (uint32_t base = 0; base < w; base += 64) {
                            const uint32_t done = min(w - base, 64u);
                            if (lane < done) stage[(op + lane) & (kStage - 1)] = (uint8_t)c;
                            op += done;
                            if (op - flushed >= 256) flush_stage(false);
                        }
                        const uint32_t ww = min(min(count, kRleWindowMax), W - wp);
                        if (lane < ww) win[wp + lane] = (uint8_t)c;
                        wp = (wp + ww) & mask;
                        if (w < count) { res = kOutputFull; break; }
                    } else {  // extended match, decompressor.c:187-273
                        const uint32_t match_len = value + minp + 12;
                        fetch();
                        settle_mark();
                        if (8 * ip_ref - T < wbits) mark_refill();  // the reference refills only when its buffer runs short
                        if (bits_left() < wbits) { marked = false; ip_ref = n; break; }
                        const uint32_t off = (uint32_t)(acc >> (64 - wbits));
                        take(wbits);
                        if (off >= W || off + match_len > W) { res = kOob; break; }
                        const uint32_t room = cap - op;
                        const uint32_t w = match_len <= room ? match_len : room;
                        for (uint32_t base = 0; base < w; base += 64) {
                            const uint32_t done = min(w - base, 64u);
                            if (lane < done) stage[(op + lane) & (kStage - 1)] = win[off + base + lane];
                            op += done;
                            if (op - flushed >= 256) flush_stage(false);
                        }
                        if (w < match_len) { res = kOutputFull; break; }
                        // window <- the same bytes up to the end of the buffer, no wrap; memmove semantics: chunks in
                        // descending order when the destination runs into the source
                        const uint32_t ww = min(match_len, W - wp);
                        const uint32_t dist = (wp - off) & mask;
                        const bool reverse = dist > 0 && dist < ww;
                        const uint32_t nchunks = (ww + 63) >> 6;
                        for (uint32_t ci = 0; ci < nchunks; ci++) {
                            const uint32_t base = (reverse ? nchunks - 1 - ci : ci) << 6;
                            uint32_t b = 0;
                            if (base + lane < ww) b = win[off + base + lane];
                            __builtin_amdgcn_wave_barrier();
                            if (base + lane < ww) win[wp + base + lane] = (uint8_t)b;
                            __builtin_amdgcn_wave_barrier();
                        }
                        wp = (wp + ww) & mask;
                    }
                    continue;
                }

                // plain match, decompressor.c:529-572
                if (avail < used + wbits) break;
                const uint32_t match_len = (uint32_t)sym + minp;
                const uint32_t off = (uint32_t)((acc << used) >> (64 - wbits));
                if (off >= W || off + match_len > W) { res = kOob; break; }
                const uint32_t room = cap - op;
                const uint32_t w = match_len <= room ? match_len : room;
                uint32_t b = 0;
                if (lane < match_len) b = win[off + lane];
                if (lane < w) stage[(op + lane) & (kStage - 1)] = (uint8_t)b;
                op += w;
                if (w < match_len) { res = kOutputFull; break; }  // token not consumed (decompressor.c:553-557)
                take(used + wbits);
                if (lane < match_len) win[(wp + lane) & mask] = (uint8_t)b;  // all sources were read above: memmove semantics
                wp = (wp + match_len) & mask;
                if (op - flushed >= 256) flush_stage(false);
            }
        } while (false);

        settle_mark();
        flush_stage(true);
        if (lane == 0) {
            a.out_len[s] = op;
            a.status[s] = (int8_t)res;
            if (a.in_consumed) a.in_consumed[s] = ip_ref;
        }
    }
}

}  // namespace tamp_amd
