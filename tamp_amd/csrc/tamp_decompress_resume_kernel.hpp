// tamp_decompress_resume_kernel.hpp -- resumable `.tamp` decoder: one call of tamp_decompressor_decompress
// (tamp/_c_src/tamp/decompressor.c:371-578) per decoder OBJECT, many objects per launch, one wavefront each.
//
// An object is what the reference keeps between calls (decompressor.h:13-57): 16 bytes of state -- bit buffer,
// window position, the token that was cut short by a full output buffer or by the end of the input -- next to its
// window buffer.  Here both live in HBM (TampAmdDecoderState + window, include/tamp_amd.h); a call loads the window
// into LDS, decodes whatever this call's input and output room allow, and stores window and state back, so that
// the sequence of (status, bytes written, bytes consumed) over any chunking of the stream equals the reference's.
//
// Machinery as in tamp_decompress_wave_kernel.hpp: scalar token loop, 64-lane copies, 256-byte input fetches and
// output stores.  The reference's 32-bit bit buffer is not simulated byte by byte: `T` counts the bits consumed since
// the start of this call's input (negative -- as a wrapped uint32 -- while the bits carried over from the previous
// call are being used), and the reference's input cursor follows from its refill rule (decompressor.c:357-365: before
// a token it has pulled bytes until more than 24 bits are buffered), i.e. floor((T + 24) / 8) + 1 at the last refill
// point.  Buffered bits at the end of the call = 8 * cursor - T; they go back into the state.
#pragma once
#include "tamp_common.hpp"
#include "tamp_decompress_kernel.hpp"
#include "tamp_decompress_wave_kernel.hpp"

namespace tamp_amd {

struct ResumeArgs {
    DecompressArgs d;       // per-call input / output tables; d.max_wbits = window capacity of every object
    uint8_t* states;        // object i at states + i * state_stride: 16-byte state, then (1 << max_wbits) window bytes
    uint64_t state_stride;
};

// decompressor.c:39-42
enum : uint32_t { kTokNone = 0, kTokRle = 1, kTokExtFresh = 2, kTokExtHaveSize = 3 };
// flags byte of the state
enum : uint32_t { kDsConfigured = 1, kDsHeaderStashed = 2, kDsLastWasFlush = 4 };

__global__ void __launch_bounds__(256) tamp_decompress_resume_kernel(ResumeArgs ra) {
    const DecompressArgs& a = ra.d;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const uint32_t wave = uni32(threadIdx.x >> 6);
    uint8_t* const lut = smem;
    uint8_t* const win = smem + 128 + wave * ((1u << a.max_wbits) + kStage);
    uint8_t* const stage = win + (1u << a.max_wbits);

    for (uint32_t v = threadIdx.x; v < 128; v += blockDim.x) {  // prefix-code LUT, as in the wave kernel
        const uint64_t codes_lo = 0x2b2624140b080300ull, codes_hi = 0x00ab27aa9594544bull, nbits = 0x979998877765532ull;
        uint32_t entry = 0;
        for (int s = 1; s < 15; s++) {
            const uint32_t l = (uint32_t)((nbits >> (4 * s)) & 15) - 1u;
            const uint32_t code = (uint32_t)((s < 8 ? codes_lo >> (8 * s) : codes_hi >> (8 * (s - 8))) & 0xFF);
            if ((code & ((1u << (l - 1)) - 1)) == (v >> (7 - (l - 1)))) entry = ((l - 1) << 4) | (uint32_t)s;
        }
        lut[v] = (uint8_t)entry;
    }
    __syncthreads();

    const uint32_t gw = blockIdx.x * nwaves + wave, tw = gridDim.x * nwaves;
    for (uint32_t s = gw; s < a.n_streams; s += tw) {
        uint8_t* const slot = ra.states + (uint64_t)s * ra.state_stride;
        uint8_t* const gwin = slot + 16;
        const uint32_t* const sw = reinterpret_cast<const uint32_t*>(slot);
        // state words: [bit_buffer][window_pos | bit_buffer_pos << 16 | token_state << 24]
        //              [pending_window_offset | pending_match_size << 16][conf | skip << 8 | flags << 16 | bits_max << 24]
        const uint32_t s0 = uni32(sw[0]), s1 = uni32(sw[1]), s2 = uni32(sw[2]), s3 = uni32(sw[3]);
        uint32_t wp = s1 & 0xFFFFu, ts = s1 >> 24;
        const uint32_t carried = (s1 >> 16) & 0xFFu;  // bits left over from the previous call (0..32)
        uint32_t pend_off = s2 & 0xFFFFu, pend_size = s2 >> 16;
        uint32_t conf = s3 & 0xFFu, skip = (s3 >> 8) & 0xFFu, flags = (s3 >> 16) & 0xFFu;
        const uint32_t bits_max = s3 >> 24;

        const uint8_t* const in = a.in + a.in_off[s];
        // (32-bit bit counters: one call looks at kMaxDecodeCall bytes at most and reports what it consumed)
        const uint32_t n = a.in_len[s] < kMaxDecodeCall ? a.in_len[s] : kMaxDecodeCall;
        uint8_t* const out = a.out + a.out_off[s];
        const uint32_t cap = a.out_cap[s];
        uint32_t op = 0, flushed = 0, ip_ref = 0;
        int res = kInputExhausted;

        // ---- input: 256-byte aligned chunks, one dword per lane; the carried bits go in front ----
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3);
        const uint32_t* const in32 = reinterpret_cast<const uint32_t*>(in - mis);
        const uint32_t nwords = n ? (mis + n + 3) >> 2 : 0;
        uint32_t chunk = 0xFFFFFFFFu, inreg = 0;
        uint64_t acc = carried ? (uint64_t)(s0 & (0xFFFFFFFFu << (32 - carried))) << 32 : 0;
        uint32_t have = carried;
        uint32_t wnext = 0;
        uint32_t T = 0u - carried;  // bits consumed, counted from the first bit of this call's input
        auto fetch = [&]() {
            while (have <= 32 && wnext < nwords) {
                if ((wnext >> 6) != chunk) {
                    chunk = wnext >> 6;
                    const uint32_t idx = (chunk << 6) + lane;
                    inreg = __builtin_bswap32(idx < nwords ? in32[idx] : 0);
                }
                uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)inreg, (int)(wnext & 63));
                uint32_t nbv = 32;
                if (wnext == 0 && mis) {
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
        auto mark_refill = [&]() {
            t_mark = T;
            marked = true;
        };
        auto settle_mark = [&]() {
            if (marked) {
                const int32_t r = (((int32_t)t_mark + 24) >> 3) + 1;  // floor division: T may be negative
                const uint32_t v = r <= 0 ? 0u : ((uint32_t)r < n ? (uint32_t)r : n);
                ip_ref = v > ip_ref ? v : ip_ref;
                marked = false;
            }
        };
        const uint32_t total_bits = 8 * n;
        auto bits_left = [&]() { return total_bits - T; };  // carried bits included (T starts below zero)

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
        auto put = [&](uint32_t b, uint32_t count) {  // lanes [0, count) append their byte b to the output (count <= 64)
            if (lane < count) stage[(op + lane) & (kStage - 1)] = (uint8_t)b;
            op += count;
            if (op - flushed >= 256) flush_stage(false);
        };

        bool window_live = false;  // LDS holds the object's window: store it back at the end
        uint32_t W = 0, mask = 0;
        auto load_window = [&](const uint8_t* src) {
            if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
                for (uint32_t k = lane * 4; k < W; k += 256)
                    *reinterpret_cast<uint32_t*>(win + k) = *reinterpret_cast<const uint32_t*>(src + k);
            } else {
                for (uint32_t k = lane; k < W; k += 64) win[k] = src[k];
            }
            __builtin_amdgcn_wave_barrier();
        };

        do {
            if (bits_max < 8 || bits_max > 15 || bits_max > a.max_wbits) { res = kInvalidConf; break; }
            // ---- header (decompressor.c:389-429), possibly split over two calls ----
            if (!(flags & kDsConfigured)) {
                uint32_t h0, hs;
                fetch();
                if (flags & kDsHeaderStashed) {
                    h0 = skip;  // the stashed first byte shares storage with skip_bytes (decompressor.h:47-50)
                    if (n == 0) break;
                    if ((uint32_t)(acc >> 56)) { res = kInvalidConf; break; }
                    hs = 1;  // bytes of THIS call's input that belong to the header
                } else {
                    if (n == 0) break;
                    h0 = (uint32_t)(acc >> 56);
                    if ((h0 & 1) && n < 2) {
                        skip = h0, flags |= kDsHeaderStashed;
                        ip_ref = 1;
                        break;
                    }
                    if ((h0 & 1) && ((acc >> 48) & 0xFF)) { res = kInvalidConf; break; }
                    hs = 1 + (h0 & 1);
                }
                take(8 * hs);
                ip_ref = hs;
                // tamp_decompressor_populate_from_conf, decompressor.c:304-329
                const uint32_t wb = ((h0 >> 5) & 7) + 8;
                if (wb > bits_max) { res = kInvalidConf; break; }
                conf = h0, flags = (flags | kDsConfigured) & ~kDsHeaderStashed, skip = 0;
                W = 1u << wb, mask = W - 1;
                const uint32_t lb = ((h0 >> 3) & 3) + 5;
                const uint32_t table = (!((h0 >> 1) & 1) || lb >= 7) ? 2u : (lb == 6 ? 1u : 0u);
                load_window(((h0 >> 2) & 1) ? gwin : a.seed_dicts + ((size_t)table << 15));  // custom: the caller's bytes
            } else {
                W = 1u << (((conf >> 5) & 7) + 8), mask = W - 1;
                load_window(gwin);
            }
            window_live = true;
            const uint32_t wbits = ((conf >> 5) & 7) + 8, lbits = ((conf >> 3) & 3) + 5;
            const bool extended = (conf >> 1) & 1, dreset = conf & 1;
            const uint32_t minp = (uint32_t)min_pattern_size((int)wbits, (int)lbits);
            const uint32_t table = (!extended || lbits >= 7) ? 2u : (lbits == 6 ? 1u : 0u);
            bool last_flush = (flags & kDsLastWasFlush) != 0;

            for (;;) {  // decompressor.c:431-575
                if (bits_left() == 0 && ts == kTokNone) break;
                if (op == cap) { res = kOutputFull; break; }
                fetch();
                mark_refill();
                const uint32_t avail = bits_left();
                bool dispatch = ts != kTokNone;

                if (!dispatch) {
                    if (acc >> 63) {  // literal, decompressor.c:466-482
                        last_flush = false;
                        if (avail < 1 + lbits) break;
                        const uint32_t c = (uint32_t)((acc << 1) >> (64 - lbits));
                        take(1 + lbits);
                        if (lane == 0) win[wp] = (uint8_t)c;
                        put(c, 1);
                        wp = (wp + 1) & mask;
                        continue;
                    }
                    uint32_t used;  // flag + symbol bits
                    int sym;
                    if (avail < 2) break;
                    if (((acc >> 62) & 1) == 0) {
                        sym = 0, used = 2;
                    } else {
                        const uint32_t e = uni32(lut[(uint32_t)(acc >> 55) & 0x7F]);
                        sym = (int)(e & 15), used = 2 + (e >> 4);
                        if (avail < used) break;
                    }
                    if (sym == kSymFlush) {  // decompressor.c:501-514
                        take(used);
                        take((8 - (T & 7)) & 7);
                        if (dreset && last_flush) {
                            wp = 0;
                            load_window(a.seed_dicts + ((size_t)table << 15));
                        }
                        last_flush = true;
                        continue;
                    }
                    last_flush = false;
                    if (extended && sym >= kSymRle) {  // symbol committed, then straight into the dispatch (:521-527)
                        take(used);
                        ts = sym == kSymRle ? kTokRle : kTokExtFresh;
                        fetch();
                        dispatch = true;
                    } else {  // plain match, decompressor.c:529-572
                        if (avail < used + wbits) break;
                        const uint32_t match_len = (uint32_t)sym + minp;
                        const uint32_t off = (uint32_t)((acc << used) >> (64 - wbits));
                        if (off >= W || off + match_len > W) { res = kOob; break; }
                        const uint32_t room = cap - op;
                        uint32_t w = match_len - skip;
                        const bool partial = w > room;  // the token stays in the bit buffer; the next call skips `skip`
                        if (partial) w = room;
                        uint32_t b = 0;
                        if (lane < match_len) b = win[off + lane];
                        if (lane >= skip && lane < skip + w) stage[(op + lane - skip) & (kStage - 1)] = (uint8_t)b;
                        op += w;
                        if (op - flushed >= 256) flush_stage(false);
                        if (partial) {
                            skip += w;
                        } else {
                            skip = 0;
                            take(used + wbits);
                            if (lane < match_len) win[(wp + lane) & mask] = (uint8_t)b;  // sources read above: memmove
                            wp = (wp + match_len) & mask;
                        }
                        continue;
                    }
                }

                // ---- RLE / extended match, fresh or picked up (decode_rle / decode_extended_match, :114-273) ----
                const bool rle = ts == kTokRle;
                const uint32_t trailing = rle ? 4u : 3u;
                uint32_t count, off = 0;
                if (skip) {
                    count = rle ? pend_off : pend_size;
                    off = pend_off;
                } else {
                    if (ts == kTokExtHaveSize) {
                        count = pend_size;
                    } else {
                        uint32_t pl;
                        int hsym;
                        const uint32_t av = bits_left();
                        const bool ok = av >= 1 + trailing;
                        if (ok && (acc >> 63) == 0) {
                            hsym = 0, pl = 1;
                        } else if (ok) {
                            const uint32_t e = uni32(lut[(uint32_t)(acc >> 56) & 0x7F]);
                            hsym = (int)(e & 15), pl = 1 + (e >> 4);
                        } else {
                            hsym = -1, pl = 0;
                        }
                        if (hsym >= 0 && av < pl + trailing) hsym = -1;
                        if (hsym < 0) {  // starved: everything was pulled and it is still not enough (:447-456)
                            marked = false;
                            ip_ref = n;
                            break;
                        }
                        count = ((uint32_t)hsym << trailing) + (uint32_t)((acc << pl) >> (64 - trailing));
                        take(pl + trailing);
                        count += rle ? 2u : minp + 12u;
                    }
                    if (!rle) {
                        fetch();
                        settle_mark();
                        if (8 * ip_ref - T < wbits) {  // the reference refills only when it runs short, and by then it
                            mark_refill();             // has parked the size (:215-222): an out-of-bounds offset
                            ts = kTokExtHaveSize, pend_size = count;  // leaves the object in that state
                        }
                        if (bits_left() < wbits) {  // size known, offset still to come (:215-222)
                            ts = kTokExtHaveSize, pend_size = count;
                            marked = false;
                            ip_ref = n;
                            break;
                        }
                        off = (uint32_t)(acc >> (64 - wbits));
                        take(wbits);
                    }
                }
                if (!rle && (off >= W || off + count > W)) { res = kOob; break; }
                const uint32_t remaining = count - skip, room = cap - op;
                const bool partial = remaining > room;
                const uint32_t w = partial ? room : remaining;
                const uint32_t skip0 = skip;
                if (partial) {
                    skip = skip0 + w;
                    ts = rle ? kTokRle : kTokExtHaveSize;
                    pend_off = rle ? count : off;
                    if (!rle) pend_size = count;
                } else {
                    skip = 0;
                    ts = kTokNone;
                }
                if (rle) {
                    const uint32_t c = uni32(win[(wp - 1) & mask]);
                    for (uint32_t base = 0; base < w; base += 64) put(c, min(w - base, 64u));
                    if (skip0 == 0) {  // window: first piece only, at most 8 bytes, no wrap (:160-170)
                        const uint32_t ww = min(min(count, kRleWindowMax), W - wp);
                        if (lane < ww) win[wp + lane] = (uint8_t)c;
                        wp = (wp + ww) & mask;
                    }
                } else {
                    for (uint32_t base = 0; base < w; base += 64) {
                        const uint32_t done = min(w - base, 64u);
                        put(lane < done ? win[off + skip0 + base + lane] : 0u, done);
                    }
                    if (!partial) {  // window only on the complete token, up to the end of the buffer (:262-270)
                        const uint32_t ww = min(count, W - wp);
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
                }
                if (partial) { res = kOutputFull; break; }
            }
            flags = last_flush ? (flags | kDsLastWasFlush) : (flags & ~kDsLastWasFlush);
        } while (false);

        settle_mark();
        flush_stage(true);
        // ---- state back: the bits the reference would still hold = 8 * cursor - T of them, starting at T ----
        fetch();
        const uint32_t nb = window_live ? 8 * ip_ref - T : carried;
        const uint32_t bb = window_live ? (nb ? (uint32_t)(acc >> 32) & (0xFFFFFFFFu << (32 - nb)) : 0u) : s0;
        if (window_live) {
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = lane * 4; k < W; k += 256)
                *reinterpret_cast<uint32_t*>(gwin + k) = *reinterpret_cast<const uint32_t*>(win + k);
        }
        if (lane == 0) {
            uint32_t* const dw = reinterpret_cast<uint32_t*>(slot);
            dw[0] = bb;
            dw[1] = (wp & 0xFFFFu) | (nb << 16) | (ts << 24);
            dw[2] = (pend_off & 0xFFFFu) | (pend_size << 16);
            dw[3] = (conf & 0xFFu) | ((skip & 0xFFu) << 8) | ((flags & 0xFFu) << 16) | (bits_max << 24);
            a.out_len[s] = op;
            a.status[s] = (int8_t)res;
            if (a.in_consumed) a.in_consumed[s] = ip_ref;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace tamp_amd
