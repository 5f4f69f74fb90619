// tamp_compress_resume_kernel.hpp -- compressor OBJECTS below flush granularity: tamp_compressor_poll /
// tamp_compressor_compress_cb / tamp_compressor_flush (tamp/_c_src/tamp/compressor.c:532-660,681-722,728-810) on
// state that survives between calls, many objects per launch, one wavefront each.
//
// The batch kernel (tamp_compress_kernel.hpp) encodes whole segments: everything between two flush points at once.
// What it cannot express is the reference's object in the middle of a segment -- a 16-byte input ring that is only
// parsed while full, an RLE run or an extended match still growing, a lazily cached match, up to 31 pending output
// bits, an output buffer that fills up.  This kernel keeps exactly that state (TampAmdEncoderState in
// include/tamp_amd.h = the fields of TampCompressor, compressor.h:13-66) next to the object's window in HBM and runs
// the reference's call on it: one parse step per ring fill, the reference's own search per step.
//
// Work split inside the wavefront: the state machine is wave-uniform (scalar unit); the two searches --
// find_best_match (compressor.c:113-172: every window index against the ring) and find_extended_match (:297-333) --
// spread the candidate indices over the 64 lanes and reduce "longest, then lowest index" with one wave max.  The window
// lives in LDS for the duration of the call.  Throughput per object is that of a serial parser; the point of this
// kernel is exactness for callers that feed small pieces (many objects advance together in one launch), not speed --
// whole segments belong to the batch kernel.
#pragma once
#include "tamp_common.hpp"
#include "tamp_compress_kernel.hpp"
#include "tamp_decompress_wave_kernel.hpp"

namespace tamp_amd {

enum : uint32_t { kEncPoll = 1, kEncCompress = 2, kEncFlush = 3, kEncCompressAndFlush = 4 };
// flags byte of TampAmdEncoderState
enum : uint32_t { kEsCustom = 1, kEsExtended = 2, kEsDictReset = 4, kEsAppend = 8, kEsLazy = 16 };

struct EncodeResumeArgs {
    uint8_t* states;  // object i at states + i * state_stride: 40-byte state, then (1 << window_bits_max) window bytes
    uint64_t state_stride;
    const uint8_t* in;
    const uint64_t* in_off;
    const uint32_t* in_len;
    uint8_t* out;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    uint32_t* out_len;
    int8_t* status;
    uint32_t* in_consumed;  // may be null
    uint32_t n_objects;
    uint32_t op;            // kEncPoll ...
    uint32_t write_token;   // flush ops
    uint8_t max_wbits;      // window capacity of every object
};

__host__ __device__ inline uint32_t encode_resume_lds(uint32_t max_wbits, uint32_t waves) {
    return waves * ((1u << max_wbits) + 32);  // window + ring (16 B) + slack
}

__global__ void __launch_bounds__(256) tamp_compress_resume_kernel(EncodeResumeArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const uint32_t wave = uni32(threadIdx.x >> 6);
    uint8_t* const win = smem + wave * ((1u << a.max_wbits) + 32);
    uint8_t* const ring = win + (1u << a.max_wbits);

    const uint32_t gw = blockIdx.x * nwaves + wave, tw = gridDim.x * nwaves;
    for (uint32_t s = gw; s < a.n_objects; s += tw) {
        uint8_t* const slot = a.states + (uint64_t)s * a.state_stride;
        uint8_t* const gwin = slot + 40;
        uint32_t* const sw = reinterpret_cast<uint32_t*>(slot);
        // ---- state in (layout: include/tamp_amd.h TampAmdEncoderState) ----
        uint32_t bb = uni32(sw[0]);
        const uint32_t s1 = uni32(sw[1]), s2 = uni32(sw[2]), s7 = uni32(sw[7]), s8 = uni32(sw[8]);
        uint32_t wp = s1 & 0xFFFFu, nb = (s1 >> 16) & 0xFFu, in_size = s1 >> 24;
        uint32_t in_pos = s2 & 0xFFu;
        const uint32_t wbits = (s2 >> 8) & 0xFFu, lbits = (s2 >> 16) & 0xFFu, flags = s2 >> 24;
        int32_t cidx = (int32_t)(int16_t)(s7 & 0xFFFFu);  // lazy cache, -1 = none
        uint32_t extp = s7 >> 16;
        uint32_t csize = s8 & 0xFFu, rle = (s8 >> 8) & 0xFFu, extc = (s8 >> 16) & 0xFFu, lwf = s8 >> 24;

        const uint8_t* const in = a.in + a.in_off[s];
        uint32_t in_left = a.in_len[s];
        uint8_t* const out = a.out + a.out_off[s];
        const uint32_t cap = a.out_cap[s];
        uint32_t op = 0, ip = 0;
        int res = kOk;

        const bool conf_ok = wbits >= 8 && wbits <= 15 && wbits <= a.max_wbits && lbits >= 5 && lbits <= 8;
        if (!conf_ok) {
            if (lane == 0) {
                a.out_len[s] = 0;
                a.status[s] = (int8_t)kInvalidConf;
                if (a.in_consumed) a.in_consumed[s] = 0;
            }
            continue;
        }
        const uint32_t W = 1u << wbits, mask = W - 1;
        const bool extended = flags & kEsExtended, lazy = flags & kEsLazy, dreset = flags & kEsDictReset;
        const uint32_t minp = (uint32_t)min_pattern_size((int)wbits, (int)lbits);
        const uint32_t maxp = extended ? minp + 11 + kExtExtraMax : minp + 13;  // compressor.c:12-19

        for (uint32_t k = lane * 4; k < W; k += 256)
            *reinterpret_cast<uint32_t*>(win + k) = *reinterpret_cast<const uint32_t*>(gwin + k);
        if (lane < 4) reinterpret_cast<uint32_t*>(ring)[lane] = sw[3 + lane];
        __builtin_amdgcn_wave_barrier();

        auto rd_in = [&](uint32_t k) __attribute__((always_inline)) -> uint32_t { return uni32(ring[(in_pos + k) & 15]); };  // read_input
        auto put = [&](uint32_t bits, uint32_t n) __attribute__((always_inline)) {  // write_to_bit_buffer, compressor.c:49-52
            nb += n;
            bb |= bits << (32 - nb);
        };
        auto partial_flush = [&]() __attribute__((always_inline)) -> int {  // compressor.c:65-75
            while (nb >= 8 && op < cap) {
                if (lane == 0) out[op] = (uint8_t)(bb >> 24);
                op++;
                nb -= 8;
                bb <<= 8;
            }
            return nb >= 8 ? kOutputFull : kOk;
        };
        auto last_byte = [&]() __attribute__((always_inline)) -> uint32_t { return uni32(win[(wp - 1) & mask]); };
        auto win_write = [&](uint32_t b) __attribute__((always_inline)) {  // one byte at the cursor, wrapping
            if (lane == 0) win[wp] = (uint8_t)b;
            wp = (wp + 1) & mask;
        };
        auto put_exthuff = [&](uint32_t value, uint32_t trailing) __attribute__((always_inline)) {  // compressor.c:257-263
            const uint32_t ci = value >> trailing;
            put((tok_code(ci) << trailing) | (value & ((1u << trailing) - 1)), (tok_nbits(ci) - 1) + trailing);
        };

        // find_best_match (compressor.c:113-172) for the ring from offset `o`, `R` bytes of it
        auto find_best = [&](uint32_t o, uint32_t R, uint32_t& idx, uint32_t& len) __attribute__((always_inline)) {
            idx = 0, len = 0;
            if (R < minp) return;
            const uint32_t cmax = min(R, maxp);
            // the (up to 16) pattern bytes as four dwords (always indexed by constants: no scratch array)
            uint32_t pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                pw0 |= (uint32_t)ring[(in_pos + o + k) & 15] << (8 * k);
                pw1 |= (uint32_t)ring[(in_pos + o + 4 + k) & 15] << (8 * k);
                pw2 |= (uint32_t)ring[(in_pos + o + 8 + k) & 15] << (8 * k);
                pw3 |= (uint32_t)ring[(in_pos + o + 12 + k) & 15] << (8 * k);
            }
            const uint32_t p01 = pw0 & 0xFFFFu;
            uint32_t key = 0;
            // 16 candidates per lane and pass: the two-byte filter is 16 independent LDS reads (one round trip); a
            // survivor's length comes from four more, compared a dword at a time (reads may run up to 19 bytes past
            // the window into the ring / slack behind it: cut off by the W - c cap)
            for (uint32_t c0 = lane; c0 + 1 < W; c0 += 16 * kWave) {
                uint32_t hits = 0;
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) {
                    const uint32_t cc = c0 + k * kWave;
                    const bool valid = cc + 1 < W;
                    const uint32_t v = lds_u32_unaligned(win, valid ? cc : 0);
                    hits |= (uint32_t)(valid && (v & 0xFFFFu) == p01) << k;
                }
                while (hits) {
                    const uint32_t k = (uint32_t)__builtin_ctz(hits);
                    hits &= hits - 1;
                    const uint32_t cc = c0 + k * kWave;
                    const uint32_t x0 = lds_u32_unaligned(win, cc) ^ pw0, x1 = lds_u32_unaligned(win, cc + 4) ^ pw1;
                    const uint32_t x2 = lds_u32_unaligned(win, cc + 8) ^ pw2, x3 = lds_u32_unaligned(win, cc + 12) ^ pw3;
                    uint32_t l = 16;
                    if (x3) l = 12 + ((uint32_t)__builtin_ctz(x3) >> 3);
                    if (x2) l = 8 + ((uint32_t)__builtin_ctz(x2) >> 3);
                    if (x1) l = 4 + ((uint32_t)__builtin_ctz(x1) >> 3);
                    if (x0) l = (uint32_t)__builtin_ctz(x0) >> 3;
                    l = min(l, min(cmax, W - cc));
                    key = max(key, (l << 16) | (0xFFFFu - cc));  // longest; ties -> lowest index
                }
            }
            key = wave_max_u32(key);
            if (key) {
                len = key >> 16;
                idx = 0xFFFFu - (key & 0xFFFFu);
            }
        };
        // find_extended_match (compressor.c:297-333)
        auto find_ext = [&](uint32_t pos, uint32_t cnt, uint32_t& npos, uint32_t& ncnt) __attribute__((always_inline)) {
            const uint32_t mp = min(cnt + in_size, maxp);
            const uint32_t nextb = rd_in(0);
            uint32_t key = 0;
            for (uint32_t c = pos + lane; c + cnt + 1 <= W; c += kWave) {
                if (win[c + cnt] != nextb) continue;
                uint32_t i = 0;
                while (i < cnt && win[c + i] == win[pos + i]) i++;
                if (i < cnt) continue;
                const uint32_t cmax = min(mp, W - c);
                uint32_t l = cnt + 1;
                while (l < cmax && win[c + l] == ring[(in_pos + l - cnt) & 15]) l++;
                key = max(key, (l << 16) | (0xFFFFu - c));
            }
            key = wave_max_u32(key);
            ncnt = key >> 16;
            npos = key ? 0xFFFFu - (key & 0xFFFFu) : pos;
        };
        auto write_rle = [&](uint32_t count) __attribute__((always_inline)) {  // write_rle_token, compressor.c:342-359
            const uint32_t sym = last_byte();
            put(tok_code(kSymRle), tok_nbits(kSymRle));
            put_exthuff(count - 2, 4);
            const uint32_t ww = min(min(count, kRleWindowMax), W - wp);
            if (lane < ww) win[wp + lane] = (uint8_t)sym;
            __builtin_amdgcn_wave_barrier();
            wp = (wp + ww) & mask;
        };
        auto write_ext = [&]() __attribute__((always_inline)) -> int {  // write_extended_match_token, compressor.c:377-415
            if (cap - op < 6) return kOutputFull;
            const uint32_t count = extc, pos = extp;
            put(tok_code(kSymExt), tok_nbits(kSymExt));
            put_exthuff(count - minp - 12, 3);
            int r = partial_flush();
            if (r != kOk) return r;
            put(pos, wbits);
            r = partial_flush();
            if (r != kOk) return r;
            const uint32_t ww = min(count, W - wp);  // to the end of the buffer, no wrap; memmove semantics
            const uint32_t dist = (wp - pos) & mask;
            const bool reverse = dist > 0 && dist < ww;
            const uint32_t nchunks = (ww + 63) >> 6;
            for (uint32_t ci = 0; ci < nchunks; ci++) {
                const uint32_t base = (reverse ? nchunks - 1 - ci : ci) << 6;
                uint32_t b = 0;
                if (base + lane < ww) b = win[pos + base + lane];
                __builtin_amdgcn_wave_barrier();
                if (base + lane < ww) win[wp + base + lane] = (uint8_t)b;
                __builtin_amdgcn_wave_barrier();
            }
            wp = (wp + ww) & mask;
            extc = 0;
            return kOk;
        };
        auto consume = [&](uint32_t k) __attribute__((always_inline)) {
            in_pos = (in_pos + k) & 15;
            in_size -= k;
        };

        constexpr int kPollContinue = 127;
        // poll_extended_handling, compressor.c:437-525
        auto poll_ext = [&](uint32_t& midx, uint32_t& msize) __attribute__((always_inline)) -> int {
            if (extc) {
                const uint32_t max_ext = minp + 11 + kExtExtraMax;
                while (in_size > 0) {
                    if (extp + extc >= W || extc >= max_ext) return write_ext();
                    uint32_t npos, ncnt;
                    find_ext(extp, extc, npos, ncnt);
                    if (ncnt > extc) {
                        const uint32_t extra = ncnt - extc;
                        extp = npos, extc = ncnt;
                        consume(extra);
                        continue;
                    }
                    return write_ext();
                }
                return kOk;
            }
            const uint32_t lastb = last_byte();
            uint32_t avail = 0;
            while (avail < in_size && rle + avail < kRleMax && rd_in(avail) == lastb) avail++;
            const uint32_t total = rle + avail;
            const bool ended = avail < in_size || total >= kRleMax;
            if (!ended && total > 0) {
                rle = total;
                consume(avail);
                return kOk;
            }
            if (total >= 2) {
                if (total == avail && total <= 6) {
                    uint32_t pidx, psize;
                    find_best(0, in_size, pidx, psize);
                    if (psize > total) {
                        rle = 0;
                        midx = pidx, msize = psize;
                        return kPollContinue;
                    }
                }
                consume(avail);
                write_rle(total);
                rle = 0;
                return kOk;
            }
            if (rle == 1) {  // a lone run byte swallowed by an earlier poll: back out as a literal (:512-523)
                put((1u << lbits) | lastb, lbits + 1);
                win_write(lastb);
                rle = 0;
                return kOk;
            }
            return kPollContinue;
        };
        // tamp_compressor_poll, compressor.c:532-660
        auto poll = [&]() __attribute__((always_inline)) -> int {
            if (in_size == 0) return kOk;
            lwf = 0;
            int r = partial_flush();
            if (r != kOk) return r;
            if (op == cap) return kOutputFull;
            uint32_t msize = 0, midx = 0;
            if (extended) {
                r = poll_ext(midx, msize);
                if (r != kPollContinue) {
                    cidx = -1;
                    return r;
                }
            }
            if (lazy) {  // compressor.c:576-619
                if (cidx >= 0) {
                    midx = (uint32_t)cidx, msize = csize;
                    cidx = -1;
                } else if (msize == 0) {
                    find_best(0, in_size, midx, msize);
                }
                if (msize >= minp && msize <= 8 && in_size > msize + 2) {
                    uint32_t nidx, nsize;
                    find_best(1, in_size - 1, nidx, nsize);
                    // the literal's slot must not lie inside the match found for the next position (:185-188)
                    if (nsize > msize && (wp < nidx || wp >= nidx + nsize)) {
                        cidx = (int32_t)nidx, csize = nsize;
                        msize = 0;
                    } else {
                        cidx = -1;
                    }
                } else {
                    cidx = -1;
                }
            } else if (msize == 0) {
                find_best(0, in_size, midx, msize);
            }
            if (msize < minp) {
                msize = 1;
                const uint32_t c = rd_in(0);
                if (c >> lbits) return kExcessBits;
                put((1u << lbits) | c, lbits + 1);
            } else {
                if (extended && msize > minp + 11) {  // start of an extended match: nothing is written yet (:636-644)
                    extc = msize, extp = midx;
                    consume(msize);
                    return kOk;
                }
                put((tok_code(msize - minp) << wbits) | midx, tok_nbits(msize - minp) + wbits);
            }
            for (uint32_t i = 0; i < msize; i++) {  // the consumed bytes enter the window, wrapping (:651-657)
                win_write(rd_in(0));
                in_pos = (in_pos + 1) & 15;
            }
            __builtin_amdgcn_wave_barrier();
            in_size -= msize;
            return kOk;
        };
        // tamp_compressor_sink + the loop of tamp_compressor_compress_cb, compressor.c:665-722
        auto compress = [&]() __attribute__((always_inline)) -> int {
            while (in_left > 0 && op < cap) {
                const uint32_t take = min(16u - in_size, in_left);
                if (lane < take) ring[(in_pos + in_size + lane) & 15] = in[ip + lane];
                __builtin_amdgcn_wave_barrier();
                in_size += take, ip += take, in_left -= take;
                if (in_size == 16) {
                    const int r = poll();
                    if (r != kOk) return r;
                }
            }
            return kOk;
        };
        // tamp_compressor_flush, compressor.c:728-810
        auto flush = [&](bool write_token) __attribute__((always_inline)) -> int {
            for (;;) {
                int r = partial_flush();
                if (r != kOk) return r;
                if (in_size) {
                    r = poll();
                } else if (extended && rle >= 1) {
                    if (rle == 1) {
                        const uint32_t c = last_byte();
                        put((1u << lbits) | c, lbits + 1);
                        win_write(c);
                    } else {
                        write_rle(rle);
                    }
                    rle = 0;
                } else if (extended && extc) {
                    r = write_ext();
                } else {
                    break;
                }
                if (r != kOk) return r;
            }
            if (write_token && !lwf && (nb || dreset)) {
                if (cap - op < 2) return kOutputFull;
                put(0xABu, 9);
                lwf = 1;
            }
            const int r = partial_flush();
            if (nb) {
                if (op == cap) return kOutputFull;
                if (lane == 0) out[op] = (uint8_t)(bb >> 24);
                op++;
                nb = 0, bb = 0;
            }
            return r;
        };

        switch (a.op) {
            case kEncPoll: res = poll(); break;
            case kEncCompress: res = compress(); break;
            case kEncFlush: res = flush(a.write_token != 0); break;
            case kEncCompressAndFlush:
                res = compress();
                if (res == kOk) res = flush(a.write_token != 0);
                break;
            default: res = kError; break;
        }

        // ---- state out ----
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane * 4; k < W; k += 256)
            *reinterpret_cast<uint32_t*>(gwin + k) = *reinterpret_cast<const uint32_t*>(win + k);
        if (lane < 4) sw[3 + lane] = reinterpret_cast<const uint32_t*>(ring)[lane];
        if (lane == 0) {
            sw[0] = bb;
            sw[1] = (wp & 0xFFFFu) | (nb << 16) | (in_size << 24);
            sw[2] = (in_pos & 0xFFu) | (wbits << 8) | (lbits << 16) | (flags << 24);
            sw[7] = ((uint32_t)cidx & 0xFFFFu) | (extp << 16);
            sw[8] = (csize & 0xFFu) | ((rle & 0xFFu) << 8) | ((extc & 0xFFu) << 16) | (lwf << 24);
            a.out_len[s] = op;
            a.status[s] = (int8_t)res;
            if (a.in_consumed) a.in_consumed[s] = ip;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace tamp_amd
