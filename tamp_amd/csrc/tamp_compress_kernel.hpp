// tamp_compress_kernel.hpp -- batch LZSS compressor for gfx950: one workgroup per stream.
//
// Replaces, per stream, tamp_compressor_init + tamp_compressor_compress_and_flush(write_token=false)
// (tamp/_c_src/tamp/compressor.c:191-245,815-845): find_best_match
// (compressor_find_match_desktop.c:82-167), tamp_compressor_poll (compressor.c:532-660),
// poll_extended_handling (compressor.c:437-525) and tamp_compressor_flush (compressor.c:728-810).
//
// Design (scalar model in oracle/tamp_model.c):
//   * The window is not kept as a ring.  LDS holds the linear history E' = dictionary ++ every byte
//     written to the window; the live window is its last W bytes, and input is appended behind it:
//         ebuf[0..W)   window at epoch start (oldest first)      ebuf[W+k] = input[p0+k]
//         window index of ebuf[c] = (wp_e + c) mod W
//   * MATCH PHASE (all threads): under the speculation "every consumed byte was written" the window
//     seen at input position q is ebuf[q..q+W) no matter how earlier bytes were parsed, so
//     find_best_match is evaluated for every position of the block at once.  Candidates come from a
//     counting-sorted bigram index over the buffer (the reference needs a 2-byte prefix hit too),
//     so work per position is O(#positions sharing the bigram), not O(W).
//   * WALK (wave 0, all lanes computing the same scalars): the greedy parse, RLE / extended-match
//     state machine and bit packing.  Tokens that write fewer bytes than they consume
//     (compressor.c:352-358,404-410) break the speculation; the next find_best_match request then
//     re-bases the buffer and starts a new epoch.
#pragma once
#include "tamp_common.hpp"

namespace tamp_amd {

constexpr uint32_t kHashBits = 11;
constexpr uint32_t kHashBuckets = 1u << kHashBits;
constexpr uint32_t kObuf = 1024;  // output staging bytes in LDS (multiple of 4)

struct CompressArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    const uint32_t* in_len;
    uint8_t* out;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    uint32_t* out_len;
    int8_t* status;
    const uint8_t* dict;  // 1<<wbits bytes: the custom dictionary or the seeded default
    uint32_t n_streams;
    uint32_t blk;  // epoch block: positions matched per epoch (multiple of 16)
    uint8_t wbits, lbits, extended, header, dict_reset;
    unsigned long long* prof;  // optional: per-phase cycle sums (debug builds with -DTAMP_PROF)
};

// LDS carve-up, shared by the host launcher and the kernel.
struct CompressLds {
    uint32_t ebuf, cnt, ent, blen, bidx, obuf, ctl, total;
    __host__ __device__ CompressLds(uint32_t W, uint32_t blk) {
        uint32_t o = 0;
        ebuf = o;
        o += align_up(W + blk + kRing + kPendMax + 32, 16);
        cnt = o;
        o += kHashBuckets * 4;
        ent = o;
        o += align_up((W + blk) * 2, 16);
        blen = o;
        o += align_up(blk, 16);
        bidx = o;
        o += align_up(blk * 2, 16);
        obuf = o;
        o += kObuf;
        ctl = o;
        o += 64;
        total = o;
    }
};

__device__ __forceinline__ uint32_t bigram_hash(uint32_t pair16) {
    return ((pair16 * 40503u) >> (16 - kHashBits)) & (kHashBuckets - 1);
}

// ---------------------------------------------------------------------------------------------
// Walk state: lives in registers of wave 0, identical in every lane.
// ---------------------------------------------------------------------------------------------
struct Walk {
    uint8_t* ebuf;
    const uint8_t* blen;
    const uint16_t* bidx;
    uint8_t* obuf;
    uint8_t* gout;
    uint32_t cap;
    uint32_t W, mask, wbits, lbits, minp;
    bool ext;
    uint32_t n;       // stream length
    uint32_t p0;      // input position of ebuf[W]
    uint32_t wp_e;    // window_pos at epoch start
    uint32_t wr, rd;  // bytes written / consumed since epoch start
    uint32_t nvalid, blk;
    uint32_t rle_count, ext_count, ext_pos;
    uint64_t acc;     // pending output bits, right aligned
    uint32_t nacc;    // < 32 between tokens
    uint32_t opos;    // bytes staged in obuf (multiple of 4)
    uint32_t gpos;    // bytes already copied to global
    uint32_t tbits;   // total bits emitted
    int lane;

    __device__ __forceinline__ uint32_t wp() const { return (wp_e + wr) & mask; }
    // byte at window index i of the live window ebuf[wr .. wr+W)
    __device__ __forceinline__ uint32_t win(uint32_t i) const { return ebuf[wr + ((i - wp()) & mask)]; }
    __device__ __forceinline__ uint32_t inb(uint32_t k) const { return ebuf[W + rd + k]; }

    __device__ void flush_stage() {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k < opos; k += kWave) {
            uint8_t b = obuf[k];
            if (gpos + k < cap) gout[gpos + k] = b;
        }
        __builtin_amdgcn_wave_barrier();
        gpos += opos;
        opos = 0;
    }

    // write_to_bit_buffer + partial_flush (compressor.c:49-52,65-75): MSb-first append.
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb) {
        acc = (acc << nb) | v;
        nacc += nb;
        tbits += nb;
        if (nacc >= 32) {
            uint32_t word = (uint32_t)(acc >> (nacc - 32));
            if (lane == 0) *reinterpret_cast<uint32_t*>(obuf + opos) = __builtin_bswap32(word);
            opos += 4;
            nacc -= 32;
            if (opos == kObuf) flush_stage();
        }
    }

    // Move the (< 32) pending bits that form whole bytes into the staging buffer and copy it out.
    __device__ void drain_whole_bytes() {
        uint32_t nbytes = nacc >> 3;
        for (uint32_t j = 0; j < nbytes; j++) {
            uint8_t b = (uint8_t)(acc >> (nacc - 8 * (j + 1)));
            if (lane == 0) obuf[opos + j] = b;
        }
        opos += nbytes;
        nacc -= 8 * nbytes;
        flush_stage();
    }

    // Append `cnt` bytes to the history.  `clean` = the bytes are exactly the input bytes already
    // sitting at that place (speculation intact), so nothing has to be stored.
    template <class F>
    __device__ __forceinline__ void append(uint32_t cnt, bool clean, F byte_at) {
        if (!clean) {
            for (uint32_t i = 0; i < cnt; i++) {
                uint32_t b = byte_at(i);
                if (lane == 0) ebuf[W + wr + i] = (uint8_t)b;
            }
            __builtin_amdgcn_wave_barrier();
        }
        wr += cnt;
    }

    __device__ __forceinline__ void put_exthuff(uint32_t value, uint32_t trailing) {  // compressor.c:257-263
        uint32_t ci = value >> trailing;
        put(((uint32_t)d_code[ci] << trailing) | (value & ((1u << trailing) - 1)), (d_nbits[ci] - 1) + trailing);
    }

    // write_rle_token (compressor.c:342-359); the run's `count` bytes are already consumed.
    __device__ void emit_rle(uint32_t count) {
        const uint32_t sym = win((wp() - 1) & mask);
        put(d_code[kSymRle], d_nbits[kSymRle]);
        put_exthuff(count - 2, 4);
        uint32_t w = min(min(count, kRleWindowMax), W - wp());
        const bool clean = (wr + count == rd) && (w == count);
        append(w, clean, [&](uint32_t) { return sym; });
    }

    // write_extended_match_token (compressor.c:377-415)
    __device__ void emit_ext() {
        const uint32_t count = ext_count, pos = ext_pos;
        put(d_code[kSymExt], d_nbits[kSymExt]);
        put_exthuff(count - minp - 12, 3);
        put(pos, wbits);
        uint32_t w = min(count, W - wp());
        const bool clean = (wr + count == rd) && (w == count);
        const uint32_t wr0 = wr, wp0 = wp();
        // Sources are read in the pre-token window; appended bytes land beyond it (memmove
        // semantics of tamp_window_copy, common.c:58-86, for free).
        append(w, clean, [&](uint32_t i) { return (uint32_t)ebuf[wr0 + ((pos + i - wp0) & mask)]; });
        ext_count = 0;
    }

    // find_extended_match (compressor.c:297-333), candidates spread over the 64 lanes.
    __device__ void ext_search(uint32_t R, uint32_t& npos, uint32_t& ncnt) {
        const uint32_t pos = ext_pos, cnt = ext_count;
        const uint32_t maxp = min(cnt + R, minp + 11 + kExtExtraMax);
        const uint32_t nextb = inb(0);
        uint32_t key = 0;
        for (uint32_t c = pos + lane; c + cnt + 1 <= W; c += kWave) {
            if (win(c + cnt) != nextb) continue;
            uint32_t i = 0;
            while (i < cnt && win(c + i) == win(pos + i)) i++;
            if (i < cnt) continue;
            const uint32_t cmax = min(maxp, W - c);
            uint32_t len = cnt + 1;
            while (len < cmax && win(c + len) == inb(len - cnt)) len++;
            uint32_t k = (len << 16) | (0xFFFFu - c);
            if ((k >> 16) > (key >> 16)) key = k;  // first-longest within this lane (c ascending)
        }
        key = wave_max_u32(key);  // longest; ties -> lowest candidate
        ncnt = key >> 16;
        npos = 0xFFFFu - (key & 0xFFFFu);
    }

    enum { kStepOk = 0, kStepRebase = 1, kStepExcess = 2 };

    __device__ __forceinline__ bool best(uint32_t& idx, uint32_t& len) const {
        if (wr != rd || rd >= nvalid) return false;
        len = blen[rd];
        idx = bidx[rd];
        return true;
    }

    // One parse step = tamp_compressor_poll (compressor.c:532-660) with the ring = next R input bytes.
    // Returns kStepRebase *before mutating anything* when it needs a find_best_match result that the
    // current epoch cannot supply.
    __device__ int step(uint32_t R) {
        uint32_t idx = 0, len = 0;
        if (ext) {
            if (ext_count) {  // compressor.c:439-468
                const uint32_t max_ext = minp + 11 + kExtExtraMax;
                while (R > 0) {
                    if (ext_pos + ext_count >= W || ext_count >= max_ext) {
                        emit_ext();
                        return kStepOk;
                    }
                    uint32_t npos, ncnt;
                    ext_search(R, npos, ncnt);
                    if (ncnt > ext_count) {
                        uint32_t extra = ncnt - ext_count;
                        ext_pos = npos;
                        ext_count = ncnt;
                        rd += extra;
                        R -= extra;
                        continue;
                    }
                    emit_ext();
                    return kStepOk;
                }
                return kStepOk;
            }
            // RLE accumulation, compressor.c:470-525
            const uint32_t last = win((wp() - 1) & mask);
            uint32_t avail = 0;
            while (avail < R && rle_count + avail < kRleMax && inb(avail) == last) avail++;
            const uint32_t total = rle_count + avail;
            const bool ended = (avail < R) || (total >= kRleMax);
            if (!ended && total > 0) {
                rle_count = total;
                rd += avail;
                return kStepOk;
            }
            if (total >= 2) {
                bool use_pattern = false;
                if (total == avail && total <= 6) {
                    if (!best(idx, len)) return kStepRebase;
                    if (len > total)
                        use_pattern = true;
                    else
                        len = 0;
                }
                if (!use_pattern) {
                    rd += avail;
                    emit_rle(total);
                    rle_count = 0;
                    return kStepOk;
                }
            } else if (rle_count == 1) {
                put((1u << lbits) | last, lbits + 1);
                append(1, wr + 1 == rd, [&](uint32_t) { return last; });
                rle_count = 0;
                return kStepOk;
            }
        }
        if (len == 0 && !best(idx, len)) return kStepRebase;

        if (len < minp) {  // literal, compressor.c:625-632
            const uint32_t c = inb(0);
            if (c >> lbits) return kStepExcess;
            put((1u << lbits) | c, lbits + 1);
            len = 1;
        } else {
            if (ext && len > minp + 11) {  // compressor.c:636-644
                ext_count = len;
                ext_pos = idx;
                rd += len;
                return kStepOk;
            }
            put(((uint32_t)d_code[len - minp] << wbits) | idx, d_nbits[len - minp] + wbits);
        }
        // compressor.c:651-657: the consumed bytes enter the window
        const uint32_t rd0 = rd;
        rd += len;
        append(len, wr + len == rd, [&](uint32_t i) { return (uint32_t)ebuf[W + rd0 + i]; });
        return kStepOk;
    }
};

// ---------------------------------------------------------------------------------------------
// Match phase helpers
// ---------------------------------------------------------------------------------------------

// Length (0..16) of the common prefix of ebuf[c..c+16) and the pattern dwords P[0..3].
__device__ __forceinline__ uint32_t prefix_len16(const uint8_t* ebuf, uint32_t c, const uint32_t (&P)[4]) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(ebuf + (c & ~3u));
    const uint32_t sh = c & 3u;
    uint32_t lo = w[0], hi = w[1];
    uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ P[0];
    if (x) return (uint32_t)__builtin_ctz(x) >> 3;
    lo = hi;
    hi = w[2];
    x = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ P[1];
    if (x) return 4 + ((uint32_t)__builtin_ctz(x) >> 3);
    lo = hi;
    hi = w[3];
    x = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ P[2];
    if (x) return 8 + ((uint32_t)__builtin_ctz(x) >> 3);
    lo = hi;
    hi = w[4];
    x = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ P[3];
    if (x) return 12 + ((uint32_t)__builtin_ctz(x) >> 3);
    return 16;
}

// Candidate whose bytes may run past the newest window byte (ring continues with the oldest).
__device__ __forceinline__ uint32_t prefix_len_wrapped(const uint8_t* ebuf, uint32_t c, uint32_t q, uint32_t W,
                                                       uint32_t lim) {
    uint32_t len = 0;
    while (len < lim) {
        uint32_t s = c + len;
        if (s >= q + W) s -= W;
        if (ebuf[s] != ebuf[W + q + len]) break;
        len++;
    }
    return len;
}

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tamp_compress_kernel(CompressArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t W = 1u << a.wbits, mask = W - 1;
    const CompressLds L(W, a.blk);
    uint8_t* const ebuf = smem + L.ebuf;
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(smem + L.cnt);
    uint16_t* const ent = reinterpret_cast<uint16_t*>(smem + L.ent);
    uint8_t* const blen = smem + L.blen;
    uint16_t* const bidx = reinterpret_cast<uint16_t*>(smem + L.bidx);
    uint8_t* const obuf = smem + L.obuf;
    volatile uint32_t* const ctl = reinterpret_cast<volatile uint32_t*>(smem + L.ctl);

    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    const uint32_t wave = tid >> 6;
    const uint32_t minp = (uint32_t)min_pattern_size(a.wbits, a.lbits);
    const bool ext = a.extended != 0;
    const uint32_t maxp = ext ? minp + 11 + kExtExtraMax : minp + 13;  // compressor.c:12-19

    for (uint32_t s = blockIdx.x; s < a.n_streams; s += gridDim.x) {
        const uint8_t* const in = a.in + a.in_off[s];
        const uint32_t n = a.in_len[s];

        // window <- dictionary (custom, or the seeded default prepared by the host shim)
        if ((reinterpret_cast<uintptr_t>(a.dict) & 3) == 0) {
            for (uint32_t k = tid * 4; k < W; k += nt * 4)
                *reinterpret_cast<uint32_t*>(ebuf + k) = *reinterpret_cast<const uint32_t*>(a.dict + k);
        } else {
            for (uint32_t k = tid; k < W; k += nt) ebuf[k] = a.dict[k];
        }

        Walk wk;
        wk.ebuf = ebuf, wk.blen = blen, wk.bidx = bidx, wk.obuf = obuf;
        wk.gout = a.out + a.out_off[s], wk.cap = a.out_cap[s];
        wk.W = W, wk.mask = mask, wk.wbits = a.wbits, wk.lbits = a.lbits, wk.minp = minp, wk.ext = ext;
        wk.n = n, wk.p0 = 0, wk.wp_e = 0, wk.wr = 0, wk.rd = 0, wk.nvalid = 0, wk.blk = a.blk;
        wk.rle_count = 0, wk.ext_count = 0, wk.ext_pos = 0;
        wk.acc = 0, wk.nacc = 0, wk.opos = 0, wk.gpos = 0, wk.tbits = 0, wk.lane = lane;
        wk.put(a.header, 8);  // compressor.c:236-241
        if (a.dict_reset) wk.put(0, 8);

        uint32_t e_p0 = 0, e_pending = 0, e_wp = 0;  // epoch parameters, uniform over the workgroup
#ifdef TAMP_PROF
        unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long pc = __builtin_readcyclecounter();
#define TAMP_PROF_MARK(i) do { __syncthreads(); unsigned long long _n = __builtin_readcyclecounter(); pt[i] += _n - pc; pc = _n; } while (0)
#else
#define TAMP_PROF_MARK(i) do { } while (0)
#endif
        for (;;) {
            // ---------------- load: ebuf[W + k] = in[e_p0 + k] ----------------
            const uint32_t left = n - e_p0;
            const uint32_t room = a.blk + kRing + kPendMax;
            const uint32_t nload = left < room ? left : room;
            const uint32_t nvalid = left < a.blk ? left : a.blk;
            {
                const uint8_t* src = in + e_p0;
                const uint32_t nfill = align_up(nload + 16, 4);  // zero tail so stray look-ahead reads are defined
                if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
                    const uint32_t nw = nload >> 2;
                    for (uint32_t k = tid; k < nw; k += nt)
                        reinterpret_cast<uint32_t*>(ebuf + W)[k] = reinterpret_cast<const uint32_t*>(src)[k];
                    for (uint32_t k = (nw << 2) + tid; k < nfill; k += nt) ebuf[W + k] = k < nload ? src[k] : 0;
                } else {
                    for (uint32_t k = tid; k < nfill; k += nt) ebuf[W + k] = k < nload ? src[k] : 0;
                }
            }
            for (uint32_t k = tid; k < kHashBuckets; k += nt) cnt[k] = 0;
            __syncthreads();
            TAMP_PROF_MARK(0);

            // ---------------- index: counting sort of buffer positions by bigram hash ----------------
            // positions c in [0, NE): every candidate any query of this block may need
            const uint32_t NE = nvalid >= 1 ? W + nvalid - 2 : 0;
            for (uint32_t c4 = tid * 4; c4 < NE; c4 += nt * 4) {
                const uint32_t d0 = *reinterpret_cast<const uint32_t*>(ebuf + c4);
                const uint32_t d1 = *reinterpret_cast<const uint32_t*>(ebuf + c4 + 4);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    if (c4 + j < NE) {
                        const uint32_t pair = __builtin_amdgcn_alignbyte(d1, d0, j) & 0xFFFFu;
                        atomicAdd(&cnt[bigram_hash(pair)], 1u);
                    }
                }
            }
            __syncthreads();
            {  // exclusive scan of cnt[0..HB) in place
                const uint32_t per = kHashBuckets / nt;  // 8 (256 threads) or 32 (64 threads)
                uint32_t sum = 0;
                for (uint32_t k = 0; k < per; k++) sum += cnt[tid * per + k];
                uint32_t incl = sum;
#pragma unroll
                for (int off = 1; off < kWave; off <<= 1) {
                    uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                    if (lane >= off) incl += o;
                }
                if (lane == kWave - 1) ctl[8 + wave] = incl;
                __syncthreads();
                uint32_t base = 0;
                for (uint32_t w2 = 0; w2 < wave; w2++) base += ctl[8 + w2];
                uint32_t run = base + incl - sum;
                for (uint32_t k = 0; k < per; k++) {
                    uint32_t v = cnt[tid * per + k];
                    cnt[tid * per + k] = run;
                    run += v;
                }
            }
            __syncthreads();
            for (uint32_t c4 = tid * 4; c4 < NE; c4 += nt * 4) {
                const uint32_t d0 = *reinterpret_cast<const uint32_t*>(ebuf + c4);
                const uint32_t d1 = *reinterpret_cast<const uint32_t*>(ebuf + c4 + 4);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    if (c4 + j < NE) {
                        const uint32_t pair = __builtin_amdgcn_alignbyte(d1, d0, j) & 0xFFFFu;
                        const uint32_t slot = atomicAdd(&cnt[bigram_hash(pair)], 1u);
                        ent[slot] = (uint16_t)(c4 + j);
                    }
                }
            }
            __syncthreads();
            TAMP_PROF_MARK(1);
            // now bucket h = ent[(h ? cnt[h-1] : 0) .. cnt[h])

            // ---------------- match phase: find_best_match for every position of the block ----------------
            for (uint32_t q = e_pending + tid; q < nvalid; q += nt) {
                const uint32_t leftq = n - (e_p0 + q);
                const uint32_t R = leftq < kRing ? leftq : kRing;
                uint32_t key = 0;
                if (R >= minp) {
                    const uint32_t cap = R < maxp ? R : maxp;
                    uint32_t P[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) P[j] = lds_u32_unaligned(ebuf, W + q + 4 * j);
                    const uint32_t h = bigram_hash(P[0] & 0xFFFFu);
                    const uint32_t s0 = h ? cnt[h - 1] : 0, s1 = cnt[h];
                    const uint32_t chi = q + W - 2;  // last candidate served by the index
                    for (uint32_t sl = s0; sl < s1; sl++) {
                        const uint32_t c = ent[sl];
                        if (c < q || c > chi) continue;
                        const uint32_t i = (e_wp + c) & mask;  // window index of the candidate
                        if (i == mask) continue;               // index W-1 cannot start a match
                        const uint32_t lim = min(cap, W - i);  // may not run past index W-1
                        uint32_t len;
                        if (c + 16 <= q + W)
                            len = min(prefix_len16(ebuf, c, P), lim);
                        else
                            len = prefix_len_wrapped(ebuf, c, q, W, lim);
                        const uint32_t k = (len << 16) | (0xFFFFu - i);
                        if (len >= 2 && k > key) key = k;
                    }
                    {  // the newest window byte pairs with the OLDEST one: not in the index
                        const uint32_t c = q + W - 1;
                        const uint32_t i = (e_wp + c) & mask;
                        if (i != mask) {
                            const uint32_t len = prefix_len_wrapped(ebuf, c, q, W, min(cap, W - i));
                            const uint32_t k = (len << 16) | (0xFFFFu - i);
                            if (len >= 2 && k > key) key = k;
                        }
                    }
                }
                blen[q] = (uint8_t)(key >> 16);
                bidx[q] = (uint16_t)(0xFFFFu - (key & 0xFFFFu));
            }
            __syncthreads();
            TAMP_PROF_MARK(2);

            // ---------------- walk: wave 0 ----------------
            if (wave == 0) {
                wk.nvalid = nvalid;
                uint32_t done = 0, result = 0, total_bytes = 0;
                for (;;) {
                    const uint32_t p = wk.p0 + wk.rd;
                    if (p < n) {
                        const uint32_t pending = wk.rle_count + wk.ext_count;
                        int r = Walk::kStepRebase;
                        if (wk.rd <= wk.blk + pending) {
                            const uint32_t leftp = n - p;
                            r = wk.step(leftp < kRing ? leftp : kRing);
                        }
                        if (r == Walk::kStepRebase) break;
                        if (r == Walk::kStepExcess) {
                            total_bytes = wk.tbits >> 3;  // whole bytes emitted before the offending literal
                            wk.drain_whole_bytes();
                            result = (uint32_t)(int32_t)kExcessBits;
                            done = 1;
                            break;
                        }
                    } else if (ext && wk.rle_count >= 1) {  // compressor.c:748-763
                        if (wk.rle_count == 1) {
                            const uint32_t c = wk.win((wk.wp() - 1) & mask);
                            wk.put((1u << a.lbits) | c, a.lbits + 1u);
                            wk.append(1, wk.wr + 1 == wk.rd, [&](uint32_t) { return c; });
                        } else {
                            wk.emit_rle(wk.rle_count);
                        }
                        wk.rle_count = 0;
                    } else if (ext && wk.ext_count) {  // compressor.c:764-766
                        wk.emit_ext();
                    } else {
                        if (wk.nacc & 7) wk.put(0, 8 - (wk.nacc & 7));  // compressor.c:799-807
                        total_bytes = wk.tbits >> 3;
                        wk.drain_whole_bytes();
                        result = (uint32_t)(int32_t)kOk;
                        done = 1;
                        break;
                    }
                }
                if (done) {
                    if (total_bytes > wk.cap) {
                        result = (uint32_t)(int32_t)kOutputFull;
                        total_bytes = wk.cap;
                    }
                    if (lane == 0) {
                        a.out_len[s] = total_bytes;
                        a.status[s] = (int8_t)(int32_t)result;
                    }
                } else {
                    // re-base request: drop the lag, keep bytes a pending RLE run / extended match has
                    // consumed but not written (oracle/tamp_model.c m_epoch_begin)
                    const uint32_t pending = wk.rle_count + wk.ext_count;
                    const uint32_t shift = wk.wr;
                    wk.wp_e = wk.wp();
                    wk.p0 += wk.rd - pending;
                    wk.wr = 0;
                    wk.rd = pending;
                    if (lane == 0) {
                        ctl[1] = shift;
                        ctl[2] = wk.p0;
                        ctl[3] = pending;
                        ctl[4] = wk.wp_e;
                    }
                }
                if (lane == 0) ctl[0] = done;
            }
            __syncthreads();
            TAMP_PROF_MARK(3);
            if (ctl[0]) break;
            const uint32_t shift = ctl[1];
            e_p0 = ctl[2];
            e_pending = ctl[3];
            e_wp = ctl[4];
            // ---------------- re-base: ebuf[0..W) <- ebuf[shift..shift+W) (moving left, chunked) ----------------
            if (shift) {
                for (uint32_t base = 0; base < W; base += nt * 4) {
                    const uint32_t k = base + tid * 4;
                    uint32_t v = 0;
                    if (k < W) v = lds_u32_unaligned(ebuf, shift + k);
                    __syncthreads();
                    if (k < W) *reinterpret_cast<uint32_t*>(ebuf + k) = v;
                    __syncthreads();
                }
            }
        }
#ifdef TAMP_PROF
        if (tid == 0 && a.prof) for (int i = 0; i < 6; i++) atomicAdd(&a.prof[i], pt[i]);
#endif
        __syncthreads();  // ctl / LDS reuse by the next stream
    }
}

}  // namespace tamp_amd
