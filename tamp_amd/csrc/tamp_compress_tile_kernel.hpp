// tamp_compress_tile_kernel.hpp -- batch LZSS compressor for gfx950, third formulation: a ring of tile indexes in
// HISTORY coordinates.  One workgroup (256 threads) per stream, windows up to 2^10.
//
// Same contract as tamp_compress_kernel.hpp (per stream: tamp_compressor_init + tamp_compressor_compress_and_flush,
// tamp/_c_src/tamp/compressor.c:191-245,532-660,728-845; find_best_match compressor_find_match_desktop.c:82-167), same
// phases (index -> match every position speculatively -> walk -> emit), different bookkeeping:
//
//   * hist[] is a ring over the linear history E' = dictionary ++ every byte written to the window, addressed by the
//     history index h (mod 2048).  The window ring index of history byte h never changes: (h + wp0) mod W.  Behind the
//     written bytes the ring is filled speculatively with the input ("every token writes what it consumes"):
//     hist[Hb + j] = in[Pb + j].
//   * The bigram index is a ring of TILES of 512 history positions, each a small counting sort (256 buckets) built in
//     one pass when the parse enters the tile.  A query scans its bucket in the (at most three) tiles its window touches.
//     Nothing is re-indexed while the parse runs clean: the old kernel re-sorted W + block positions per epoch.
//   * A token that writes fewer bytes than it consumes (RLE runs over 8 bytes, tokens clipped at the ring end,
//     compressor.c:352-358,404-410) invalidates only what lies behind it: the current tile is re-indexed from the
//     bytes the input really delivers there and the rest of its positions are matched again.  The old kernel re-based
//     its buffer and rebuilt everything (4.9 epochs per 4 KiB of prose instead of 3).
//   * Tile k lists positions [512k - 3, 512k + 509) and serves queries whose history index lies in [512k, 512k + 512):
//     an entry carries the two bytes behind its bigram, so everything an entry says lies inside the region that is
//     rewritten when the parse diverges there.  Candidates one and two bytes in front of the query (their payload would
//     reach the query's own bytes) are tested directly.
#pragma once
#include "tamp_compress_kernel.hpp"

namespace tamp_amd {

constexpr uint32_t kTileLog2 = 9, kTile = 1u << kTileLog2;
constexpr uint32_t kHR = 2048, kHM = kHR - 1;  // history ring bytes: window (<= 1024) + tile + look-ahead
constexpr uint32_t kHMirror = 32;              // hist[kHR + i] mirrors hist[i]: unaligned multi-dword reads cross the ring end
constexpr uint32_t kTBucketBits = 10, kTBuckets = 1u << kTBucketBits, kTSlots = 3;  // (a query visits one bucket per tile: collisions = region / buckets)
constexpr uint32_t kTRem = 16 - kTBucketBits;  // bigram-mix bits carried by the entry
constexpr uint32_t kCursStride = kTBuckets + 4;  // u16 per tile: [0] = 0, [b + 1] = end of bucket b
constexpr uint32_t kLook = 288;                // input kept filled behind the tile: ring (16) + longest pending token (256) + slack
constexpr uint32_t kTSlowCap = 64;             // explicit token pieces per walk segment
constexpr uint32_t kTokCapT = kTile + 96;

struct TileLds {
    uint32_t hist, ent, curs, sorted, bins, blen, bidx, jc, toklist, stok, obuf, ctl, total, obuf_words;
    __host__ __device__ TileLds() {
        uint32_t o = 0;
        hist = o, o += kHR + kHMirror;
        ent = o, o += kTSlots * kTile * 4;
        curs = o, o += align_up(kTSlots * kCursStride * 2, 16);
        sorted = o, o += kTile * 2;
        bins = o, o += 64 * 4;
        blen = o, o += kTile + 64;
        bidx = o, o += kTile * 2;
        jc = o, o += kTile * 4;
        toklist = o, o += align_up(kTokCapT * 2, 16);
        stok = o, o += kTSlowCap * 8;
        obuf_words = 288;  // a round emits at most 9.5 bits per matched position + 64 explicit pieces, see DESIGN.md
        obuf = o, o += obuf_words * 4;
        ctl = o, o += 64 * 4 + 16;
        total = o;
    }
};

// Instruction-count experiments (-DTAMP_TILE_DBG): an idempotent section runs twice when its bit of CompressArgs::dbg is
// set; the difference of two rocprofv3 --pmc SQ_INSTS_VALU runs is that section's exact cost (tools/tile_sections.sh).
#ifdef TAMP_TILE_DBG
#define TILE_REP(bit) for (uint32_t rep_ = 0; rep_ < ((a.dbg & (bit)) ? 2u : 1u); rep_++)
#else
#define TILE_REP(bit)
#endif

#ifdef TAMP_TILE_DBG
#define TILE_MARK(i) do { __syncthreads(); const unsigned long long n_ = __builtin_readcyclecounter(); pt[i] += n_ - pc; pc = n_; } while (0)
#else
#define TILE_MARK(i) do { } while (0)
#endif

// ctl words of the tile kernel
enum : uint32_t { tAct = 0, tH = 1, tP = 2, tPend = 3, tDirty = 4, tNtok = 5, tExcess = 6, tWave = 8, tQlo = 12, tQhi = 13,
                  tHb = 14, tPb = 15, tPf = 16, tBuilds = 17 };
enum : uint32_t { kTActDone = 1, kTActBuild = 2, kTActContinue = 3 };

__device__ __forceinline__ void hist_put(uint8_t* hist, uint32_t off, uint32_t b) {
    hist[off] = (uint8_t)b;
    if (off < kHMirror) hist[off + kHR] = (uint8_t)b;
}

// Common prefix (0..16) of the 16 history bytes from ring offset `ca` with the pattern dwords P -- the candidate's
// first t bytes -- continued, from byte t on, with the bytes one window earlier (ring offset `cb` = ca - W): a
// candidate that reaches the newest window byte goes on with the OLDEST ones (the ring has not been overwritten there).
__device__ __forceinline__ uint32_t prefix_len_wrapped16_ring(const uint8_t* hist, uint32_t ca, uint32_t cb, uint32_t t,
                                                              const uint32_t (&P)[4]) {
    const uint32_t* wa = reinterpret_cast<const uint32_t*>(hist + (ca & ~3u));
    const uint32_t sa = ca & 3u;
    const uint32_t* wb = reinterpret_cast<const uint32_t*>(hist + (cb & ~3u));
    const uint32_t sb = cb & 3u;
    uint32_t res = 16;
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        const uint32_t xa = __builtin_amdgcn_alignbyte(wa[j + 1], wa[j], sa);
        const uint32_t xb = __builtin_amdgcn_alignbyte(wb[j + 1], wb[j], sb);
        const uint32_t lo = 4u * (uint32_t)j;
        const uint32_t m = t >= lo + 4 ? 0xFFFFFFFFu : (t <= lo ? 0u : (1u << (8 * (t - lo))) - 1u);
        const uint32_t x = ((xa & m) | (xb & ~m)) ^ P[j];
        if (x) res = lo + ((uint32_t)__builtin_ctz(x) >> 3);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// Walk state (wave 0, wave-uniform): the greedy parse over the match tables of the current tile, and the scalar
// state machine of poll_extended_handling (compressor.c:437-525) for RLE runs and extended matches.
// ---------------------------------------------------------------------------------------------
template <uint32_t WB>
struct TWalk {
    static constexpr uint32_t W = 1u << WB, mask = W - 1;
    uint8_t* hist;
    const uint8_t* blen;
    const uint16_t* bidx;
    uint16_t* toklist;
    uint32_t* stok;
    uint32_t wbits, lbits, minp;
    bool ext;
    uint32_t wp0;
    uint32_t H, P;          // history bytes (dictionary included) / input bytes consumed
    uint32_t Hb, Pb, Pf;    // speculative fill: hist[Hb + j] = in[Pb + j] for Pb + j < Pf
    uint32_t tbase, qlo, qhi;  // match tables: index h - tbase, valid for h in [qlo, qhi)
    int32_t dirty;          // lowest history index whose index entries may be stale (INT32_MAX: none)
    uint32_t rle_count, ext_count, ext_pos;
    bool ext_resolved;
    uint32_t ntok, ns;
    int lane;

    static __device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    __device__ __forceinline__ uint32_t wp() const { return (H + wp0) & mask; }
    // ring offset of the live window's byte with window index i
    __device__ __forceinline__ uint32_t woff(uint32_t i) const { return (H - W + ((i - wp()) & mask)) & kHM; }
    // ring offset of input byte P + k (inside the filled look-ahead)
    __device__ __forceinline__ uint32_t ioff(uint32_t k) const { return (Hb + (P - Pb) + k) & kHM; }
    __device__ __forceinline__ uint32_t win(uint32_t i) const { return uni(hist[woff(i)]); }
    __device__ __forceinline__ uint32_t inb(uint32_t k) const { return uni(hist[ioff(k)]); }
    __device__ __forceinline__ uint32_t win_l(uint32_t i) const { return hist[woff(i)]; }
    __device__ __forceinline__ uint32_t inb_l(uint32_t k) const { return hist[ioff(k)]; }
    __device__ __forceinline__ bool mapped() const { return H - Hb == P - Pb; }  // everything consumed has been written
    __device__ __forceinline__ void mark_dirty() {
        const int32_t d = (int32_t)H - 3;
        dirty = d < dirty ? d : dirty;
    }

    // Common prefix (<= lim) of the window from index i with the window from index j (other_is_window) or with the
    // input from byte P + j.  Dword compares while the window ranges stay in front of the newest byte.
    __device__ uint32_t common_l(uint32_t i, uint32_t j, uint32_t lim, bool other_is_window) const {
        const uint32_t ri = (i - wp()) & mask, rj = (j - wp()) & mask;
        const uint32_t a = H - W + ri, b = other_is_window ? H - W + rj : Hb + (P - Pb) + j;
        uint32_t contig = W - ri;
        if (other_is_window) contig = min(contig, W - rj);
        contig = min(contig, lim);
        uint32_t nn = 0;
        while (nn + 8 <= contig) {
            const uint32_t x0 = lds_u32_unaligned(hist, (a + nn) & kHM) ^ lds_u32_unaligned(hist, (b + nn) & kHM);
            const uint32_t x1 = lds_u32_unaligned(hist, (a + nn + 4) & kHM) ^ lds_u32_unaligned(hist, (b + nn + 4) & kHM);
            if (x0 | x1) return nn + (x0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : 4 + ((uint32_t)__builtin_ctz(x1) >> 3));
            nn += 8;
        }
        while (nn + 4 <= contig) {
            const uint32_t x = lds_u32_unaligned(hist, (a + nn) & kHM) ^ lds_u32_unaligned(hist, (b + nn) & kHM);
            if (x) return nn + ((uint32_t)__builtin_ctz(x) >> 3);
            nn += 4;
        }
        while (nn < lim && win_l(i + nn) == (other_is_window ? win_l(j + nn) : inb_l(j + nn))) nn++;
        return nn;
    }

    __device__ __forceinline__ void put(uint32_t v, uint32_t nb) {
        v = uni(v), nb = uni(nb);
        asm volatile("" : "+s"(v), "+s"(nb));
        if (lane == 0) {
            stok[2 * ns] = v;
            stok[2 * ns + 1] = nb;
            toklist[ntok] = (uint16_t)(0x8000u | ns);
        }
        ns++;
        ntok++;
    }

    // Append `cnt` bytes to the history.  `clean` = they are exactly the speculative bytes already sitting there.
    template <class F>
    __device__ __forceinline__ void append(uint32_t cnt, bool clean, F byte_at) {
        if (!clean) {
            mark_dirty();
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t b = byte_at(i);
                if (lane == 0) hist_put(hist, (H + i) & kHM, b);
            }
            __builtin_amdgcn_wave_barrier();
        }
        H += cnt;
    }

    __device__ __forceinline__ void put_exthuff(uint32_t value, uint32_t trailing) {  // compressor.c:257-263
        const uint32_t ci = value >> trailing;
        put((tok_code(ci) << trailing) | (value & ((1u << trailing) - 1)), (tok_nbits(ci) - 1) + trailing);
    }

    // write_rle_token (compressor.c:342-359); the run's `count` bytes are already consumed.
    __device__ void emit_rle(uint32_t count) {
        const uint32_t sym = uni(hist[(H - 1) & kHM]);  // the last byte written (compressor.c:270-273)
        put(tok_code(kSymRle), tok_nbits(kSymRle));
        put_exthuff(count - 2, 4);
        const uint32_t w = min(min(count, kRleWindowMax), W - wp());
        const bool clean = (H - Hb + count == P - Pb) && (w == count);
        if (w < count) mark_dirty();
        append(w, clean, [&](uint32_t) { return sym; });
    }

    // write_extended_match_token (compressor.c:377-415)
    __device__ void emit_ext() {
        const uint32_t count = ext_count, pos = ext_pos;
        put(tok_code(kSymExt), tok_nbits(kSymExt));
        put_exthuff(count - minp - 12, 3);
        put(pos, wbits);
        const uint32_t w = min(count, W - wp());
        const bool clean = (H - Hb + count == P - Pb) && (w == count);
        if (w < count) mark_dirty();
        const uint32_t H0 = H, wp0v = wp();
        // sources are read in the pre-token window; appended bytes land beyond it (memmove semantics, common.c:58-86)
        append(w, clean, [&](uint32_t i) { return uni(hist[(H0 - W + ((pos + i - wp0v) & mask)) & kHM]); });
        ext_count = 0;
    }

    // find_extended_match (compressor.c:297-333) as ONE search per extended match, candidates spread over the lanes
    // (see Walk::ext_search in tamp_compress_kernel.hpp for why one search equals the reference's rounds).
    __device__ void ext_search(uint32_t avail, uint32_t& npos, uint32_t& ncnt) {
        const uint32_t pos = ext_pos, cnt = ext_count;
        const uint32_t maxp = min(cnt + avail, minp + 11 + kExtExtraMax);
        // filter: candidate bytes cnt-3 .. cnt == the last three consumed bytes + the next input byte
        const uint32_t tail4 = uni(lds_u32_unaligned(hist, ioff(0u - 3u)));
        const uint32_t nextb = tail4 >> 24;
        const uint32_t wpv = wp();
        const uint32_t wbase = H - W;
        uint32_t key = 0;
        for (uint32_t c0 = pos + lane; c0 + cnt + 1 <= W; c0 += 16 * kWave) {
            uint32_t hits = 0;
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                if (k * kWave >= W) break;
                const uint32_t c = c0 + k * kWave;
                const bool valid = c + cnt + 1 <= W;
                const uint32_t r = ((valid ? c : pos) + cnt - 3 - wpv) & mask;  // oldest-first offset of byte cnt-3
                bool hit = lds_u32_unaligned(hist, (wbase + r) & kHM) == tail4;
                if (r > W - 4) hit = hist[(wbase + ((r + 3) & mask)) & kHM] == nextb;  // the four bytes straddle the write cursor
                hits |= (uint32_t)(valid && hit) << k;
            }
            while (hits) {
                const uint32_t k = (uint32_t)__builtin_ctz(hits);
                hits &= hits - 1;
                const uint32_t c = c0 + k * kWave;
                if (c != pos && common_l(c, pos, cnt, true) < cnt) continue;
                const uint32_t cmax = min(maxp, W - c);
                const uint32_t len = cnt + 1 + common_l(c + cnt + 1, 1, cmax - cnt - 1, false);
                const uint32_t kk = (len << 16) | (0xFFFFu - c);
                if ((kk >> 16) > (key >> 16)) key = kk;
            }
        }
        key = wave_max_u32(key);
        ncnt = key >> 16;
        npos = 0xFFFFu - (key & 0xFFFFu);
    }

    enum { kStepOk = 0, kStepRebase = 1, kStepExcess = 2 };

    __device__ __forceinline__ bool best(uint32_t& idx, uint32_t& len) {
        if (!mapped() || H < qlo || H >= qhi) return false;
        const uint32_t sv = uni(blen[H - tbase]);
        len = sv & 0x1Fu;
        idx = uni(bidx[H - tbase]);
        ext_resolved = (sv & 0x40u) != 0;
        return true;
    }

    // One parse step = tamp_compressor_poll (compressor.c:532-660) with the ring = next R input bytes.  Returns
    // kStepRebase *before mutating anything* when it needs a find_best_match result the tables cannot supply.
    __device__ int step(uint32_t R, uint32_t left) {
        uint32_t idx = 0, len = 0;
        if (ext) {
            if (ext_count) {  // compressor.c:439-468, all polls of the continuation at once
                const uint32_t max_ext = minp + 11 + kExtExtraMax;
                while (left > 0) {
                    if (ext_pos + ext_count >= W || ext_count >= max_ext) {
                        emit_ext();
                        return kStepOk;
                    }
                    uint32_t npos, ncnt;
                    const uint32_t reach = min(ext_count + left, max_ext);
                    ext_search(left, npos, ncnt);
                    if (ncnt > ext_count) {
                        const uint32_t extra = ncnt - ext_count;
                        ext_pos = npos;
                        ext_count = ncnt;
                        P += extra;
                        left -= extra;
                        if (ncnt == reach) continue;
                    }
                    emit_ext();
                    return kStepOk;
                }
                return kStepOk;
            }
            // RLE accumulation, compressor.c:470-525: the whole run at once (four bytes per lane, one ballot)
            const uint32_t last = uni(hist[(H - 1) & kHM]);
            uint32_t avail = 0;
            {
                const uint32_t x = lds_u32_unaligned(hist, ioff(4u * (uint32_t)lane)) ^ (last * 0x01010101u);
                const uint64_t bal = __ballot(x != 0);
                uint32_t run = 256;
                if (bal) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(bal);
                    const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)f);
                    run = 4u * f + ((uint32_t)__builtin_ctz(xf) >> 3);
                }
                avail = min(min(uni(run), left), kRleMax - rle_count);
            }
            const uint32_t total = rle_count + avail;
            const bool ended = (avail < left) || (total >= kRleMax);
            if (!ended && total > 0) {
                rle_count = total;
                P += avail;
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
                    P += avail;
                    emit_rle(total);
                    rle_count = 0;
                    return kStepOk;
                }
            } else if (rle_count == 1) {
                put((1u << lbits) | last, lbits + 1);
                append(1, H - Hb + 1 == P - Pb, [&](uint32_t) { return last; });
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
                if (ext_resolved) {
                    // no other candidate can carry the continuation: count the common prefix of that window position
                    // and the input (four bytes per lane) -- unless the match reaches the newest window byte
                    ext_resolved = false;
                    const uint32_t off = (idx - wp()) & mask, t0 = W - off;
                    const uint32_t xx = lds_u32_unaligned(hist, (H - W + off + 4u * (uint32_t)lane) & kHM) ^
                                        lds_u32_unaligned(hist, ioff(4u * (uint32_t)lane));
                    const uint64_t bal = __ballot(xx != 0);
                    uint32_t lcp = 256;
                    if (bal) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(bal);
                        const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)xx, (int)f);
                        lcp = 4u * f + ((uint32_t)__builtin_ctz(xf) >> 3);
                    }
                    const uint32_t cnt = min(min(lcp, W - idx), min(minp + 11 + kExtExtraMax, left));
                    if (cnt < t0 && cnt >= len) {
                        ext_pos = idx;
                        ext_count = cnt;
                        P += cnt;
                        emit_ext();
                        return kStepOk;
                    }
                }
                ext_count = len;
                ext_pos = idx;
                P += len;
                return kStepOk;
            }
            put((tok_code(len - minp) << wbits) | idx, tok_nbits(len - minp) + wbits);
        }
        // compressor.c:651-657: the consumed bytes enter the window
        const uint32_t src = ioff(0);
        P += len;
        append(len, H - Hb + len == P - Pb, [&](uint32_t i) { return uni(hist[(src + i) & kHM]); });
        return kStepOk;
    }
};

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------
template <uint32_t WB>
__global__ void __launch_bounds__(256, 6) tamp_compress_tile_kernel(CompressArgs a) {
    static_assert(WB >= 8 && WB <= 10, "history ring of 2048 bytes: windows up to 2^10");
    constexpr uint32_t W = 1u << WB, mask = W - 1;
    constexpr uint32_t NTW = (W >> kTileLog2) + 1 < 2 ? 2u : (W >> kTileLog2) + 1;  // tiles a window touches
    constexpr uint32_t nt = 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const TileLds L;
    uint8_t* const hist = smem + L.hist;
    uint32_t* const ent = reinterpret_cast<uint32_t*>(smem + L.ent);
    uint16_t* const curs16 = reinterpret_cast<uint16_t*>(smem + L.curs);
    uint32_t* const cursw = reinterpret_cast<uint32_t*>(smem + L.curs);
    uint16_t* const sorted = reinterpret_cast<uint16_t*>(smem + L.sorted);
    uint32_t* const bins = reinterpret_cast<uint32_t*>(smem + L.bins);
    uint8_t* const blen = smem + L.blen;
    uint16_t* const bidx = reinterpret_cast<uint16_t*>(smem + L.bidx);
    uint32_t* const jc32 = reinterpret_cast<uint32_t*>(smem + L.jc);
    uint16_t* const toklist = reinterpret_cast<uint16_t*>(smem + L.toklist);
    uint32_t* const stok = reinterpret_cast<uint32_t*>(smem + L.stok);
    uint32_t* const obuf = reinterpret_cast<uint32_t*>(smem + L.obuf);
    typedef __attribute__((address_space(3))) volatile uint32_t LdsCtl;
    LdsCtl* const ctl = (LdsCtl*)(smem + L.ctl);
    uint8_t* const codetab = smem + L.ctl + 256;

    const uint32_t tid_k = threadIdx.x;
    uint32_t tid = tid_k;
    int lane = tid & (kWave - 1);
    uint32_t wave = tid >> 6;
    const uint32_t lbits = a.lbits, wbits = WB;
    const uint32_t minp = (uint32_t)min_pattern_size((int)WB, (int)lbits);
    const bool ext = a.extended != 0;
    const uint32_t maxp = ext ? minp + 11 + kExtExtraMax : minp + 13;  // compressor.c:12-19
    if (tid_k < 15) codetab[tid_k] = (uint8_t)tok_code(tid_k);

    const uint32_t s = blockIdx.x + a.first_stream;
    const uint8_t* const in = a.in + uni_u64(a.in_off[s]);
    const uint32_t n = TWalk<WB>::uni(a.in_len[s]);
    uint8_t* const gout = a.out + uni_u64(a.out_off[s]);
    const uint32_t cap = TWalk<WB>::uni(a.out_cap[s]);
    uint8_t* const st_io = a.state ? a.state + (size_t)s * (W + kSegStateExtra) : nullptr;
    uint32_t wp0 = 0;
    if (st_io && (a.seg_flags & kSegResume)) {
        // history <- saved window (ring order) rotated so that the oldest byte comes first
        wp0 = (uint32_t)st_io[W] | ((uint32_t)st_io[W + 1] << 8);
        for (uint32_t k = tid; k < W; k += nt) hist[k] = st_io[(wp0 + k) & mask];
    } else if ((reinterpret_cast<uintptr_t>(a.dict) & 3) == 0) {
        for (uint32_t k = tid * 4; k < W; k += nt * 4)
            *reinterpret_cast<uint32_t*>(hist + k) = *reinterpret_cast<const uint32_t*>(a.dict + k);
    } else {
        for (uint32_t k = tid; k < W; k += nt) hist[k] = a.dict[k];
    }
    for (uint32_t k = tid; k < L.obuf_words; k += nt) obuf[k] = k == 0 ? __builtin_bswap32((uint32_t)a.lead << 16) : 0;
    // bucket ends of never-built tiles read as empty; curs[slot][0] stays 0 for good
    for (uint32_t k = tid; k < kTSlots * kCursStride / 2; k += nt) cursw[k] = 0;

    TWalk<WB> wk;
    wk.hist = hist, wk.blen = blen, wk.bidx = bidx, wk.toklist = toklist, wk.stok = stok;
    wk.wbits = wbits, wk.lbits = lbits, wk.minp = minp, wk.ext = ext, wk.wp0 = wp0;
    wk.H = W, wk.P = 0, wk.Hb = W, wk.Pb = 0, wk.Pf = 0, wk.tbase = 0, wk.qlo = 0, wk.qhi = 0;
    wk.dirty = -3;  // nothing is indexed yet: the first build lists every tile the window touches
    wk.rle_count = 0, wk.ext_count = 0, wk.ext_pos = 0, wk.ext_resolved = false, wk.ntok = 0, wk.ns = 0, wk.lane = lane;
    if (tid_k == 0) {
        ctl[tH] = W, ctl[tP] = 0, ctl[tPend] = 0, ctl[tDirty] = (uint32_t)-3;
    }

    uint32_t carry = 8u * a.nlead;  // bits already sitting in obuf
    uint32_t gpos = 0;              // bytes already flushed to HBM
    bool need_build = true;
    uint32_t b_tbase = 0, b_qlo = 0, b_qhi = 0;  // tables of the current build (all threads)
    __syncthreads();
#ifdef TAMP_TILE_DBG
    unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pc = __builtin_readcyclecounter();
#endif

    for (;;) {
        asm volatile("" : "+v"(tid));
        lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;
        if (need_build) {
            // ---------------- fill: hist[Hb + j] = in[Pb + j] ----------------
            const uint32_t bH = TWalk<WB>::uni(ctl[tH]), bP = TWalk<WB>::uni(ctl[tP]), bPend = TWalk<WB>::uni(ctl[tPend]);
            const int32_t bDirty = (int32_t)TWalk<WB>::uni(ctl[tDirty]);
            const uint32_t Hb = bH, Pb = bP - bPend;
            const uint32_t k = bH >> kTileLog2;
            const uint32_t tbase = k << kTileLog2;
            const uint32_t availb = n - Pb;
            const uint32_t flen = min(availb, (tbase + kTile - Hb) + kLook);
            TILE_REP(8u) {
                // destination-aligned dwords from (possibly unaligned) global dwords; head and tail bytes one by one
                const uint8_t* src = in + Pb;
                const uint32_t head = min((4u - (Hb & 3u)) & 3u, flen);
                const uint32_t ndw = (flen - head) >> 2;
                if (tid < head) hist[(Hb + tid) & kHM] = src[tid];
                for (uint32_t d = tid; d < ndw; d += nt) {
                    uint32_t v;
                    __builtin_memcpy(&v, src + head + 4 * d, 4);
                    *reinterpret_cast<uint32_t*>(hist + ((Hb + head + 4 * d) & kHM)) = v;
                }
                const uint32_t done = head + 4 * ndw;
                // tail bytes, then 24 zero bytes: stray look-ahead reads are defined
                for (uint32_t j = done + tid; j < flen + 24; j += nt) hist[(Hb + j) & kHM] = j < flen ? src[j] : 0;
            }
            __syncthreads();
            if (tid < kHMirror / 4) reinterpret_cast<uint32_t*>(hist + kHR)[tid] = reinterpret_cast<const uint32_t*>(hist)[tid];
            __syncthreads();

            TILE_MARK(0);
#ifdef TAMP_TILE_DBG
            pt[8] += 1;
#endif
            // ---------------- index: every tile whose listed bytes may have changed, oldest first ----------------
            asm volatile("" : "+v"(tid));
            lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;
            const uint32_t hend = Hb + flen;  // history bytes that exist (speculatively) right now
            uint32_t k0 = k;
            if (bDirty != 0x7FFFFFFF) k0 = bDirty + 3 <= 0 ? 0u : (uint32_t)(bDirty + 3) >> kTileLog2;
            if (k0 + (NTW - 1) < k) k0 = k - (NTW - 1);
            if (k0 > k) k0 = k;
            TILE_REP(4u)
            for (uint32_t kk = k0; kk <= k; kk++) {
                const uint32_t slot = kk % kTSlots;
                uint32_t* const cw = cursw + slot * (kCursStride / 2);
                uint16_t* const c16 = curs16 + slot * kCursStride;
                for (uint32_t j = tid; j < kCursStride / 2; j += nt) cw[j] = 0;
                __syncthreads();
                // tile kk lists positions [512 kk - 3, 512 kk + 509) whose bigram exists
                const int32_t h0 = (int32_t)(kk << kTileLog2) - 3 + 2 * (int32_t)tid;
                uint32_t b4a = 0, b4b = 0;
                bool va = false, vb = false;
                {
                    const uint32_t ha = (uint32_t)h0, hb2 = (uint32_t)(h0 + 1);
                    va = h0 >= 0 && ha + 1 < hend;
                    vb = h0 + 1 >= 0 && hb2 + 1 < hend;
                    if (va) b4a = lds_u32_unaligned(hist, ha & kHM);
                    if (vb) b4b = lds_u32_unaligned(hist, hb2 & kHM);
                }
                const uint32_t mxa = mix16(b4a & 0xFFFFu), mxb = mix16(b4b & 0xFFFFu);
                const uint32_t ba = (mxa >> kTRem) + 1, bb = (mxb >> kTRem) + 1;  // counter index of the bucket (end slot)
                if (va) atomicAdd(&cw[ba >> 1], 1u << ((ba & 1) * 16));
                if (vb) atomicAdd(&cw[bb >> 1], 1u << ((bb & 1) * 16));
                __syncthreads();
                {   // exclusive scan of the counters in place (thread t owns buckets 4t .. 4t+3)
                    constexpr uint32_t per = kTBuckets / nt;
                    uint32_t sum = 0;
#pragma unroll
                    for (uint32_t j = 0; j < per; j++) sum += c16[tid * per + j + 1];
                    const uint32_t incl = wave_scan_add(sum);
                    if (lane == kWave - 1) ctl[tWave + wave] = incl;
                    __syncthreads();
                    uint32_t run = incl - sum;
                    for (uint32_t w2 = 0; w2 < wave; w2++) run += ctl[tWave + w2];
#pragma unroll
                    for (uint32_t j = 0; j < per; j++) {
                        const uint32_t v = c16[tid * per + j + 1];
                        c16[tid * per + j + 1] = (uint16_t)run;
                        run += v;
                    }
                }
                __syncthreads();
                // entry = history index (11 bits) | rest of the bigram mix (8) | third byte (8) | low 5 bits of the fourth
                if (va) {
                    const uint32_t old = atomicAdd(&cw[ba >> 1], 1u << ((ba & 1) * 16));
                    ent[slot * kTile + ((old >> ((ba & 1) * 16)) & 0xFFFFu)] =
                        ((uint32_t)h0 & kHM) | ((mxa & ((1u << kTRem) - 1)) << 11) | (((b4a >> 16) & 0xFFu) << (11 + kTRem)) | ((b4a >> 24) << (19 + kTRem));
                }
                if (vb) {
                    const uint32_t old = atomicAdd(&cw[bb >> 1], 1u << ((bb & 1) * 16));
                    ent[slot * kTile + ((old >> ((bb & 1) * 16)) & 0xFFFFu)] =
                        ((uint32_t)(h0 + 1) & kHM) | ((mxb & ((1u << kTRem) - 1)) << 11) | (((b4b >> 16) & 0xFFu) << (11 + kTRem)) | ((b4b >> 24) << (19 + kTRem));
                }
                __syncthreads();
            }

            TILE_MARK(1);
            // ---------------- queries: history positions [qlo, qhi) of tile k, ordered by scan length ----------------
            const uint32_t qlo = bH + bPend;
            const uint32_t qend = min(tbase + kTile, Hb + availb);  // positions whose first byte is an input byte
            const uint32_t qhi = qend > qlo ? qend : qlo;
            const uint32_t nq = qhi - qlo;
            b_tbase = tbase, b_qlo = qlo, b_qhi = qhi;
            TILE_REP(16u) {
            if (tid < 64) bins[tid] = 0;
            __syncthreads();
            // the (at most three) tiles the windows of this tile's queries touch: k - NTW + 1 .. k
            for (uint32_t q = qlo + tid; q < qhi; q += nt) {
                const uint32_t b = mix16(lds_u32_unaligned(hist, q & kHM) & 0xFFFFu) >> kTRem;
                uint32_t tot = 0;
#pragma unroll
                for (uint32_t d = 0; d < NTW; d++) {
                    if (k >= d) {
                        const uint16_t* c16 = curs16 + ((k - d) % kTSlots) * kCursStride;
                        tot += (uint32_t)c16[b + 1] - (uint32_t)c16[b];
                    }
                }
                tot = min(tot, 63u);
                blen[q - tbase] = (uint8_t)tot;
                atomicAdd(&bins[63 - tot], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                const uint32_t v = bins[lane];
                const uint32_t incl = wave_scan_add(v);
                bins[lane] = incl - v;
            }
            __syncthreads();
            for (uint32_t q = qlo + tid; q < qhi; q += nt)
                sorted[atomicAdd(&bins[63 - (uint32_t)blen[q - tbase]], 1u)] = (uint16_t)(q - tbase);
            __syncthreads();
            }
            for (uint32_t j = (qhi - tbase) + tid; j < (qhi - tbase) + 64 && j < kTile + 64; j += nt) blen[j] = 0x80;  // sentinels

            TILE_MARK(2);
            // ---------------- match: find_best_match for every query ----------------
            asm volatile("" : "+v"(tid));
            lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;
            TILE_REP(2u)
            for (uint32_t j = tid; j < nq; j += nt) {
                const uint32_t ql = sorted[j];
                const uint32_t Hq = tbase + ql;            // history size when the parse stands here
                const uint32_t leftq = n - (Pb + (Hq - Hb));
                const uint32_t R = leftq < kRing ? leftq : kRing;
                uint32_t key = 0, wrapbest = 0, n16 = 0;
                bool sole_ext = false;
                uint32_t Pq[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) Pq[jj] = lds_u32_unaligned(hist, (Hq + 4 * jj) & kHM);
                // Inside a run of one byte the extended state machine never asks for a match (compressor.c:470-503 only
                // consults find_best_match for runs of 2..6 that END inside the ring)
                const uint32_t rep = (Pq[0] & 0xFFu) * 0x01010101u;
                const uint32_t prevb = hist[(Hq - 1) & kHM];
                bool in_run = false;
                if (ext && prevb == (Pq[0] & 0xFFu)) {
                    const uint32_t x0 = Pq[0] ^ rep, x1 = (Pq[1] ^ rep) & 0x00FFFFFFu;
                    const uint32_t r = x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : (x1 ? 4 + ((uint32_t)__builtin_ctz(x1) >> 3) : 7u);
                    in_run = r >= 7 || r >= R;
                }
                if (R >= minp && !in_run) {
                    const uint32_t cap_len = R < maxp ? R : maxp;
                    const uint32_t mx = mix16(Pq[0] & 0xFFFFu);
                    const uint32_t b = mx >> kTRem;
                    const uint32_t pk = ((mx & ((1u << kTRem) - 1)) << 11) | (((Pq[0] >> 16) & 0xFFu) << (11 + kTRem)) | ((Pq[0] >> 24) << (19 + kTRem));
                    // three segments (tiles k-2, k-1, k): virtual slot v -> ent[v + a_seg]
                    uint32_t l0 = 0, l01 = 0, Ltot = 0;
                    int32_t a0 = 0, a1 = 0, a2 = 0;
                    {
                        uint32_t acc = 0;
#pragma unroll
                        for (uint32_t d = NTW; d-- > 0;) {  // oldest tile first
                            uint32_t sb = 0, se = 0;
                            const uint32_t slot = (k - d) % kTSlots;
                            if (k >= d) {
                                const uint16_t* c16 = curs16 + slot * kCursStride;
                                sb = c16[b], se = c16[b + 1];
                            }
                            const int32_t adj = (int32_t)(slot * kTile + sb) - (int32_t)acc;
                            if (d == NTW - 1) a0 = adj;
                            if (d + 2 == NTW || (NTW == 2 && d == 0)) a1 = adj;
                            if (d == 0) a2 = adj;
                            acc += se - sb;
                            if (d == NTW - 1) l0 = acc;
                            if (d + 2 == NTW) l01 = acc;
                        }
                        if (NTW == 2) l01 = acc;  // two tiles: the third segment is empty
                        Ltot = acc;
                    }
                    const uint32_t wq = (wk.wp0 + 0u) & mask;  // window index = (history index + wp0) mod W
                    // (plain selects on values: a lambda over the three adjustments was compiled into a pointer select
                    // over stack slots -- a scratch load and a flat load per entry)
                    const uint32_t ua0 = (uint32_t)a0, ua1 = (uint32_t)a1, ua2 = (uint32_t)a2;
                    constexpr uint32_t kEntLast = kTSlots * kTile - 1;
                    uint32_t wrapmask = 0;
                    uint32_t e_next = ent[min(l0 ? ua0 : (l01 ? ua1 : ua2), kEntLast)];
#ifdef TAMP_TILE_DBG
                    const uint32_t n16_0 = n16;
                    for (uint32_t rep2_ = 0; rep2_ < ((a.dbg & 1u) ? 2u : 1u); rep2_++) {
                    if (rep2_) { n16 = n16_0; e_next = ent[min(l0 ? ua0 : (l01 ? ua1 : ua2), kEntLast)]; }
#endif
                    for (uint32_t v = 0; v < Ltot;) {
                        const uint32_t e = e_next;
                        v++;
                        uint32_t adj = v < l01 ? ua1 : ua2;
                        adj = v < l0 ? ua0 : adj;
                        e_next = ent[min(v + adj, kEntLast)];  // one past the range at the end: harmless
                        const uint32_t x = e ^ pk;
                        const uint32_t t = (Hq - e) & kHM;  // bytes from the candidate to the end of the written history
                        // in the window (3 <= t <= W; t = 1, 2 are tested below) and the same bigram
                        if (t - 3u <= W - 3u && (x & (((1u << kTRem) - 1) << 11)) == 0) {
                            uint32_t len = (x & (0xFFu << (11 + kTRem))) ? 2u : 3u;
                            if ((x >> (11 + kTRem)) == 0) len = prefix_len16(hist, e & kHM, Pq);
                            if (t < 16 && len >= t) {
                                wrapmask |= 1u << t;  // reaches the newest byte: goes on with the oldest ones, resolved below
                            } else {
                                n16 += len >> 4;
                                const uint32_t lim_i = W - ((e + wq) & mask);
                                key = max(key, (min(len, min(cap_len, lim_i)) << 16) | lim_i);
                            }
                        }
                    }
#ifdef TAMP_TILE_DBG
                    }
#endif
                    // the two candidates right in front of the query are not in any tile's list for it
                    {
                        const uint32_t n2 = lds_u32_unaligned(hist, (Hq - 2) & kHM);  // bytes Hq-2, Hq-1, then the pattern's
                        const uint32_t p01 = Pq[0] & 0xFFFFu;
                        if ((n2 & 0xFFFFu) == p01) wrapmask |= 4u;
                        if (((n2 >> 8) & 0xFFu) == (p01 & 0xFFu) && hist[(Hq - W) & kHM] == (p01 >> 8)) wrapmask |= 2u;
                    }
                    while (wrapmask) {
                        const uint32_t t = (uint32_t)__builtin_ctz(wrapmask);
                        wrapmask &= wrapmask - 1;
                        const uint32_t c = Hq - t;
                        const uint32_t i = (c + wq) & mask;
                        if (i == mask) continue;
                        const uint32_t lw = prefix_len_wrapped16_ring(hist, c & kHM, (c - W) & kHM, t, Pq);
                        wrapbest = max(wrapbest, lw);
                        const uint32_t len = min(lw, min(cap_len, W - i));
                        const uint32_t kk2 = (len << 16) | (W - i);
                        if (len >= 2 && kk2 > key) key = kk2;
                    }
                    // extended matches without a rival (see tamp_compress_kernel.hpp): flagged, the walk counts the
                    // common prefix instead of searching the window
                    const uint32_t len0 = key >> 16;
                    if (ext && len0 > minp + 11 && (len0 < 16 || (n16 == 1 && wrapbest < 16))) sole_ext = true;
                }
                const uint32_t len = key >> 16;
                bool slow = false;
                if (ext) {
                    const uint32_t b0 = Pq[0] & 0xFFu, b1 = (Pq[0] >> 8) & 0xFFu;
                    slow = (prevb == b0 && (b1 == b0 || R == 1)) || len > minp + 11;
                }
                blen[ql] = (uint8_t)(len | (slow ? 0x80u : 0u) | (sole_ext ? 0x40u : 0u));
                bidx[ql] = (uint16_t)(W - (key & 0xFFFFu));
            }
            __syncthreads();
            TILE_MARK(3);
#ifdef TAMP_TILE_DBG
            pt[9] += nq;
#endif
            // ---------------- jump tables: pointer doubling inside every 64-position block ----------------
            TILE_REP(32u)
            for (uint32_t b = wave * 64; b < kTile; b += (nt >> 6) * 64) {
                const uint32_t sv = blen[b + lane];
                const bool slowp = (sv & 0x80u) != 0;
                const uint32_t stepv = (sv & 0x1Fu) >= minp ? (sv & 0x1Fu) : 1u;
                constexpr uint32_t kFin = 1u << 18, kCnt = 0xFFu << 10;
                uint32_t st = slowp ? (((uint32_t)lane << 2) | kFin) : ((((uint32_t)lane + stepv) << 2) | (1u << 10));
                if ((st & 0x3FFu) >= 256u) st |= kFin;
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const uint32_t o2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(st & 0x3FFu), (int)st);
                    const uint32_t nw = o2 + (st & kCnt);
                    st = (st & kFin) ? st : nw;
                }
                st = ((st & 0x3FFu) >> 2) | (((st >> 10) & 0xFFu) << 8);
                jc32[b + lane] = (b + (st & 0xFFu)) | (((st >> 8) & 0xFFu) << 16);
            }
            if (tid == 0) {
                ctl[tQlo] = qlo, ctl[tQhi] = qhi, ctl[tHb] = Hb, ctl[tPb] = Pb, ctl[tPf] = Pb + flen;
            }
            __syncthreads();
        }

        TILE_MARK(4);
        // ---------------- walk: wave 0 ----------------
        if (wave == 0) {
            if (need_build) {
                wk.Hb = TWalk<WB>::uni(ctl[tHb]), wk.Pb = TWalk<WB>::uni(ctl[tPb]), wk.Pf = TWalk<WB>::uni(ctl[tPf]);
                wk.qlo = TWalk<WB>::uni(ctl[tQlo]), wk.qhi = TWalk<WB>::uni(ctl[tQhi]);
                wk.tbase = (wk.H >> kTileLog2) << kTileLog2;
                wk.dirty = 0x7FFFFFFF;
            }
            wk.ntok = 0, wk.ns = 0;
            uint32_t act = 0, excess_tok = 0xFFFFFFFFu;
            uint32_t nqueued = 0;
            uint32_t segv = 0;  // lane k: first position | first token slot << 16 of queued block k
            const uint32_t tb = wk.tbase;
            auto list_queued = [&]() {
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)lane < nqueued) {
                    uint32_t pp = segv & 0xFFFFu;
                    uint32_t slot = segv >> 16;
                    for (uint32_t cleft = jc32[pp] >> 16; cleft; cleft--) {
                        toklist[slot++] = (uint16_t)pp;
                        const uint32_t sv = blen[pp] & 0x1Fu;
                        pp += sv >= minp ? sv : 1u;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                nqueued = 0;
            };
            for (;;) {
                if (wk.ntok + 72 > kTokCapT || wk.ns + 8 > kTSlowCap) {
                    act = kTActContinue;
                    break;
                }
                const bool clean = wk.mapped() && wk.rle_count == 0 && wk.ext_count == 0;
                if (clean && wk.H >= wk.qlo && wk.H < wk.qhi) {
                    // plain steps: hop from block to block through the jump tables, queue the blocks passed
                    uint32_t pos = wk.H - tb;
                    const uint32_t nv = wk.qhi - tb;
                    uint32_t total = 0, nhop = 0;
                    while (pos < nv && nqueued < 64 && wk.ntok + total + 64 <= kTokCapT) {
                        const uint32_t jcv = TWalk<WB>::uni(jc32[pos]);
                        const uint32_t j = jcv & 0xFFFFu, cpos = jcv >> 16;
                        if (j == pos) break;  // a position the state machine has to look at
                        segv = (uint32_t)lane == nqueued ? (pos | ((wk.ntok + total) << 16)) : segv;
                        nqueued++, nhop++;
                        total += cpos;
                        pos = j;
                    }
                    if (nqueued > 64 - 10) list_queued();
                    wk.ntok += total;
                    wk.P += (tb + pos) - wk.H;
                    wk.H = tb + pos;
                    if (nhop) continue;
                }
                if (wk.P < n) {
                    const uint32_t leftp = n - wk.P;
                    // the look-ahead a slow step may read must be filled (a pending token's bytes sit in front of it)
                    const uint32_t pend = wk.rle_count + wk.ext_count;
                    const uint32_t need = pend ? wk.P - pend + 264u : wk.P + 272u;
                    int r = TWalk<WB>::kStepRebase;
                    if (wk.Pf >= (need < n ? need : n)) r = wk.step(leftp < kRing ? leftp : kRing, leftp);
                    if (r == TWalk<WB>::kStepRebase) {
                        act = kTActBuild;
                        break;
                    }
                    if (r == TWalk<WB>::kStepExcess) {
                        excess_tok = wk.ntok;
                        act = kTActDone;
                        break;
                    }
                } else if (ext && wk.rle_count >= 1) {  // compressor.c:748-763
                    if (wk.rle_count == 1) {
                        const uint32_t c = TWalk<WB>::uni(hist[(wk.H - 1) & kHM]);
                        wk.put((1u << lbits) | c, lbits + 1u);
                        wk.append(1, wk.H - wk.Hb + 1 == wk.P - wk.Pb, [&](uint32_t) { return c; });
                    } else {
                        wk.emit_rle(wk.rle_count);
                    }
                    wk.rle_count = 0;
                } else if (ext && wk.ext_count) {  // compressor.c:764-766
                    wk.emit_ext();
                } else {
                    if (st_io && (a.seg_flags & kSegSave)) {  // hand the window back in ring order
                        for (uint32_t i = lane; i < W; i += kWave) st_io[i] = (uint8_t)wk.win_l(i);
                        if (lane == 0) {
                            st_io[W] = (uint8_t)wk.wp();
                            st_io[W + 1] = (uint8_t)(wk.wp() >> 8);
                        }
                    }
                    act = kTActDone;
                    break;
                }
            }
            list_queued();
            if (lane == 0) {
                ctl[tAct] = act;
                ctl[tNtok] = wk.ntok;
                ctl[tExcess] = excess_tok;
                if (act == kTActBuild) {
                    ctl[tH] = wk.H, ctl[tP] = wk.P, ctl[tPend] = wk.rle_count + wk.ext_count, ctl[tDirty] = (uint32_t)wk.dirty;
                }
            }
        }
        __syncthreads();

        TILE_MARK(5);
#ifdef TAMP_TILE_DBG
        pt[10] += 1;
#endif
        // ---------------- emit: token list -> bits (all threads) ----------------
        asm volatile("" : "+v"(tid));
        lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;
        uint32_t act = ctl[tAct];
        const uint32_t ntok = ctl[tNtok];
        const uint32_t K = (ntok + nt - 1) >> 8;
        const uint32_t k0 = min(tid * K, ntok), k1 = min(k0 + K, ntok);
        auto token = [&](uint32_t k, uint32_t& v, uint32_t& nb) -> bool {  // false: literal with excess bits
            const uint32_t e = toklist[k];
            if (e & 0x8000u) {
                v = stok[2 * (e & 0x7FFFu)];
                nb = stok[2 * (e & 0x7FFFu) + 1];
                return true;
            }
            const uint32_t pos = e, len = blen[pos] & 0x1Fu, idx = bidx[pos];
            if (len < minp) {  // compressor.c:625-632
                const uint32_t c = hist[(b_tbase + pos) & kHM];
                v = (1u << lbits) | c;
                nb = lbits + 1;
                return (c >> lbits) == 0;
            }
            v = ((uint32_t)codetab[len - minp] << wbits) | idx;  // compressor.c:646-649
            nb = tok_nbits(len - minp) + wbits;
            return true;
        };
        if (lbits < 8) {  // TAMP_EXCESS_BITS: the stream ends at the first literal that does not fit
            uint32_t v, nb;
            for (uint32_t k = k0; k < k1; k++)
                if (!token(k, v, nb)) {
                    atomicMin((uint32_t*)&ctl[tExcess], k);
                    break;
                }
            __syncthreads();
        }
        const uint32_t limit = min((uint32_t)ctl[tExcess], ntok);
        const bool excess = ctl[tExcess] != 0xFFFFFFFFu;
        uint32_t mybits = 0;
        for (uint32_t k = k0; k < k1 && k < limit; k++) {
            uint32_t v, nb;
            token(k, v, nb);
            mybits += nb;
        }
        const uint32_t incl = wave_scan_add(mybits);
        if (lane == kWave - 1) ctl[tWave + wave] = incl;
        __syncthreads();
        uint32_t o = carry + incl - mybits, segbits = 0;
        for (uint32_t w2 = 0; w2 < (nt >> 6); w2++) {
            const uint32_t wt = ctl[tWave + w2];
            if (w2 < wave) o += wt;
            segbits += wt;
        }
        {  // MSb-first scatter of this thread's contiguous run of tokens
            uint32_t wi = o >> 5, ph = o & 31, fill = 0;
            uint64_t acc = 0;
            for (uint32_t k = k0; k < k1 && k < limit; k++) {
                uint32_t v, nb;
                token(k, v, nb);
                acc = (acc << nb) | v;
                fill += nb;
                while (ph + fill >= 32) {
                    const uint32_t take = 32 - ph;
                    uint32_t w = (uint32_t)(acc >> (fill - take));
                    if (take < 32) w &= (1u << take) - 1;
                    if (ph == 0)
                        obuf[wi] = __builtin_bswap32(w);
                    else
                        atomicOr(&obuf[wi], __builtin_bswap32(w));
                    fill -= take;
                    ph = 0;
                    wi++;
                }
            }
            if (fill) {
                const uint32_t w = ((uint32_t)acc & ((1u << fill) - 1)) << (32 - ph - fill);
                atomicOr(&obuf[wi], __builtin_bswap32(w));
            }
        }
        __syncthreads();
        uint32_t tot = carry + segbits;  // bits now in obuf
        if (excess) act = kTActDone;
        if (act == kTActDone && !excess && (a.seg_flags & kSegFlushToken)) {
            // compressor.c:784-794: FLUSH (9 bits) only if bits are pending or the stream allows dictionary resets
            const bool want = (tot & 7) != 0 || a.dict_reset;
            if (want && tid == 0) {
                const uint32_t wi = tot >> 5, ph = tot & 31, v = 0xABu;
                if (ph + 9 <= 32) {
                    obuf[wi] |= __builtin_bswap32(v << (32 - ph - 9));
                } else {
                    const uint32_t hi = 32 - ph;
                    obuf[wi] |= __builtin_bswap32(v >> (9 - hi));
                    obuf[wi + 1] |= __builtin_bswap32((v & ((1u << (9 - hi)) - 1)) << (32 - (9 - hi)));
                }
            }
            if (want) tot += 9;
            if (st_io && tid == 0) st_io[W + 2] = want ? 1 : 0;
            __syncthreads();
        }
        uint32_t nbytes;
        if (act == kTActDone)
            nbytes = excess ? (tot >> 3) : ((tot + 7) >> 3);  // compressor.c:629-631 / :799-807
        else
            nbytes = (tot >> 5) << 2;
        {   // HBM stores are whole aligned dwords whatever the slab's byte alignment
            const uint8_t* ob = reinterpret_cast<const uint8_t*>(obuf);
            uint8_t* dst = gout + gpos;
            const uint32_t room2 = gpos < cap ? cap - gpos : 0;
            const uint32_t nw = nbytes < room2 ? nbytes : room2;
            const uint32_t head = min((uint32_t)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3), nw);
            const uint32_t ndw = (nw - head) >> 2;
            if (tid < head) dst[tid] = ob[tid];
            uint32_t* dst32 = reinterpret_cast<uint32_t*>(dst + head);
            for (uint32_t k = tid; k < ndw; k += nt) dst32[k] = lds_u32_unaligned(ob, head + 4 * k);
            for (uint32_t k = head + 4 * ndw + tid; k < nw; k += nt) dst[k] = ob[k];
        }
        TILE_MARK(6);
        if (act == kTActDone) {
#ifdef TAMP_TILE_DBG
            if (tid == 0 && a.prof) for (int i = 0; i < 12; i++) atomicAdd(&a.prof[i], pt[i]);
#endif
            if (tid == 0) {
                const uint32_t total_bytes = gpos + nbytes;
                a.out_len[s] = total_bytes < cap ? total_bytes : cap;
                a.status[s] = total_bytes > cap ? kOutputFull : (excess ? kExcessBits : kOk);
            }
            break;
        }
        __syncthreads();
        {  // keep the partial last word, clear the rest for the next segment
            const uint32_t nfull = tot >> 5;
            const uint32_t lastw = obuf[nfull];
            const uint32_t used = ((tot + 31) >> 5) + 1;
            __syncthreads();
            for (uint32_t k = tid; k < used && k < L.obuf_words; k += nt) obuf[k] = k == 0 ? lastw : 0;
            gpos += nbytes;
            carry = tot & 31;
        }
        need_build = act == kTActBuild;
        __syncthreads();
    }
}

}  // namespace tamp_amd
