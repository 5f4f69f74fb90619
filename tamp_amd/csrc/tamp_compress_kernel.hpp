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
//   * INDEX (all threads): counting sort of buffer positions by bigram (find_best_match needs a 2-byte
//     prefix hit), scattered tile by tile so every bucket is position-ordered at tile granularity.
//     An entry carries the position, the rest of the bigram and the next two bytes, so that 2- and
//     3-byte candidates are classified without touching the buffer.
//   * MATCH (all threads, one position per lane): under the speculation "every consumed byte was
//     written" the window seen at input position q is ebuf[q..q+W) however earlier bytes were parsed,
//     so find_best_match is evaluated for every position of the block at once; each lane walks only
//     the in-window part of its bucket.  Work per position is O(#window positions sharing the bigram).
//   * WALK (wave 0): the greedy parse.  Plain steps (literal / match token, the ~97 % case) are chased
//     64 positions at a time out of a register with v_readlane; only RLE / extended-match situations
//     (compressor.c:437-525) run the scalar state machine.  The walk emits a token list, not bits.
//   * EMIT (all threads): token bit strings from the list, a workgroup prefix sum of their lengths,
//     and an MSb-first scatter into an LDS bit buffer that is flushed to HBM.
//   * Tokens that write fewer bytes than they consume (compressor.c:352-358,404-410) break the
//     speculation; the next find_best_match request re-bases the buffer and starts a new epoch.
#pragma once
#include "tamp_common.hpp"

namespace tamp_amd {

constexpr uint32_t kHashBits = 11;
constexpr uint32_t kHashBuckets = 1u << kHashBits;
constexpr uint32_t kRemBits = 16 - kHashBits;  // bigram bits not implied by the bucket number
constexpr uint32_t kRunCap = 128;   // long runs listed per epoch (RUNS builds); further runs stay fully indexed
#ifndef TAMP_LONG_RUN
#define TAMP_LONG_RUN 8
#endif
#ifndef TAMP_DEFER_MIN
#define TAMP_DEFER_MIN 24
#endif
// Extended format, default parse: a position INSIDE a short run of one byte (previous byte, this one and the next equal,
// 2..6 of them ahead) whose bucket lists this many entries or more is not matched in the match phase: blen = kDeferred | slow.
// The walk gets there only when a token happens to end inside the run -- rarely -- and asks Walk::best_on_demand then.
// Wavefront priorities (s_setprio): the kernel is bound by instruction issue, six workgroups share a CU, and what a
// workgroup's wavefronts wait for at a barrier is the slowest of them getting its turn.  The bucket scan -- the bulk, four
// wavefronts wide, no barrier inside -- runs at the lowest priority; the short phases between barriers (load, index, the
// passes behind the scan, jump tables, emit) above it; the serial walk, one wavefront with three waiting for it, on top.
// Same instructions, less waiting: synthetic text 6.66 -> 6.43 ms, prose 13.7 -> 13.2, Python sources 33.4 -> 32.7
// (profiles/ab/r3_persistent_grid.log).
constexpr int kPrioScan = 0, kPrioShort = 2, kPrioWalk = 3;
constexpr uint32_t kDeferMin = TAMP_DEFER_MIN;
constexpr uint32_t kDeferred = 0x1Fu;  // length field of blen: no real first match is longer than the 16-byte ring
constexpr uint32_t kLongRun = TAMP_LONG_RUN;    // a run of one byte this long is listed; its interior leaves the bigram index
// Round 6, late: EIGHT workgroups per CU for the W = 2^10 run-aware build.  The round's loop fits 64 VGPRs without a spill, and
// 1,024 buckets (2 KB of cursors instead of 4; a foreign entry costs six instructions since the one-compare layout) with 128
// explicit pieces per walk segment bring the workgroup to 19,616 B of LDS: synthetic 5.31 -> 4.97 ms, prose / markup / Python
// sources -6 % each (profiles/ab/r6_experiments.log).  TAMP_SEVEN restores the round's earlier shape for A/B runs.
#ifdef TAMP_SEVEN
#define TAMP_WG_PER_CU 7
constexpr uint32_t kSlowCap = 256, kHb1024 = kHashBits;
#else
constexpr uint32_t kSlowCap = 128;             // explicit (non-derivable) token pieces per walk segment
constexpr uint32_t kHb1024 = 10;               // bucket bits of the W = 2^10 run-aware build
#endif

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
    uint32_t first_stream;  // stream index of workgroup 0 (batches above 2^20 streams take several launches)
    uint32_t blk;  // epoch block: positions matched per epoch (multiple of 64)
    uint8_t wbits, lbits, extended, dict_reset, lazy;
    // Segment mode (streaming Compressor over the engine, compressor.c:227-241,728-810): what opens the output, what
    // closes it, and an optional per-stream window state that is read at the start and written back at the end.
    uint8_t nlead;      // leading bytes 0..2: header (+ zero byte with dictionary_reset), FLUSH+pad when appending, none when resuming
    uint16_t lead;      // those bytes, first one in the high byte
    uint8_t seg_flags;  // kSegResume | kSegSave | kSegFlushToken
    uint8_t* state;     // per stream: (1 << wbits) + kSegStateExtra bytes, see kSegStateExtra
    uint32_t* work_counter;    // LOOP builds: next stream index to hand out (zeroed before the launch)
    uint32_t claim;            // LOOP builds: streams a workgroup takes per fetch from the counter
    unsigned long long* prof;  // optional: per-phase cycle sums (debug builds with -DTAMP_PROF)
    uint32_t cut_run;          // epoch cut: a run of this many aligned dwords of one byte ends the block (0 = off)
    uint32_t dbg;              // debug builds only: bit mask of phases to skip (instruction-count experiments)
    // BLOCK builds (round 5: ONE long v1 stream spread over all workgroups, see the template parameter BLOCKM): the "streams"
    // the work counter hands out are the stream's blocks of `blk` positions
    uint32_t* blk_table;            // n_blocks x 16 words, pass 1 writes: entry offset e -> exit offset | bits of the chain << 4
    unsigned long long* blk_info;   // n_blocks, pass 2 writes and pass 3 reads: bit position of the block's first token << 4 | entry offset
    uint32_t block_pass;            // 1 = tables, 3 = walk + emit
    uint32_t n_blocks;
    uint8_t* blk_len;               // optional, n bytes: pass 1 leaves every position's match length here ...
    uint16_t* blk_idx;              // ... and its window index here, and pass 3 reads them instead of matching again
};

// LDS carve-up, shared by the host launcher and the kernel.
struct CompressLds {
    uint32_t ebuf, cnt, ent, blen, bidx, blen2, bidx2, obuf, ctl, runs, runsx, rxset, rbits, total;  // blen2/bidx2: lazy-matching probe results
    uint32_t tokcap, obuf_words, jump, count, vstep;  // jump/count/vstep: byte offsets of the walk's tables inside `ent`
    __host__ __device__ CompressLds(uint32_t W, uint32_t blk, bool packed, bool lazy = false, bool runlist = false, uint32_t hb = kHashBits) {
        uint32_t o = 16;  // slack: the wrapped compare reads up to 15 bytes in front of ebuf (masked out)
        ebuf = o;
        o += align_up(W + blk + kRing + kPendMax + 32, 16);
        cnt = o;  // 2048 x u16 bucket cursors; the walk reuses it for explicit token pieces (256 x 8 B)
        o += (hb < kHashBits ? (1u << hb) : kHashBuckets) * 2;  // (512-bucket builds keep the full region: the walk needs it)
        tokcap = blk + kPendMax + kRing + 80;
        // lazy matching walks (position, state) pairs: two table slots per position and the transitions themselves
        const uint32_t vblk = lazy ? 2 * blk : blk;
        jump = align_up(tokcap * 2, 16);
        count = jump + vblk * 2;  // (default parse: one u32 per position instead, jump | count << 16, at `jump`)
        vstep = count + vblk;
        ent = o;  // (W + blk) index entries (u32 packed, or u16 position-only for the largest windows);
                  // the walk reuses the space for the token list (tokcap x u16)
        {
            // after the match phase the same space holds the token list (tokcap x u16) and the per-position
            // jump tables of the walk: jump target (u16) and token count (u8)
            const uint32_t index_bytes = (W + blk + 16) * (packed ? 4u : 2u);
            const uint32_t walk_bytes = align_up(tokcap * 2, 16) + (lazy ? vblk * 2 + vblk + 16 + vblk + 144 : vblk * 4 + 16);
            o += align_up(index_bytes > walk_bytes ? index_bytes : walk_bytes, 16);
        }
        blen = o;
        o += align_up(blk + 128, 16);
        bidx = o;
        o += align_up(blk * 2, 16);
        blen2 = o;
        if (lazy) o += align_up(blk + 16, 16);
        bidx2 = o;
        if (lazy) o += align_up(blk * 2 + 16, 16);
        obuf_words = ((blk + kPendMax + kRing + 64) * 9 + kSlowCap * 25) / 32 + 8;
        if (obuf_words < 4 + blk / 2) obuf_words = 4 + blk / 2;  // words 4.. hold the u16 scan starts during match
        obuf = o;
        o += align_up(obuf_words * 4, 16);
        ctl = o;
        o += 80 + 64 * 4 + 16;  // 20 control words + 64 sort bins + the 15 prefix codes as a byte table
        // RUNS builds: the long runs of the epoch buffer (start | end << 16) and one bit per buffer position that
        // is left out of the bigram index (the interior of a listed run)
        runs = o;
        if (runlist) o += kRunCap * 4;
        runsx = o;  // per listed run: its byte | the two bytes behind it << 8 (what the second pass needs without touching ebuf)
        if (runlist) o += kRunCap * 4;
        rxset = o;  // one bit per byte value: a listed run of that byte exists in this epoch
        if (runlist) o += 32;
        rbits = o;
        if (runlist) o += align_up((W + blk + 96) / 8, 16);
        total = o;
    }
};

// Match-length prefix code (compressor.c:33-36) packed into immediates: no memory access on the hot path.
constexpr uint64_t pack_bytes(const uint8_t* v, int first, int count) {
    uint64_t r = 0;
    for (int i = 0; i < count; i++) r |= (uint64_t)v[first + i] << (8 * i);
    return r;
}
constexpr uint8_t kCodeTab[15] = {0x00, 0x03, 0x08, 0x0b, 0x14, 0x24, 0x26, 0x2b, 0x4b, 0x54, 0x94, 0x95, 0xaa, 0x27, 0xab};
constexpr uint8_t kNbitsTab[15] = {2, 3, 5, 5, 6, 7, 7, 7, 8, 8, 9, 9, 9, 7, 9};  // incl. the flag bit
constexpr uint64_t pack_nibbles(const uint8_t* v, int count) {
    uint64_t r = 0;
    for (int i = 0; i < count; i++) r |= (uint64_t)v[i] << (4 * i);
    return r;
}
constexpr uint64_t kCodeLo = pack_bytes(kCodeTab, 0, 8), kCodeHi = pack_bytes(kCodeTab, 8, 7);
constexpr uint64_t kNbitsPacked = pack_nibbles(kNbitsTab, 15);
__device__ __forceinline__ uint64_t uni_u64(uint64_t x) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t tok_code(uint32_t i) {
    return (uint32_t)((i < 8 ? kCodeLo >> (8 * i) : kCodeHi >> (8 * (i - 8))) & 0xFF);
}
__device__ __forceinline__ uint32_t tok_nbits(uint32_t i) { return (uint32_t)(kNbitsPacked >> (4 * i)) & 15; }

// 16-bit bijective mix of a bigram: top kHashBits select the bucket, the rest ride in the entry.  Any odd multiplier is a
// bijection mod 2^16, so exactness does not depend on it -- only how many FOREIGN bigrams share a query's bucket.  Round 4
// (tools/hash_search.py: the brackets replayed on the host for all 32,768 odd multipliers, synthetic text + both frozen
// corpora): 40503 (round 1's golden-ratio constant) scans 8.00 entries per query on the synthetic text where a
// collision-free key would scan 7.34; 46437 scans 7.38 (prose 11.29 -> 10.93 of 10.72) -- 7 % fewer lock-step iterations.
#ifndef TAMP_MIX_MUL
#define TAMP_MIX_MUL 46437u
#endif
static_assert((TAMP_MIX_MUL & 1u) == 1u && TAMP_MIX_MUL < 65536u, "odd 16-bit multiplier: a bijection on bigrams");
// (1,024 buckets -- the W = 2^10 build since the end of round 6 -- have their own best multiplier: tools/hash_search.py with HB=10
// scans 7.53 entries per query of the synthetic text with 5455 where 46437 scans 8.03; prose 11.19 / 11.54, markup 10.72 / 11.30)
#ifndef TAMP_MIX_MUL10
#define TAMP_MIX_MUL10 5455u
#endif
static_assert((TAMP_MIX_MUL10 & 1u) == 1u && TAMP_MIX_MUL10 < 65536u, "odd 16-bit multiplier: a bijection on bigrams");
template <uint32_t HBITS = kHashBits>
__device__ __forceinline__ uint32_t mix16(uint32_t pair16) { return (pair16 * (HBITS == 10 ? TAMP_MIX_MUL10 : TAMP_MIX_MUL)) & 0xFFFFu; }
// entry payload from 4 little-endian bytes b0..b3 at a position: rem | b2 | low bits of b3, in bits 16..31
template <uint32_t REM = kRemBits>
__device__ __forceinline__ uint32_t entry_payload(uint32_t bytes4, uint32_t mix) {
    return ((mix & ((1u << REM) - 1)) << 16) | (((bytes4 >> 16) & 0xFFu) << (16 + REM)) |
           ((bytes4 >> 24) << (24 + REM));
}

// Length (0..16) of the common prefix of ebuf[c..c+16) and the pattern dwords P[0..3].  Branch-free: 64 lanes in
// lockstep would walk every branch of a staged compare anyway, so all 16 bytes are fetched (five aligned dwords,
// funnel-shifted by the byte phase) and the first differing byte is found with two 64-bit count-trailing-zeros.
// v_ffbl_b32 as the hardware defines it: 0xFFFFFFFF for a zero operand (__builtin_ctz leaves that case undefined)
__device__ __forceinline__ uint32_t ffbl_or_ones(uint32_t x) {
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t prefix_len16(const uint8_t* ebuf, uint32_t c, const uint32_t (&P)[4]) {
#ifndef TAMP_CMP_ALIGNED
    // Round 6: ONE hardware-unaligned ds_read_b128 (gfx950's LDS serves it) instead of five aligned dwords + four funnel shifts:
    // with the rest of the loop slimmed down the LDS pipe has the room (synthetic 5.64 -> 5.51 ms; in rounds 1-2 it was the bound:
    // +7 %).  profiles/ab/r6_experiments.log
    const LdsU128 v = *reinterpret_cast<const LdsU128*>(ebuf + c);
    const uint32_t x0 = v.x ^ P[0], x1 = v.y ^ P[1], x2 = v.z ^ P[2], x3 = v.w ^ P[3];
#else
    const uint32_t* w = reinterpret_cast<const uint32_t*>(ebuf + (c & ~3u));
    const uint32_t sh = c & 3u;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sh) ^ P[0];
    const uint32_t x1 = __builtin_amdgcn_alignbyte(w2, w1, sh) ^ P[1];
    const uint32_t x2 = __builtin_amdgcn_alignbyte(w3, w2, sh) ^ P[2];
    const uint32_t x3 = __builtin_amdgcn_alignbyte(w4, w3, sh) ^ P[3];
#endif
#ifndef TAMP_CMP_STAGED
    // Round 6: no branch at all.  First set bit of each difference dword (all ones when it has none), the dword's bit offset
    // added with unsigned saturation (v_add_u32 ... clamp keeps "none" at all ones), the smallest of the four: 4 ffbl + 3 add +
    // min3 + min + shift = 10 instructions behind the xors where the staged form ran 6 + 13 in nearly every lock-step
    // iteration (some lane's first eight bytes agree) with two exec-mask regions: synthetic 5.96 -> 5.81 ms, prose 9.96 -> 9.63,
    // markup 9.88 -> 9.53, Python sources 22.24 -> 21.74 (profiles/ab/r6_experiments.log).
    const uint32_t t0 = ffbl_or_ones(x0);
    const uint32_t t1 = __builtin_elementwise_add_sat(ffbl_or_ones(x1), 32u);
    const uint32_t t2 = __builtin_elementwise_add_sat(ffbl_or_ones(x2), 64u);
    const uint32_t t3 = __builtin_elementwise_add_sat(ffbl_or_ones(x3), 96u);
    return min(min(min(t0, t1), min(t2, t3)) >> 3, 16u);
#endif
    const uint64_t lo = (uint64_t)x0 | ((uint64_t)x1 << 32), hi = (uint64_t)x2 | ((uint64_t)x3 << 32);
    const uint32_t nlo = (uint32_t)__builtin_ctzll(lo | (1ull << 63)) >> 3;          // 0..7 (7 also when lo == 0)
    const uint32_t nhi = 8u + ((uint32_t)__builtin_ctzll(hi | (1ull << 63)) >> 3);   // 8..15
    return lo ? nlo : (hi ? nhi : 16u);
}

// Candidate whose bytes run past the newest window byte: the ring continues with the OLDEST window byte, i.e.
// candidate byte k is ebuf[c + k] for k < t and ebuf[c - W + k] for k >= t, with t = q + W - c in 1..15.
// Both halves are fetched as 16 unaligned bytes and blended; returns the common prefix length with P (0..16).
__device__ __forceinline__ uint32_t prefix_len_wrapped16(const uint8_t* ebuf, uint32_t c, uint32_t t, uint32_t W,
                                                         const uint32_t (&P)[4]) {
    const uint32_t* wa = reinterpret_cast<const uint32_t*>(ebuf + (c & ~3u));
    const uint32_t sa = c & 3u;
    const int32_t ob = (int32_t)c - (int32_t)W;  // >= -15: the slack in front of ebuf keeps this inside LDS
    const uint32_t* wb = reinterpret_cast<const uint32_t*>(ebuf + (ob & ~3));
    const uint32_t sb = (uint32_t)ob & 3u;
    uint32_t res = 16;
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        const uint32_t xa = __builtin_amdgcn_alignbyte(wa[j + 1], wa[j], sa);
        const uint32_t xb = __builtin_amdgcn_alignbyte(wb[j + 1], wb[j], sb);
        const uint32_t lo = 4u * (uint32_t)j;  // bytes lo..lo+3 of the candidate
        const uint32_t m = t >= lo + 4 ? 0xFFFFFFFFu : (t <= lo ? 0u : (1u << (8 * (t - lo))) - 1u);
        const uint32_t x = ((xa & m) | (xb & ~m)) ^ P[j];
        if (x) res = lo + ((uint32_t)__builtin_ctz(x) >> 3);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// Walk state: registers of wave 0, identical in every lane (lane 0 performs the LDS stores).
// ---------------------------------------------------------------------------------------------
struct Walk {
    uint8_t* ebuf;
    const uint8_t* blen;
    const uint16_t* bidx;
    uint16_t* toklist;  // token list: q (derivable token at input position q) or 0x8000|k (explicit piece k)
    uint32_t* stok;     // explicit pieces: stok[2k] = bits, stok[2k+1] = bit count
    uint32_t W, mask, wbits, lbits, minp;
    bool ext;
    uint32_t wp_e;    // window_pos at epoch start
    uint32_t wr, rd;  // bytes written / consumed since epoch start
    uint32_t nvalid;
    uint32_t rle_count, ext_count, ext_pos;
    bool ext_resolved;
    bool last_ext_direct = false;  // (instrumented builds)
    uint32_t dbg_lag_rle = 0, dbg_lag_ext = 0, dbg_lag_rle_short = 0;  // (instrumented builds)
    uint32_t ntok, ns;
    bool partial = false;  // kSegPartial: every slow step is ONE poll with a full 16-byte ring (no whole-run / whole-match shortcuts)
    uint32_t dbg = 0;      // (instrumented builds: 0x400000 / 0x800000 skip the two searches -- wrong bytes, the time they cost)
    // Cooperative searches (256-thread workgroups): the walk is ONE wavefront and the other three wait for it at a barrier;
    // a search is posted in the control words, every wavefront takes a quarter of the candidate rounds, and the four keys
    // are combined by the walking one (kCoop* below; two s_barrier per search).
    typedef __attribute__((address_space(3))) volatile uint32_t CtlWord;
    CtlWord* ctlw = nullptr;
    bool coop = false;
    bool lazy;           // lazy matching (compressor.c:576-619)
    bool lazy_valid;     // a match cached by the previous step's probe
    uint32_t lazy_idx, lazy_len;
    const uint8_t* blen2;   // probe results: best match of the pattern at q+1 in the window as it is at q
    const uint16_t* bidx2;
    int lane;

    // The walk's scalars are identical in all 64 lanes; values that come back from LDS are passed through
    // v_readfirstlane so the compiler keeps the whole state machine on the scalar unit (s_cbranch_scc, no
    // exec-mask juggling).
    static __device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    __device__ __forceinline__ uint32_t wp() const { return (wp_e + wr) & mask; }
    // byte at window index i of the live window ebuf[wr .. wr+W)
    __device__ __forceinline__ uint32_t win(uint32_t i) const { return uni(ebuf[wr + ((i - wp()) & mask)]); }
    __device__ __forceinline__ uint32_t inb(uint32_t k) const { return uni(ebuf[W + rd + k]); }
    // per-lane variants for the cooperative search
    __device__ __forceinline__ uint32_t win_l(uint32_t i) const { return ebuf[wr + ((i - wp()) & mask)]; }
    __device__ __forceinline__ uint32_t inb_l(uint32_t k) const { return ebuf[W + rd + k]; }
    // Common prefix (<= lim) of the window from index i with either the window from index j (other_is_window) or the
    // ring from byte j.  Dword compares while the window ranges stay on one side of the write cursor (contiguous in
    // ebuf), byte compares for the remainder.
    __device__ uint32_t common_l(uint32_t i, uint32_t j, uint32_t lim, bool other_is_window) const {
        const uint32_t ri = (i - wp()) & mask, rj = (j - wp()) & mask;
        const uint32_t a = wr + ri, b = other_is_window ? wr + rj : W + rd + j;
        uint32_t contig = W - ri;  // bytes before index i's range passes the newest window byte
        if (other_is_window) contig = min(contig, W - rj);
        contig = min(contig, lim);
        uint32_t nn = 0;
        while (nn + 8 <= contig) {  // eight bytes per LDS round trip (extended matches run to 100+ bytes; sixteen spill a VGPR)
            const uint32_t x0 = lds_u32_unaligned(ebuf, a + nn) ^ lds_u32_unaligned(ebuf, b + nn);
            const uint32_t x1 = lds_u32_unaligned(ebuf, a + nn + 4) ^ lds_u32_unaligned(ebuf, b + nn + 4);
            if (x0 | x1) return nn + (x0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : 4 + ((uint32_t)__builtin_ctz(x1) >> 3));
            nn += 8;
        }
        while (nn + 4 <= contig) {
            const uint32_t x = lds_u32_unaligned(ebuf, a + nn) ^ lds_u32_unaligned(ebuf, b + nn);
            if (x) return nn + ((uint32_t)__builtin_ctz(x) >> 3);
            nn += 4;
        }
        while (nn < lim && win_l(i + nn) == (other_is_window ? win_l(j + nn) : inb_l(j + nn))) nn++;
        return nn;
    }

    // write_to_bit_buffer (compressor.c:49-52) becomes "append an explicit piece to the token list"
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb) {
        // (opaque to the optimiser: otherwise the constant (code, length) pairs of the rare tokens are hoisted out of
        // the walk into VGPR pairs that live -- and spill -- for the whole kernel)
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

    // Append `cnt` bytes to the history.  `clean` = they are exactly the input bytes already sitting there.
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
        put((tok_code(ci) << trailing) | (value & ((1u << trailing) - 1)), (tok_nbits(ci) - 1) + trailing);
    }

    // write_rle_token (compressor.c:342-359); the run's `count` bytes are already consumed.
    __device__ void emit_rle(uint32_t count) {
        const uint32_t sym = win((wp() - 1) & mask);
        put(tok_code(kSymRle), tok_nbits(kSymRle));
        put_exthuff(count - 2, 4);
        uint32_t w = min(min(count, kRleWindowMax), W - wp());
        const bool clean = (wr + count == rd) && (w == count);
#ifdef TAMP_PROF
        if (w < count) { dbg_lag_rle++; if (count < 24) dbg_lag_rle_short++; }
#endif
        append(w, clean, [&](uint32_t) { return sym; });
    }

    // write_extended_match_token (compressor.c:377-415)
    __device__ void emit_ext() {
        const uint32_t count = ext_count, pos = ext_pos;
        put(tok_code(kSymExt), tok_nbits(kSymExt));
        put_exthuff(count - minp - 12, 3);
        put(pos, wbits);
        uint32_t w = min(count, W - wp());
        const bool clean = (wr + count == rd) && (w == count);
#ifdef TAMP_PROF
        if (w < count) dbg_lag_ext++;
#endif
        const uint32_t wr0 = wr, wp0 = wp();
        // Sources are read in the pre-token window; appended bytes land beyond it (the memmove
        // semantics of tamp_window_copy, common.c:58-86, for free).
        append(w, clean, [&](uint32_t i) { return uni(ebuf[wr0 + ((pos + i - wp0) & mask)]); });
        ext_count = 0;
    }

    // find_extended_match (compressor.c:297-333), candidates spread over the 64 lanes.  `avail` = input bytes the
    // search may look at.  The reference searches once per 16-byte ring refill (cap = count + ring); every round
    // keeps the lowest candidate among the longest and only candidates at or above it stay alive, so the rounds
    // together select "longest common prefix with the input, capped by the window end and min+131; ties -> lowest
    // index" among the candidates >= the first match -- which one search with the whole look-ahead finds directly.
    // Candidate rounds [4 * g0, 4 * g1) of the search (a round = one candidate per lane, 64 apart): the best key of this
    // wavefront's lanes.  key = length << 16 | ~candidate: longest, ties -> lowest candidate, whatever the order of evaluation.
    // Round 5: the filter looks at the candidate's FIRST four bytes as well as at the four around the current end (a 4-byte
    // filter alone passes two to five false candidates per search in 1 KiB of text, and each cost every lane of its
    // wavefront a serial byte-by-byte verification), and whatever passes is compared by the whole wavefront at once -- four
    // bytes a lane against pattern + look-ahead, ring order on the window's side, one ballot -- instead of by its own lane in
    // a loop: 1.2 k -> ~0.5 k VALU wave-instructions per search over the four wavefronts (profiles/r5_phase_valu.csv: the
    // walk was 14 / 27 / 18 % of all instructions on prose / markup / Python sources, nearly all of it searches).
    __device__ __forceinline__ uint32_t ext_search_rounds(uint32_t avail, uint32_t g0, uint32_t g1) const {
        const uint32_t pos = ext_pos, cnt = ext_count;
        const uint32_t maxp = min(cnt + avail, minp + 11 + kExtExtraMax);
        // (the consumed bytes ARE the pattern: input [rd-cnt, rd) == window[pos, pos+cnt))
        const uint32_t tail4 = uni(lds_u32_unaligned(ebuf, W + rd - 3));   // bytes cnt-3 .. cnt: the last three consumed + the next input byte
        const uint32_t head4 = uni(lds_u32_unaligned(ebuf, W + rd - cnt));
        const uint32_t nextb = tail4 >> 24;
        const uint32_t wpv = wp();
        const uint32_t inp = lds_u32_unaligned(ebuf, W + rd - cnt + 4u * (uint32_t)lane);  // pattern + look-ahead, four bytes a lane
        uint32_t key = 0;  // wave-uniform
        for (uint32_t cb0 = pos; cb0 + cnt + 1 <= W; cb0 += 16 * kWave) {
            for (uint32_t g = g0; g < g1; g++) {
                if (4 * g * kWave >= W) break;  // (2^8 / 2^9 windows: 4 / 8 candidates per lane cover the window)
                if (cb0 + 4 * g * kWave + cnt + 1 > W) break;  // (no candidate of this and the later rounds is in range)
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t cb = cb0 + (4 * g + j) * kWave;  // lane 0's candidate of this step
                    const uint32_t c = cb + (uint32_t)lane;
                    const bool valid = c + cnt + 1 <= W;
                    const uint32_t r0 = ((valid ? c : pos) - wpv) & mask;  // oldest-first offset of the candidate's first byte
                    const uint32_t r = (r0 + cnt - 3) & mask;
                    bool hit = lds_u32_unaligned(ebuf, wr + r) == tail4;
                    if (r > W - 4) hit = ebuf[wr + ((r + 3) & mask)] == nextb;  // the four bytes straddle the write cursor
                    bool hit0 = lds_u32_unaligned(ebuf, wr + r0) == head4;
                    if (r0 > W - 4) hit0 = true;                                // (likewise: left to the compare)
                    uint64_t bal = __ballot(valid && hit && hit0);
                    while (bal) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(bal);
                        bal &= bal - 1;
                        const uint32_t ci = cb + f;                      // window index of the candidate (uniform)
                        const uint32_t rc = (ci - wpv) & mask;
                        uint32_t o = rc + 4u * (uint32_t)lane;
                        uint32_t cand;
                        if (o + 3 >= W && o < W) {  // (at most one lane: its four bytes straddle the newest byte)
                            cand = 0;
                            for (uint32_t b = 0; b < 4; b++) cand |= (uint32_t)ebuf[wr + ((o + b) & mask)] << (8 * b);
                        } else {
                            if (o >= W) o -= W;
                            cand = lds_u32_unaligned(ebuf, wr + o);
                        }
                        const uint32_t x = cand ^ inp;
                        const uint64_t diff = __ballot(x != 0);
                        uint32_t lcp = 256;
                        if (diff) {
                            const uint32_t fl = (uint32_t)__builtin_ctzll(diff);
                            const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)fl);
                            lcp = 4u * fl + ((uint32_t)__builtin_ctz(xf) >> 3);
                        }
                        const uint32_t len = min(lcp, min(maxp, W - ci));
                        if (len >= cnt + 1) key = max(key, (len << 16) | (0xFFFFu - ci));
                    }
                }
            }
        }
        return uni(key);
    }

    enum : uint32_t { kCoopCmd = 17, kCoopA = 1, kCoopB = 2, kCoopC = 3, kCoopD = 4, kCoopE = 18, kCoopF = 19, kCoopKey = 8 };  // ctl words (free during the walk)
    enum : uint32_t { kCmdEnd = 0, kCmdExtSearch = 1, kCmdBest = 2 };

    // find_extended_match (compressor.c:297-333), candidates spread over the 64 lanes -- of all four wavefronts when the
    // workgroup has them.  `avail` = input bytes the search may look at.  The reference searches once per 16-byte ring
    // refill (cap = count + ring); every round keeps the lowest candidate among the longest and only candidates at or
    // above it stay alive, so the rounds together select "longest common prefix with the input, capped by the window end
    // and min+131; ties -> lowest index" among the candidates >= the first match -- which one search with the whole
    // look-ahead finds directly.
    __device__ void ext_search(uint32_t avail, uint32_t& npos, uint32_t& ncnt) {
#ifdef TAMP_PROF
        if (dbg & 0x400000u) { npos = ext_pos, ncnt = ext_count; return; }
#endif
        uint32_t key;
        if (coop) {
            if (lane == 0) {
                ctlw[kCoopA] = ext_pos, ctlw[kCoopB] = ext_count, ctlw[kCoopC] = avail, ctlw[kCoopD] = wr, ctlw[kCoopE] = rd, ctlw[kCoopF] = wp_e;
                ctlw[kCoopCmd] = kCmdExtSearch;
            }
            __syncthreads();  // the request is posted: the helpers (Walk::serve) start on rounds 4..15
            key = ext_search_rounds(avail, 0, 1);
            __syncthreads();  // their keys are in
            key = max(max(key, uni(ctlw[kCoopKey + 1])), max(uni(ctlw[kCoopKey + 2]), uni(ctlw[kCoopKey + 3])));
        } else {
            key = ext_search_rounds(avail, 0, 4);
        }
        ncnt = key >> 16;
        npos = 0xFFFFu - (key & 0xFFFFu);
    }

    // Wavefronts 1-3 while wavefront 0 walks: wait for a request, take their quarter, hand the key back; leave when the
    // walk posts kCmdEnd.  Every request is two barriers for all four wavefronts, the end is one.
    __device__ void serve(uint32_t wave) {
        for (;;) {
            __syncthreads();
            const uint32_t cmd = uni(ctlw[kCoopCmd]);
            if (cmd == kCmdEnd) return;
            const uint32_t avail = uni(ctlw[kCoopC]);  // (kCmdBest: the ring's bytes)
            wr = uni(ctlw[kCoopD]), rd = uni(ctlw[kCoopE]), wp_e = uni(ctlw[kCoopF]);
            uint32_t key;
            if (cmd == kCmdBest) {
                key = best_rounds(avail, wave, wave + 1);
            } else {
                ext_pos = uni(ctlw[kCoopA]), ext_count = uni(ctlw[kCoopB]);
                key = ext_search_rounds(avail, wave, wave + 1);
            }
            if (lane == 0) ctlw[kCoopKey + wave] = key;
            __syncthreads();
        }
    }

    enum { kStepOk = 0, kStepRebase = 1, kStepExcess = 2 };

    // find_best_match (compressor_find_match_desktop.c) over the live window, candidates spread over the 64 lanes: for the
    // positions the match phase left out (kDeferred).  Window index i = history offset r + window_pos; a match may not run
    // past index W-1; longest, ties -> lowest index; the ring goes on with the oldest byte behind the newest (common_l).
    // The result goes where the match phase would have put it (the position stays a slow one): called by the walk on
    // arrival, in front of the step that may consult it.
    // candidate rounds [4 * g0, 4 * g1) of that search: the best key (length << 16 | W - index) of this wavefront's lanes
    __device__ __forceinline__ uint32_t best_rounds(uint32_t R, uint32_t g0, uint32_t g1) const {
        const uint32_t maxp1 = ext ? minp + 11 + kExtExtraMax : minp + 13;
        const uint32_t cap = min(R, maxp1);
        const uint32_t wpv = wp();
        // clean state (wr == rd): the buffer looks like the match phase's view of position q = wr -- window ebuf[q, q+W),
        // input behind it -- so a candidate is compared the way the bucket scan compares one (16 bytes at once)
        const uint32_t q = wr;
        uint32_t P[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) P[jj] = uni(lds_u32_unaligned(ebuf, W + q + 4 * jj));
        const uint32_t b01 = P[0] & 0xFFFFu;
        uint32_t key = 0;
        for (uint32_t r0 = (uint32_t)lane; r0 < W; r0 += 16 * kWave) {
            for (uint32_t g = g0; g < g1; g++) {
                if (4 * g * kWave >= W) break;
                uint32_t hits = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t r = r0 + (4 * g + j) * kWave;
                    const bool valid = r < W;
                    // (offset W-1, the newest byte, pairs with the oldest one: left to the compare)
                    const bool hit = (lds_u32_unaligned(ebuf, q + (valid ? r : 0u)) & 0xFFFFu) == b01 || r == W - 1;
                    hits |= (uint32_t)(valid && hit) << j;
                }
                while (hits) {
                    const uint32_t j = (uint32_t)__builtin_ctz(hits);
                    hits &= hits - 1;
                    const uint32_t r = r0 + (4 * g + j) * kWave, t = W - r;
                    const uint32_t i = (r + wpv) & mask, lim = W - i;
                    const uint32_t lw = t < 16 ? prefix_len_wrapped16(ebuf, q + r, t, W, P) : prefix_len16(ebuf, q + r, P);
                    const uint32_t l = min(lw, min(cap, lim));
                    if (l >= 2) key = max(key, (l << 16) | lim);
                }
            }
        }
        return wave_max_u32(key);
    }

    __device__ void best_on_demand(uint32_t R) {
        uint32_t idx, len;
#ifdef TAMP_PROF
        if (dbg & 0x800000u) {
            if (lane == 0) const_cast<uint8_t*>(blen)[rd] = 0x80u, const_cast<uint16_t*>(bidx)[rd] = 0;
            __builtin_amdgcn_wave_barrier();
            return;
        }
#endif
        uint32_t key;
        if (coop) {
            if (lane == 0) {
                ctlw[kCoopC] = R, ctlw[kCoopD] = wr, ctlw[kCoopE] = rd, ctlw[kCoopF] = wp_e;
                ctlw[kCoopCmd] = kCmdBest;
            }
            __syncthreads();
            key = best_rounds(R, 0, 1);
            __syncthreads();
            key = max(max(key, uni(ctlw[kCoopKey + 1])), max(uni(ctlw[kCoopKey + 2]), uni(ctlw[kCoopKey + 3])));
        } else {
            key = best_rounds(R, 0, 4);
        }
        len = key >> 16;
        idx = key ? W - (key & 0xFFFFu) : 0u;
        if (lane == 0) {
            const_cast<uint8_t*>(blen)[rd] = (uint8_t)(len | 0x80u);
            const_cast<uint16_t*>(bidx)[rd] = (uint16_t)idx;
        }
        __builtin_amdgcn_wave_barrier();
    }

    template <bool RB>
    __device__ __forceinline__ bool best(uint32_t& idx, uint32_t& len) {
        if (wr != rd || rd >= nvalid) return false;
        const uint32_t sv = uni(blen[rd]);
        len = sv & 0x1Fu;
        idx = uni(bidx[rd]);
        if constexpr (RB) ext_resolved = (sv & 0x40u) != 0 && !partial;  // idx is the final index of the extended match that starts here
        return true;
    }

    // One parse step = tamp_compressor_poll (compressor.c:532-660) with the ring = next R input bytes.
    // Returns kStepRebase *before mutating anything* when it needs a find_best_match result that the
    // current epoch cannot supply.
    template <bool RB>
    __device__ int step(uint32_t R, uint32_t left) {
        uint32_t idx = 0, len = 0;
        if (ext) {
            if (ext_count) {  // compressor.c:439-468, all polls of the continuation at once (see ext_search)
                lazy_valid = false;  // extended handling consumes input: any cached lazy match is stale (:563-568)
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
                        uint32_t extra = ncnt - ext_count;
                        ext_pos = npos;
                        ext_count = ncnt;
                        rd += extra;
                        left -= extra;
                        // No candidate matched more than ncnt bytes of (pattern + input).  Unless the cap stopped
                        // it (min+131 reached -> emitted at the top of the loop; input exhausted -> the flush emits
                        // it), the reference's next search finds nothing and emits the token; skip straight to that.
                        if (ncnt == reach) continue;
                    }
                    emit_ext();
                    return kStepOk;
                }
                return kStepOk;
            }
            // RLE accumulation, compressor.c:470-525
            const uint32_t last = win((wp() - 1) & mask);
            uint32_t avail = 0;
            {   // Leading input bytes equal to `last`, the whole run at once (four bytes per lane, one ballot).  The
                // reference takes a run one 16-byte ring at a time and carries `rle_count` from poll to poll; nothing
                // else happens between those polls, so the count it arrives at is "run length, capped at 241 and at
                // the end of the input" -- the run's bytes all lie inside the loaded look-ahead (16 + 256 bytes).
                const uint32_t x = lds_u32_unaligned(ebuf, W + rd + 4u * (uint32_t)lane) ^ (last * 0x01010101u);
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
                rd += avail;
                lazy_valid = false;
                return kStepOk;
            }
            if (total >= 2) {
                bool use_pattern = false;
                if (total == avail && total <= 6) {
                    if (!best<RB>(idx, len)) return kStepRebase;
                    if (len > total)
                        use_pattern = true;
                    else
                        len = 0;
                }
                if (!use_pattern) {
                    rd += avail;
                    emit_rle(total);
                    rle_count = 0;
                    lazy_valid = false;
                    return kStepOk;
                }
            } else if (rle_count == 1) {
                put((1u << lbits) | last, lbits + 1);
                append(1, wr + 1 == rd, [&](uint32_t) { return last; });
                rle_count = 0;
                lazy_valid = false;
                return kStepOk;
            }
        }
        if (lazy) {  // compressor.c:576-619
            uint32_t cidx = lazy_idx, clen = lazy_len;
            const bool from_cache = lazy_valid;
            if (!from_cache) {
                cidx = idx, clen = len;
                if (clen == 0 && !best<RB>(cidx, clen)) return kStepRebase;
            }
            bool defer = false;
            uint32_t nidx = 0, nlen = 0;
            if (clen >= minp && clen <= 8 && R > clen + 2) {
                // probe position+1 against the window as it is now: precomputed next to the match tables
                if (wr != rd || rd + 1 >= nvalid) return kStepRebase;  // nothing has been changed yet
                nlen = uni(blen2[rd]);
                nidx = uni(bidx2[rd]);
                const uint32_t wpos = wp();
                defer = nlen > clen && (wpos < nidx || wpos >= nidx + nlen);  // validate_no_match_overlap, :185-188
            }
            idx = cidx, len = clen;
            lazy_valid = false;
            if (defer) {
                lazy_valid = true, lazy_idx = nidx, lazy_len = nlen;
                len = 0;  // literal now, the better match at the next position
            }
        } else if (len == 0 && !best<RB>(idx, len)) {
            return kStepRebase;
        }

        if (len < minp) {  // literal, compressor.c:625-632
            const uint32_t c = inb(0);
            if (c >> lbits) return kStepExcess;
            put((1u << lbits) | c, lbits + 1);
            len = 1;
        } else {
            if (ext && len > minp + 11) {  // compressor.c:636-644
                if (RB && ext_resolved) {
                    // the match phase saw no other candidate that could carry the continuation: count the common
                    // prefix of that window position and the input (four bytes per lane) and emit -- unless the match
                    // reaches the newest window byte, where the ring goes on with the oldest one (search instead)
                    ext_resolved = false;
                    const uint32_t off = (idx - wp()) & mask, t0 = W - off;
                    const uint32_t xx = lds_u32_unaligned(ebuf, wr + off + 4u * (uint32_t)lane) ^ lds_u32_unaligned(ebuf, W + rd + 4u * (uint32_t)lane);
                    const uint64_t bal = __ballot(xx != 0);
                    uint32_t lcp = 256;
                    if (bal) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(bal);
                        const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)xx, (int)f);
                        lcp = 4u * f + ((uint32_t)__builtin_ctz(xf) >> 3);
                    }
                    const uint32_t cnt = min(min(lcp, W - idx), min(minp + 11 + kExtExtraMax, left));
                    if (cnt < t0 && cnt >= len) {
                        last_ext_direct = true;
                        ext_pos = idx;
                        ext_count = cnt;
                        rd += cnt;
                        emit_ext();
                        return kStepOk;
                    }
                }
                ext_count = len;
                ext_pos = idx;
                rd += len;
                return kStepOk;
            }
            put((tok_code(len - minp) << wbits) | idx, tok_nbits(len - minp) + wbits);
        }
        // compressor.c:651-657: the consumed bytes enter the window
        const uint32_t rd0 = rd;
        rd += len;
        append(len, wr + len == rd, [&](uint32_t i) { return uni(ebuf[W + rd0 + i]); });
        return kStepOk;
    }
};

// Workgroups per CU the register allocation of the run-aware builds aims at.  Round 4: SEVEN (72 VGPRs, 5-7 of them spilled
// to scratch outside the loops) with 1,024-position blocks (21.4 KB of LDS at w = 10) instead of six (80 VGPRs) with 1,536
// (26.3 KB): a 4 KiB stream takes four epochs instead of three and the kernel is still 0.7 % faster on the synthetic text;
// real text, whose workgroups spend a third of their time in the one-wavefront walk, gains 8-10 % from the seventh
// (profiles/ab/r4_seven_workgroups_per_cu.log).  Eight (64 VGPRs) spills 92 registers.
#ifndef TAMP_WG_PER_CU
#define TAMP_WG_PER_CU 8
#endif
#ifndef TAMP_LEAN_PER_CU  // (workgroups of 256 threads per CU the lean / lazy builds' registers are budgeted for: 6 -> 80 VGPRs, 5 -> 96)
#define TAMP_LEAN_PER_CU 6
#endif
#ifndef TAMP_LAZY_PER_CU
#define TAMP_LAZY_PER_CU 5
#endif
// Block size after a break (a token that wrote fewer bytes than it consumed threw the rest of the block away): with the
// 1,536-position blocks of rounds 1-3 halving it (512 at least) was worth 15 % on real text; with 1,024-position blocks
// at seven workgroups per CU the full block is the better guess again (prose 11.55 -> 10.9 ms, Python sources 28.3 -> 26.8,
// profiles/ab/r4_seven_workgroups_per_cu.log).  Tuning builds override.
#ifndef TAMP_ALIGN_MIN  // shortest block the ring-end alignment may produce (tuning builds override; 65536 = off)
#define TAMP_ALIGN_MIN 256
#endif
#ifndef TAMP_BRK_SHIFT
#define TAMP_BRK_SHIFT 0
#define TAMP_BRK_MIN 512
#endif
enum : uint32_t { kActDone = 1, kActRebase = 2, kActContinue = 3 };
enum : uint8_t { kSegResume = 1, kSegSave = 2, kSegFlushToken = 4, kSegPartial = 8 };
// Per-stream state slot of the segment calls: the window in ring order, then
//   [W] u16 window_pos   [W+2] u8 FLUSH token written (out)   [W+3] u8 rle_count   [W+4] u8 extended-match count
//   [W+5] u8 pending output bits (in: 0..31, a reference object may sit on a whole token; out: 0..7)
//   [W+6] u16 extended-match window position   [W+9] u32 input bytes parsed (out, unaligned)
//   [W+16] u32 the pending bits, left aligned (first bit in bit 31; unaligned)
// The fields behind the token flag are what a TampCompressor carries between two calls that are NOT separated by a
// flush (compressor.h:13-66: rle_count, extended_match_count / _position, the bit buffer); kSegPartial ends the launch
// the way tamp_compressor_compress_cb ends a call (compressor.c:681-722): parse steps only while 16 bytes of look-ahead
// are there, no drain.  The bytes a pending run / extended match has consumed, and the unparsed tail of the previous
// call, come back IN FRONT of the next call's input (the host shim puts them there).
constexpr uint32_t kSegStateExtra = 24;
// ctl words
enum : uint32_t { cAct = 0, cShift = 1, cP0 = 2, cPending = 3, cWp = 4, cNtok = 5, cExcess = 6, cBlk = 7, cWave = 8, cNruns = 12, cQuad = 13, cNext = 14, cCut = 15, cCutThr = 16 };

// Instrumented builds (-DTAMP_PROF): a section whose effect does not change when it runs twice can be repeated per bit of
// CompressArgs::dbg -- 0x100 bucket loop, 0x200 wrap-zone resolution, 0x10000 load, 0x20000 index, 0x40000 jump tables,
// 0x80000 emit -- and the difference of two `rocprofv3 --pmc SQ_INSTS_VALU` runs is that section's exact instruction
// count over all epochs of all streams (tools/phase_valu.sh -> profiles/r4_phase_valu.csv).
#ifdef TAMP_PROF
#define TAMP_REPEAT(bit) for (uint32_t _rep = 0; _rep < ((a.dbg & (bit)) ? 2u : 1u); _rep++)
#else
#define TAMP_REPEAT(bit)
#endif
#ifdef TAMP_PROF
#define TAMP_PROF_MARK(i)                                     \
    do {                                                      \
        __syncthreads();                                      \
        unsigned long long _n = __builtin_readcyclecounter(); \
        pt[i] += _n - pc;                                     \
        pc = _n;                                              \
        if (a.dbg & (8u << (i))) return; /* truncation experiments: instruction counts up to this mark */ \
    } while (0)
#else
#define TAMP_PROF_MARK(i) \
    do {                  \
    } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------
// PACKED: u32 index entries (position | rest of bigram | next byte | 3 bits of the one after); otherwise u16
// positions only (window 2^15, where packed entries would not fit in 160 KiB of LDS).
// LAZY: lazy matching (compressor.c:576-619) compiled in; the default build carries none of its code.
// WSCAN != 0: the window size as a compile-time constant, used by the bucket-scan loop ONLY (the launcher passes it for
// the common 2^10 window).  The loop's range / limit constants then are immediates instead of scalar registers; with
// them in registers the allocator, short of scalar registers everywhere in this kernel, sometimes reloads a spilled one
// inside that loop (v_readlane + s_nop per entry: 4 % of the kernel, the "same code, 4 % slower" builds of section 3.6
// of DESIGN.md).  Making W a constant for the whole kernel lets the compiler unroll and hoist elsewhere and costs 50+
// spilled VGPRs, hence the narrow use.
// LOOP: persistent workgroups that take stream after stream from a counter (the launcher starts as many as the device
// holds) instead of one workgroup per stream.  Round 3: the hardware deals the workgroups of a grid to the eight XCDs round
// robin by index -- a STATIC eighth of the batch each -- so with streams of unequal cost (real text) the XCDs finish up to
// a fifth apart; with the counter an XCD that is ahead simply takes more streams: prose +27 %, Python sources +11 % at
// 65,536 x 4 KiB, synthetic text -1 % (the loop keeps a few more values alive).  Short messages cost the same all over and
// two million fetches from one address were a bottleneck: their build stays one workgroup per message.
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
    // (through the integer: a plain generic -> global -> generic pair of casts is folded away before it can tell anything)
    return (T*)(__attribute__((address_space(1))) T*)reinterpret_cast<uintptr_t>(p);
}

// BLOCKM (round 5): one LONG stream of the v1 format over all workgroups.  Without RLE / extended-match tokens every token
// writes exactly the bytes it consumes (compressor.c:651-657), so the window at input position p is input[p - W, p) (the
// dictionary where that reaches in front of the stream) HOWEVER the earlier bytes were parsed: find_best_match can be
// evaluated for any block of positions by any workgroup.  What a block's parse depends on is only WHERE it starts -- the
// previous block's last token reaches up to 14 bytes (minimum pattern + 13 - 1) into it.  Three launches:
//   pass 1 (this kernel, block_pass = 1): per block, match phase + jump tables, then for each of the 15 possible entry
//           offsets the chain's exit offset into the next block and the bits its tokens take (blk_table);
//   pass 2 (tamp_block_scan_kernel): one serial walk over the blocks' tables: entry offset and bit position of every block;
//   pass 3 (this kernel, block_pass = 3): match phase again, the walk from the block's true entry, the emitter's bits
//           ORed into the output at the block's bit position (first and last dword atomically: neighbours share them).
// The work counter hands out blocks instead of streams.  Reference shape: ONE stream of 100 MB (README.md:309-312,
// tools/c-profiler/main.c:52-54), which one workgroup takes 6 s for.
template <bool PACKED, bool LAZY, bool RUNS = false, uint32_t WSCAN = 0, uint32_t HB = kHashBits, bool LOOP = false, bool BLOCKM = false>
__global__ void __launch_bounds__(256, LAZY ? TAMP_LAZY_PER_CU : (RUNS ? TAMP_WG_PER_CU : TAMP_LEAN_PER_CU)) tamp_compress_kernel(CompressArgs a_k) {
    static_assert(!BLOCKM || (LOOP && PACKED && !LAZY), "block mode: a persistent build of the default parse");
    // HB: bucket bits of the bigram index (2,048 buckets; 512 for the short-message build, whose blocks hold a few hundred
    // positions and pay for every cursor zeroed and scanned); the cursor region keeps its size, the walk needs it
    static_assert(HB >= 9 && HB <= 11, "entry payload: 16 - HB bigram bits + 8 bits of the third byte + the rest of the fourth");
    constexpr uint32_t kRem = 16 - HB, kBuckets = 1u << HB;
    // Round 6, default parse with u32 entries: index entries laid out for ONE subtraction + ONE compare per candidate,
    //     rest of the bigram (kRem bits) << (kPB + kLB) | buffer position (kPB bits) << kLB | third byte | low bits of the fourth << 8
    // With x = entry ^ (the query's rest << (kPB + kLB) | its own bytes 2, 3) and y = x - (q << kLB):  y < (W - 1) << kLB  <=>
    // same bigram AND 0 <= position - q <= W - 2 (a position in front of the window borrows into the rest field; a foreign rest
    // leaves at least (2^kPB - q) << kLB, and 2^kPB - q >= W - 1 for every block a build runs).  y >> kLB is the distance from the
    // oldest window byte, the low kLB bits classify 2- / 3-byte candidates and the ones worth a compare -- where the 16 + 11 + 5
    // layout spent a masked subtraction, a three-way bit operation, two compares and a scalar AND.  The W = 2^10 build holds its
    // positions (below 2,352) in 12 bits and keeps 7 bits of the fourth byte; the others 16 and 3 (512 buckets: 1).
    constexpr bool kEntV2 = PACKED && !LAZY;
    constexpr uint32_t kPB = WSCAN == 1024 ? 12u : 16u, kLB = 32u - kRem - kPB;
    static_assert(!kEntV2 || kLB >= 9, "the third byte and at least one bit of the fourth");
    static_assert(!(RUNS && LAZY), "the run list serves the default parse only");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // LOOP builds: this workgroup's claim on the work counter (next stream, end of the claim, the stream in hand, start of
    // the claim after this one once fetched) lives in the first four words of LDS -- the slack in front of `ebuf`, which is only ever read under a mask -- at an address
    // that does not depend on the configuration
    typedef __attribute__((address_space(3))) volatile uint32_t LdsWord;
    constexpr uint32_t kNoClaim = 0xFFFFFFFFu;  // (the launcher keeps the counter below it)
    LdsWord* const claim = (LdsWord*)smem;
    if constexpr (LOOP) {
        if (threadIdx.x == 0) {
            const uint32_t nx = atomicAdd(a_k.work_counter, a_k.claim);
            claim[2] = nx, claim[0] = nx + 1, claim[1] = nx + a_k.claim, claim[3] = kNoClaim;
        }
        __syncthreads();
    }
    // LOOP builds (persistent grid, streams handed out by a counter): the arguments are read from the kernel-argument
    // segment again for every stream, through a pointer the compiler cannot see through.  Hoisted, the nine table pointers
    // and everything derived from the configuration stay in scalar registers across the whole stream body -- 25 more SGPR
    // spills and one spilled VGPR; reloaded, the body allocates like the one-stream build's.
    typedef const __attribute__((address_space(4))) uint32_t* KernArgs;
    struct ArgWords { uint32_t w[sizeof(CompressArgs) / 4]; };
    static_assert(sizeof(ArgWords) == sizeof(CompressArgs), "whole dwords");
    KernArgs ap = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t wave_k = Walk::uni(threadIdx.x >> 6);
    for (;;) {
        ArgWords aw;
        if constexpr (LOOP) {
            asm volatile("" : "+s"(ap));
#pragma unroll
            for (uint32_t i = 0; i < sizeof(CompressArgs) / 4; i++) aw.w[i] = ap[i];
        }
        CompressArgs a_l = __builtin_bit_cast(CompressArgs, aw);
        if constexpr (LOOP) {
            // (pointers out of plain dwords are generic ones, and every access through them a FLAT instruction: say what
            // the compiler knows by itself of a kernel argument)
            a_l.in = as_global(a_l.in), a_l.in_off = as_global(a_l.in_off), a_l.in_len = as_global(a_l.in_len);
            a_l.out = as_global(a_l.out), a_l.out_off = as_global(a_l.out_off), a_l.out_cap = as_global(a_l.out_cap);
            a_l.out_len = as_global(a_l.out_len), a_l.status = as_global(a_l.status), a_l.dict = as_global(a_l.dict);
            a_l.state = as_global(a_l.state), a_l.work_counter = as_global(a_l.work_counter), a_l.prof = as_global(a_l.prof);
            a_l.blk_table = as_global(a_l.blk_table), a_l.blk_info = as_global(a_l.blk_info);
            a_l.blk_len = as_global(a_l.blk_len), a_l.blk_idx = as_global(a_l.blk_idx);
        }
        const CompressArgs& a = LOOP ? a_l : a_k;
        const uint32_t a_wbits = a.wbits, a_blk = a.blk;
        const uint32_t W = 1u << a_wbits, mask = W - 1;
        constexpr bool lazy = LAZY;
        const CompressLds L(W, a_blk, PACKED, lazy, RUNS, HB == 9 ? kHashBits : HB);
        uint8_t* const ebuf = smem + L.ebuf;
        uint16_t* const cnt16 = reinterpret_cast<uint16_t*>(smem + L.cnt);
        uint32_t* const cntw = reinterpret_cast<uint32_t*>(smem + L.cnt);
        uint32_t* const ent = reinterpret_cast<uint32_t*>(smem + L.ent);
        uint16_t* const ent16 = reinterpret_cast<uint16_t*>(smem + L.ent);
        uint16_t* const toklist = reinterpret_cast<uint16_t*>(smem + L.ent);  // alias: index is dead during the walk
        uint16_t* const jump16 = reinterpret_cast<uint16_t*>(smem + L.ent + L.jump);   // alias, same reason
        uint8_t* const count8 = smem + L.ent + L.count;                                 // alias, same reason
        uint32_t* const jc32 = reinterpret_cast<uint32_t*>(smem + L.ent + L.jump);      // default parse: jump | count << 16
        uint8_t* const vstep = smem + L.ent + L.vstep;  // lazy builds: transition of every (position, state), same alias
        uint32_t* const stok = reinterpret_cast<uint32_t*>(smem + L.cnt);     // alias: cursors are dead during the walk
        // RUNS builds: bytes consumed by the tokens the match phase settles completely (short RLE runs, extended matches
        // without a rival): second half of the cursor space, behind the 256 explicit pieces of `stok`; written after the
        // match loops (when `sorted` is dead), read by the jump tables, the walk's token listing and the emitter
        uint8_t* const xcnt = smem + L.cnt + kSlowCap * 8;
        uint8_t* const blen = smem + L.blen;
        uint16_t* const bidx = reinterpret_cast<uint16_t*>(smem + L.bidx);
        uint8_t* const blen2 = smem + L.blen2;                                    // only carved when lazy
        uint16_t* const bidx2 = reinterpret_cast<uint16_t*>(smem + L.bidx2);
        uint32_t* const obuf = reinterpret_cast<uint32_t*>(smem + L.obuf);
        uint16_t* const qstart = reinterpret_cast<uint16_t*>(smem + L.obuf + 16);  // alias: bit buffer is idle during match
        uint16_t* const sorted = reinterpret_cast<uint16_t*>(smem + L.cnt);        // alias: cursors are dead after the scatter
        // (an explicit LDS pointer: the address-space inference leaves volatile accesses alone, and through a generic pointer
        // every control word was a FLAT load with system scope followed by a full wait)
        typedef __attribute__((address_space(3))) volatile uint32_t LdsCtl;
        LdsCtl* const ctl = (LdsCtl*)(smem + L.ctl);
        uint32_t* const bins = reinterpret_cast<uint32_t*>(smem + L.ctl + 80);
        // prefix codes by symbol for per-lane look-ups (the packed 64-bit constants would sit in four VGPRs all kernel long)
        uint8_t* const codetab = smem + L.ctl + 80 + 256;
        uint32_t* const runs = reinterpret_cast<uint32_t*>(smem + L.runs);    // RUNS builds only
        uint32_t* const runsx = reinterpret_cast<uint32_t*>(smem + L.runsx);  // RUNS builds only
        uint32_t* const rxset = reinterpret_cast<uint32_t*>(smem + L.rxset);  // RUNS builds only
        uint32_t* const rbits = reinterpret_cast<uint32_t*>(smem + L.rbits);  // RUNS builds only

        uint32_t tid_l = threadIdx.x;
        if constexpr (LOOP) {
            // the thread index put together again for every stream from the wavefront's number (one scalar register) and
            // the lane's, behind volatile asm: seen as the loop invariant it is, the index -- and everything computed from it
            // alone: table addresses, the prefix-code bytes -- is held in VGPRs across the whole stream body, and spilled
            uint32_t lane_l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l));
            asm volatile("" : "+s"(wave_k));
            tid_l = (wave_k << 6) | lane_l;
        }
        const uint32_t tid_k = tid_l, nt = blockDim.x;
        const uint32_t nt_log2 = nt == 256 ? 8u : 6u;  // (256 or 64 threads: divisions by the block size are shifts)
        uint32_t tid = tid_k;
        int lane = tid & (kWave - 1);
        uint32_t wave = tid >> 6;
        const uint32_t minp = (uint32_t)min_pattern_size((int)a_wbits, a.lbits);
        const bool ext = a.extended != 0;
        const uint32_t maxp = ext ? minp + 11 + kExtExtraMax : minp + 13;  // compressor.c:12-19
        const uint32_t wbits = a_wbits, lbits = a.lbits;
#ifdef TAMP_POISON_LDS
        // test builds: every stream starts on LDS full of a pattern that changes from launch to launch and stream to stream, so
        // that a result which depends on what an earlier workgroup left behind shows in the differential runs at once
        {
            __syncthreads();
            const uint32_t seed = (uint32_t)__builtin_readcyclecounter() | 1u;
            for (uint32_t k = 4 + tid_k; k < L.total / 4; k += nt)  // (the first four words: this workgroup's claim)
                reinterpret_cast<uint32_t*>(smem)[k] = (TAMP_POISON_LDS) ? (k + 1) * 2654435761u * seed : 0xFFFFFFFFu;
            __syncthreads();
        }
#endif
        if (tid_k < 15) codetab[tid_k] = (uint8_t)tok_code(tid_k);  // visible after the first barrier (every stream writes the same)
        if (RUNS && tid_k == 0) ctl[cQuad] = 0;  // (first read after the first barrier; every stream leaves it at zero)

        uint32_t s;
        if constexpr (LOOP) {
            // this workgroup's next stream (chosen behind the previous one, below)
            s = Walk::uni(claim[2]);
            if (s >= a.n_streams) break;
        } else {
            s = blockIdx.x + a.first_stream;
        }
        // per-stream table entries are wave-uniform but arrive through vector loads (the compiler cannot prove the
        // tables invariant): pin them to scalar registers, or the two base pointers sit in VGPR pairs -- and spill
        const uint32_t si = BLOCKM ? 0u : s;  // (block mode: `s` is the block, the tables have one row)
        const uint8_t* const in = a.in + uni_u64(a.in_off[si]);
        const uint32_t n = Walk::uni(a.in_len[si]);
        uint8_t* const gout = a.out + uni_u64(a.out_off[si]);
        const uint32_t cap = Walk::uni(a.out_cap[si]);
        const uint32_t bpos = BLOCKM ? s * a_blk : 0u;  // block mode: input position of this block
        unsigned long long binfo = 0;                   // ... and (pass 3) bit position << 4 | entry offset
        if constexpr (BLOCKM) {
            if (a.block_pass == 3) binfo = uni_u64(a.blk_info[s]);
        }

        uint8_t* const st_io = a.state ? a.state + (size_t)s * (W + kSegStateExtra) : nullptr;
        const bool partial = (a.seg_flags & kSegPartial) != 0;
        uint32_t wp0 = 0;
        if constexpr (BLOCKM) {
            // window <- the W bytes in front of the block, oldest first: input, or -- in front of the stream -- the dictionary
            // byte that still sits at that ring index (window_pos of a fresh stream = bytes written mod W)
            wp0 = bpos & mask;
            for (uint32_t k = tid; k < W; k += nt) ebuf[k] = (bpos + k >= W) ? in[bpos + k - W] : a.dict[(bpos + k) & mask];
        } else
        if (st_io && (a.seg_flags & kSegResume)) {
            // window <- saved state (ring order) rotated so that the oldest byte comes first
            wp0 = (uint32_t)st_io[W] | ((uint32_t)st_io[W + 1] << 8);
            for (uint32_t k = tid; k < W; k += nt) ebuf[k] = st_io[(wp0 + k) & mask];
        } else if ((reinterpret_cast<uintptr_t>(a.dict) & 3) == 0) {
            // window <- dictionary (custom, or the seeded default prepared by the host shim)
            for (uint32_t k = tid * 4; k < W; k += nt * 4)
                *reinterpret_cast<uint32_t*>(ebuf + k) = *reinterpret_cast<const uint32_t*>(a.dict + k);
        } else {
            for (uint32_t k = tid; k < W; k += nt) ebuf[k] = a.dict[k];
        }
        // carried over from a call that ended without a flush (kSegStateExtra): pending run / extended match, pending bits
        uint32_t c_rle = 0, c_ext = 0, c_extpos = 0, c_nbits = 0, c_bits = 0;
        if (st_io && (a.seg_flags & kSegResume)) {
            c_rle = Walk::uni(st_io[W + 3]), c_ext = Walk::uni(st_io[W + 4]), c_nbits = Walk::uni(st_io[W + 5]) & 31u;
            c_extpos = Walk::uni((uint32_t)st_io[W + 6] | ((uint32_t)st_io[W + 7] << 8));
            c_bits = Walk::uni((uint32_t)st_io[W + 16] | ((uint32_t)st_io[W + 17] << 8) | ((uint32_t)st_io[W + 18] << 16) |
                               ((uint32_t)st_io[W + 19] << 24));
        }
        // bit buffer: leading bytes (header, compressor.c:236-241; FLUSH + pad when appending, :227-235) or the carried
        // bits, rest zero
        const uint32_t word0 = (BLOCKM && s != 0) ? 0u : (a.nlead ? (uint32_t)a.lead << 16 : (c_nbits ? c_bits & (0xFFFFFFFFu << (32 - c_nbits)) : 0u));
        for (uint32_t k = tid; k < L.obuf_words; k += nt) obuf[k] = k == 0 ? __builtin_bswap32(word0) : 0;

        Walk wk;
        wk.ebuf = ebuf, wk.blen = blen, wk.bidx = bidx, wk.toklist = toklist, wk.stok = stok;
        wk.W = W, wk.mask = mask, wk.wbits = wbits, wk.lbits = lbits, wk.minp = minp, wk.ext = ext;
        wk.wp_e = wp0, wk.wr = 0, wk.rd = 0, wk.nvalid = 0;
        // (a carried run / extended match: its bytes lead the input and count as consumed, like a pending token after a re-base)
        wk.rle_count = c_rle, wk.ext_count = c_ext, wk.ext_pos = c_extpos, wk.rd = c_rle + c_ext, wk.partial = partial;
        wk.ext_resolved = false, wk.ntok = 0, wk.ns = 0, wk.lane = lane;
        wk.ctlw = (Walk::CtlWord*)ctl, wk.coop = nt == 256;
#ifdef TAMP_PROF
        wk.dbg = a.dbg;
#endif
        wk.lazy = lazy, wk.lazy_valid = false, wk.lazy_idx = 0, wk.lazy_len = 0, wk.blen2 = blen2, wk.bidx2 = bidx2;
        if (tid_k == 0) ctl[cCutThr] = a.cut_run;  // (read after the load phase's barrier)
        uint32_t w_p0 = bpos;  // wave 0: input position of ebuf[W]

        // workgroup-uniform output state
        uint32_t carry = BLOCKM ? ((uint32_t)(binfo >> 4) & 31u) : (a.nlead ? 8u * a.nlead : c_nbits);  // bits already sitting in obuf
        uint32_t gpos = 0;                         // bytes already flushed to HBM
        uint32_t e_p0 = bpos, e_pending = c_rle + c_ext, e_wp = wp0;  // epoch parameters
        bool need_match = true;
        // Positions matched per epoch.  A token that breaks the speculation throws the rest of the block away, so
        // after such a break the next block is small (data with long runs / window-end truncations tends to break
        // again soon); a block that ends cleanly doubles it back up to the LDS capacity.
        uint32_t cur_blk = a_blk;
#ifdef TAMP_PROF
        unsigned long long pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long pc = __builtin_readcyclecounter();
#endif
        for (;;) {
            // (opaque re-definition: thread-indexed LDS addresses are recomputed per epoch instead of being hoisted out of
            // the stream loop, where they would sit in -- and spill from -- registers across every phase)
            asm volatile("" : "+v"(tid));
            lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;  // (re-derived: see above)
            const uint32_t left = n - e_p0;
            // positions this epoch matches: the block, or less when a long run of one byte cuts it short (below; the walk's
            // later passes over the same tables read the figure back)
            uint32_t nvalid = left < cur_blk ? left : cur_blk;
            // Ring-end alignment (extended format).  An extended match that is being written when the window's write
            // cursor reaches the end of the ring is clipped there (compressor.c:404-410): a lag, wherever the match started.
            // Where the ring ends is known when the epoch starts -- block position W - window_pos -- so the block ends
            // THERE when that is not too close: a clipped match then costs nothing (the epoch was over anyway) instead of the
            // rest of the block's matches and an epoch of its own.  With 1,024-position blocks at W = 2^10 a stream without
            // lags keeps its epochs on the ring's revolutions by itself; after an RLE lag the next block is the rest of the
            // revolution.  Markup (1.2 clipped matches per 4 KiB), prose (0.7) and Python sources (1.0) gain; a guess about
            // cost only, like the cut below.
            if (ext && !LAZY) {
                const uint32_t k0 = W - e_wp;
                if (k0 >= TAMP_ALIGN_MIN && k0 < nvalid) nvalid = k0;
            }
            const uint32_t nplan = nvalid;  // (what the block would be without a cut)
            if (!need_match) nvalid = Walk::uni(ctl[cCut]);
            uint32_t nv = LAZY ? 2 * nvalid : nvalid;  // states the walk's tables cover (lazy: position x {fresh, cached})
            const uint8_t* const steps = LAZY ? vstep : blen;
            if (need_match) {
                __builtin_amdgcn_s_setprio(kPrioShort);
                // ---------------- load: ebuf[W + k] = in[e_p0 + k] ----------------
                const uint32_t room = cur_blk + kRing + kPendMax;
                const uint32_t nload = left < room ? left : room;
                TAMP_REPEAT(0x10000u) {
                    const uint8_t* src = in + e_p0;
                    const uint32_t nfill = align_up(nload + 20, 4);  // zero tail: stray look-ahead reads are defined
#ifndef TAMP_LOAD_BYTES
                    {
                        // whole dwords wherever the block starts in the stream (global loads need no alignment on gfx950): behind a
                        // lag the block starts at an odd position three times out of four, and a byte at a time this was six
                        // dependent round trips to HBM per epoch -- 156 us of the 1.1 ms the slowest chunk of the stand-in takes alone
                        const uint32_t nw = nload >> 2;
                        for (uint32_t k = tid; k < nw; k += nt) {
                            uint32_t v;
                            __builtin_memcpy(&v, src + 4 * k, 4);
                            reinterpret_cast<uint32_t*>(ebuf + W)[k] = v;
                        }
                        for (uint32_t k = (nw << 2) + tid; k < nfill; k += nt) ebuf[W + k] = k < nload ? src[k] : 0;
                    }
#else
                    if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
                        const uint32_t nw = nload >> 2;
                        for (uint32_t k = tid; k < nw; k += nt)
                            reinterpret_cast<uint32_t*>(ebuf + W)[k] = reinterpret_cast<const uint32_t*>(src)[k];
                        for (uint32_t k = (nw << 2) + tid; k < nfill; k += nt) ebuf[W + k] = k < nload ? src[k] : 0;
                    } else {
                        for (uint32_t k = tid; k < nfill; k += nt) ebuf[W + k] = k < nload ? src[k] : 0;
                    }
#endif
                }
                for (uint32_t k = tid; k < kBuckets / 2; k += nt) cntw[k] = 0;
                if (tid == 0) ctl[cCut] = 0xFFFFFFFFu;
                __syncthreads();
                TAMP_PROF_MARK(0);
#ifdef TAMP_PROF
                pt[12] += 1, pt[13] += nvalid;  // epochs, positions matched
#endif

                bool have_tables = false;
                if constexpr (BLOCKM) {
                    have_tables = a.block_pass == 3 && a.blk_len != nullptr;
                    if (have_tables) {  // pass 3: what pass 1 matched for this block comes back from HBM
                        const uint8_t* const gl = a.blk_len + bpos;
                        const uint16_t* const gi = a.blk_idx + bpos;
                        for (uint32_t k = tid * 4; k < nvalid; k += nt * 4) {
                            *reinterpret_cast<uint32_t*>(blen + k) = *reinterpret_cast<const uint32_t*>(gl + k);
                            *reinterpret_cast<uint2*>(bidx + k) = *reinterpret_cast<const uint2*>(gi + k);
                        }
                        __syncthreads();
                        for (uint32_t k = nvalid + tid; k < nvalid + 128 && k < a_blk + 128; k += nt) blen[k] = 0x80;  // sentinels
                        __syncthreads();
                    }
                }
                if (!have_tables) {
                // ---------------- index: counting sort of buffer positions by bigram ----------------
#ifdef TAMP_PROF
                uint32_t index_reps = (a.dbg & 0x20000u) ? 2u : 1u;
            index_again:
#endif
                asm volatile("" : "+v"(tid));
                lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;  // (re-derived: see above)
                const uint32_t NE0 = nvalid ? W + nvalid : 0;  // positions 0..NE0-1 (every query's own bigram included)
                const uint32_t cut_run = Walk::uni(ctl[cCutThr]);  // (per stream: it adapts, see the walk's re-base)
                for (uint32_t c4 = tid * 4; c4 < NE0; c4 += nt * 4) {
                    const uint32_t d0 = *reinterpret_cast<const uint32_t*>(ebuf + c4);
                    const uint32_t d1 = *reinterpret_cast<const uint32_t*>(ebuf + c4 + 4);
                    if (d0 == (d0 & 0xFFu) * 0x01010101u) {
                        // RUNS builds: any run of 7+ bytes holds an aligned dword of four equal bytes; epochs without one
                        // skip the run search
                        if constexpr (RUNS) ctl[cQuad] = 1;
                        // Epoch cut.  In the extended format a long run of one byte ahead of the walk nearly always
                        // becomes an RLE token that writes 8 bytes for all it consumes (compressor.c:342-359): what was
                        // matched behind it under "everything consumed is written" is thrown away.  The first run of
                        // `cut_run` aligned dwords among this epoch's new positions therefore ends the block just inside
                        // it; the next epoch starts where the walk comes out of the run.  (A guess about cost only: a
                        // match that covers the run after all leaves the walk at the end of a shorter block.)
                        if (cut_run && d1 == d0 && c4 >= W + e_pending + 4 && c4 + 4 * cut_run <= NE0) {
                            bool all = true;
                            for (uint32_t k = 2; k < cut_run; k++) all = all && *reinterpret_cast<const uint32_t*>(ebuf + c4 + 4 * k) == d0;
                            if (all) atomicMin((uint32_t*)&ctl[cCut], c4);
                        }
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        if (c4 + j < NE0) {
                            const uint32_t h = mix16<HB>(__builtin_amdgcn_alignbyte(d1, d0, j) & 0xFFFFu) >> kRem;
                            atomicAdd(&cntw[h >> 1], 1u << ((h & 1) * 16));
                            // (round 6: a block position's bucket stays in its `qstart` slot until the scatter turns it into the
                            // cursor snapshot -- the bigram was loaded and hashed a second time there)
                            if (c4 >= W) qstart[c4 + j - W] = (uint16_t)h;
                        }
                    }
                }
                __syncthreads();
                {
                    const uint32_t cutc = Walk::uni(ctl[cCut]);
                    if (cutc < NE0) nvalid = cutc - W + 4, nv = LAZY ? 2 * nvalid : nvalid;
                }
                const uint32_t NE = nvalid ? W + nvalid : 0;  // positions the index lists (the counts above are upper bounds)
                for (uint32_t k = nvalid + tid; k < nvalid + 128 && k < a_blk + 128; k += nt) blen[k] = 0x80;  // sentinels
                uint32_t nruns = 0;  // listed runs of this epoch (RUNS builds)
                if constexpr (RUNS) {
                    if (Walk::uni(ctl[cQuad])) {
                        // Long runs of one byte.  Every position inside such a run carries the same bigram: the bucket
                        // of (x, x) grows with the run lengths and every query that starts with x x walks all of it.
                        // The interior of a listed run [a, b) -- positions a+1 .. b-4: preceded by x and followed by
                        // three more -- leaves the index (the counts above stay upper bounds; the scatter below skips
                        // them); the match phase derives, per run, the few interior positions that can hold the best
                        // match (see there).  Heads scan their own run; ends are capped a little past the indexed
                        // range (any run remainder > 16 acts the same).
                        for (uint32_t k = tid; k < (W + a_blk + 96) / 32; k += nt) rbits[k] = 0;
                        if (tid == 0) ctl[cNruns] = 0;
                        if (tid < 8) rxset[tid] = 0;
                        __syncthreads();
                        if (tid == 0) ctl[cQuad] = 0;  // (everybody has read it: for the next epoch)
                        const uint32_t lim = NE + 16;
                        for (uint32_t c = tid; c < NE; c += nt) {
                            const uint32_t b4 = lds_u32_unaligned(ebuf, c), x = b4 & 0xFFu;
                            if (b4 == x * 0x01010101u && (c == 0 || ebuf[c - 1] != x)) {
                                uint32_t e = c + 4;
                                while ((e & 3u) && e < lim && ebuf[e] == x) e++;
                                if (!(e & 3u)) {
                                    while (e + 8 <= lim) {  // aligned dword pairs
                                        const uint32_t* wv = reinterpret_cast<const uint32_t*>(ebuf + e);
                                        const uint32_t v0 = wv[0], v1 = wv[1];
                                        if (v0 != b4 || v1 != b4) break;
                                        e += 8;
                                    }
                                    while (e < lim && ebuf[e] == x) e++;
                                }
                                if (e - c >= kLongRun) {
                                    const uint32_t slot = atomicAdd((uint32_t*)&ctl[cNruns], 1u);
                                    if (slot < kRunCap) {
                                        runs[slot] = c | (e << 16);
                                        runsx[slot] = x | ((lds_u32_unaligned(ebuf, e) & 0xFFFFu) << 8);
                                        atomicOr(&rxset[x >> 5], 1u << (x & 31u));
                                        for (uint32_t k = c + 1; k <= e - 4;) {  // bits [c+1, e-4]
                                            const uint32_t hiw = min(e - 4, k | 31u);
                                            const uint32_t m = (0xFFFFFFFFu << (k & 31u)) & (0xFFFFFFFFu >> (31u - (hiw & 31u)));
                                            atomicOr(&rbits[k >> 5], m);
                                            k = hiw + 1;
                                        }
                                    }
                                }
                            }
                        }
                        __syncthreads();
                        nruns = min(Walk::uni(ctl[cNruns]), kRunCap);
                    }
                }
                {  // exclusive scan of the 2048 u16 counters in place
                    const uint32_t per = kBuckets >> nt_log2;  // 8 (256 threads) or 32 (64 threads)
                    uint32_t sum = 0;
                    for (uint32_t k = 0; k < per; k++) sum += cnt16[tid * per + k];
                    const uint32_t incl = wave_scan_add(sum);
                    if (lane == kWave - 1) ctl[cWave + wave] = incl;
                    __syncthreads();
                    uint32_t run = incl - sum;
                    for (uint32_t w2 = 0; w2 < wave; w2++) run += ctl[cWave + w2];
                    for (uint32_t k = 0; k < per; k++) {
                        uint32_t v = cnt16[tid * per + k];
                        cnt16[tid * per + k] = (uint16_t)run;
                        run += v;
                    }
                }
                __syncthreads();
                // Tile-ordered scatter.  After tile t every bucket lists the positions of tiles 0..t in tile order.
                // Two cursor snapshots bracket what a query must scan: the cursor of its bucket when the tile holding
                // its oldest window byte starts (qstart) and after the tile holding its own position (top, in bidx).
                if (tid < nvalid) qstart[tid] = cnt16[qstart[tid]];  // queries of tile 0: bucket start (the slot held the bucket's number)
                __syncthreads();
                for (uint32_t t0 = 0; t0 < NE; t0 += nt) {
                    const uint32_t c = t0 + tid;
                    uint32_t h = 0;
                    if (c < NE) {
                        const uint32_t b4 = lds_u32_unaligned(ebuf, c);
                        const uint32_t mx = mix16<HB>(b4 & 0xFFFFu);
                        h = mx >> kRem;
                        const uint32_t sh = (h & 1) * 16;
                        bool keep = true;
                        if constexpr (RUNS) { if (nruns) keep = !((rbits[c >> 5] >> (c & 31u)) & 1u); }
                        if (keep) {
                            const uint32_t old = atomicAdd(&cntw[h >> 1], 1u << sh);
                            if (kEntV2)
                                ent[(old >> sh) & 0xFFFFu] = ((mx & ((1u << kRem) - 1)) << (kPB + kLB)) | (c << kLB) | ((b4 >> 16) & ((1u << kLB) - 1));
                            else if (PACKED)
                                ent[(old >> sh) & 0xFFFFu] = c | entry_payload<kRem>(b4, mx);
                            else
                                ent16[(old >> sh) & 0xFFFFu] = (uint16_t)c;
                        }
                    }
                    __syncthreads();
                    if (c < NE && c >= W) bidx[c - W] = cnt16[h];
                    const uint32_t q2 = t0 + nt + tid;  // queries whose oldest window byte lies in the next tile
                    if (q2 < nvalid) qstart[q2] = cnt16[qstart[q2]];
                    __syncthreads();
                }
                // Order the queries by scan length (counting sort, longest first) so that the 64 lanes of a wave
                // loop about equally often: the greedy per-lane scan is otherwise paced by its longest bucket.
                if (tid < 64) bins[tid] = 0;
                if (tid == 0) ctl[cCut] = nvalid;  // (everybody has read the cut position: several barriers back)
                __syncthreads();
                for (uint32_t q = e_pending + tid; q < nvalid; q += nt) {
                    const uint32_t Lq = min((uint32_t)bidx[q] - (uint32_t)qstart[q], 63u);
                    atomicAdd(&bins[63 - Lq], 1u);
                }
                __syncthreads();
                if (wave == 0) {
                    const uint32_t v = bins[lane];
                    const uint32_t incl = wave_scan_add(v);
                    bins[lane] = incl - v;
                }
                __syncthreads();
                for (uint32_t q = e_pending + tid; q < nvalid; q += nt) {
                    const uint32_t Lq = min((uint32_t)bidx[q] - (uint32_t)qstart[q], 63u);
                    sorted[atomicAdd(&bins[63 - Lq], 1u)] = (uint16_t)q;
                }
                __syncthreads();
#ifdef TAMP_PROF
                if (--index_reps) {  // (again, from zeroed cursors: a cut found the first time has already shortened the block)
                    for (uint32_t k = tid; k < kBuckets / 2; k += nt) cntw[k] = 0;
                    if (tid == 0) ctl[cCut] = 0xFFFFFFFFu;
                    __syncthreads();
                    goto index_again;
                }
#endif
                TAMP_PROF_MARK(1);

                // ---------------- match: find_best_match for every position of the block ----------------
                __builtin_amdgcn_s_setprio(kPrioScan);
                asm volatile("" : "+v"(tid));
                lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;  // (re-derived: see above)
                const uint32_t nq = nvalid - e_pending;
#ifdef TAMP_PROF
                unsigned long long f0 = 0, f1 = 0, f2 = 0, f3 = 0, niter = 0;
#define TAMP_FINE(v) do { unsigned long long _n = __builtin_readcyclecounter(); v += _n - fc; fc = _n; } while (0)
                unsigned long long fc = __builtin_readcyclecounter();
#else
#define TAMP_FINE(v) do { } while (0)
#endif
#ifdef TAMP_PROF
                if (!(a.dbg & 1))
#endif
                for (uint32_t j = tid; j < nq; j += nt) {
                    const uint32_t q = sorted[j];
                    const uint32_t leftq = n - (e_p0 + q);
                    const uint32_t R = leftq < kRing ? leftq : kRing;
                    uint32_t key = 0;
                    uint32_t wrapbest = 0, n16 = 0;  // RUNS builds: best wrap-zone hit, number of 16-byte hits of the scan
                    bool sole_ext = false;
                    // lazy matching: the same pattern also probes the window as it was one position earlier (window
                    // start q-1, one byte less look-ahead): compressor.c:585-596 restated per position
                    uint32_t keyB = 0, wrapmaskB = 0, cap_lenB = 0;
                    if (lazy && q >= 1) {
                        const uint32_t leftp = n - (e_p0 + q - 1);
                        const uint32_t Rb = (leftp < kRing ? leftp : kRing) - 1;
                        cap_lenB = Rb >= minp ? (Rb < maxp ? Rb : maxp) : 0;
                    }
                    uint32_t P[4];
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) P[jj] = lds_u32_unaligned(ebuf, W + q + 4 * jj);
                    // Inside a run of one byte the extended state machine never asks for a match: with the previous
                    // byte repeated 7+ times ahead, or up to the end of the ring, the RLE path owns the position
                    // (compressor.c:470-503 only consults find_best_match for runs of 2..6 that END inside the ring).
                    // Skipping the scan there removes the worst buckets.
                    const uint32_t rep = (P[0] & 0xFFu) * 0x01010101u;
                    bool in_run = false;
#ifndef TAMP_SETUP_R5
                    // (round 6: the byte in front of the pattern is read ONCE -- it is also the newest window byte of the wrap-zone
                    // test and the byte the slow flag looks at -- and the run arithmetic runs only where the pattern starts with
                    // TWO of it: with one, r <= 1 < the shortest pattern and nothing below applies, and some lane of nearly every
                    // wavefront had one)
                    const uint32_t prevb = ebuf[W + q - 1];
                    if (ext && !lazy && (P[0] & 0xFFFFu) == prevb * 0x0101u) {
#else
                    const uint32_t prevb = ebuf[W + q - 1];
                    if (ext && !lazy && prevb == (P[0] & 0xFFu)) {
#endif
                        // r = leading ring bytes equal to the previous byte, looked at up to 7
                        const uint32_t x0 = P[0] ^ rep, x1 = (P[1] ^ rep) & 0x00FFFFFFu;
                        const uint32_t r = x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : (x1 ? 4 + ((uint32_t)__builtin_ctz(x1) >> 3) : 7u);
                        in_run = r >= 7 || r >= R;  // the run reaches past the arbitration limit or to the end of the ring
                        // a short run, with a crowded bucket: the walk asks when -- if -- it gets here (kDeferred)
                        // (the marker travels in `key`: length field kDeferred, slow by the bytes' own rule below)
                        if (!in_run && r >= 2 && (uint32_t)bidx[q] - (uint32_t)qstart[q] >= kDeferMin) in_run = true, key = kDeferred << 16;
                    }
#ifdef TAMP_PROF
                    if (R >= minp && !in_run && !(a.dbg & 2)) {
#else
                    if (R >= minp && !in_run) {
#endif
                        const uint32_t cap_len = R < maxp ? R : maxp;
                        const uint32_t pk = entry_payload<kRem>(P[0], mix16<HB>(P[0] & 0xFFFFu));
                        const uint32_t chi = q + W - 2;  // newest candidate served by the index
                        const uint32_t s_hi = bidx[q];
                        uint32_t sl = qstart[q], wrapmask = 0;
                        uint32_t e_next = PACKED ? ent[sl] : (uint32_t)ent16[sl];  // software prefetch of the next entry
                        TAMP_FINE(f0);
                        if constexpr (kEntV2) {
                            const uint32_t Ws = WSCAN ? WSCAN : W;  // (see the template parameter)
                            const uint32_t pkx = ((mix16<HB>(P[0] & 0xFFFFu) & ((1u << kRem) - 1)) << (kPB + kLB)) | ((P[0] >> 16) & ((1u << kLB) - 1));
                            const uint32_t qs = q << kLB;
                            const uint32_t nb = ~(q + e_wp);  // W - window index of the candidate at distance d = ((nb - d) & (W - 1)) + 1
                            const uint32_t* pe = ent + sl;
                            const uint32_t* const pe_end = ent + s_hi;
#ifdef TAMP_PROF
                            // instruction-count experiments: run the (idempotent) loop twice, count the difference
                            uint32_t n16_first = 0;
                            for (uint32_t rep = 0; rep < ((a.dbg & 0x100u) ? 2u : 1u); rep++) {
                            if (rep) { n16_first = n16; pe = ent + sl; e_next = *pe; }
#endif
                            while (pe < pe_end) {
#ifdef TAMP_PROF
                                niter++;
#endif
                                const uint32_t e = e_next;
                                pe++;
                                e_next = *pe;  // one past the range at the end: harmless
                                const uint32_t y = (e ^ pkx) - qs;
                                if (y < ((Ws - 1) << kLB)) {  // in the window and the same bigram
                                    const uint32_t d = y >> kLB;  // Ws - d bytes before the candidate reaches the newest byte
                                    const uint32_t low = y & ((1u << kLB) - 1);
                                    uint32_t len = (low & 0xFFu) ? 2u : 3u;
#ifdef TAMP_PROF
                                    if (low == 0 && !(a.dbg & 0x8000u)) {
                                        len = prefix_len16(ebuf, q + d, P);
                                        if (a.dbg & 0x1000000u) {  // (the compare once more -- same result -- for its exact count)
                                            asm volatile("" ::: "memory");
                                            len = max(len, prefix_len16(ebuf, q + d, P));
                                        }
                                    } else if (low == 0) len = 4u;
#else
                                    if (low == 0) len = prefix_len16(ebuf, q + d, P);  // the next two bytes agree too (the entry's bits of the second)
#endif
                                    // (the wrap zone and the key: as in the generic loop below.  One test: the compare reaches the
                                    // newest byte iff d + len >= Ws; a 16-byte hit exactly 16 bytes in front of it -- exact as it
                                    // stands -- goes the same way and comes out of the wrapped compare with the same 16)
                                    if (d + len >= Ws) {
                                        wrapmask |= 1u << (Ws - d);
                                    } else {
                                        if constexpr (RUNS) n16 += len >> 4;
                                        const uint32_t lim_i = ((nb - d) & (Ws - 1)) + 1;
                                        key = max(key, (min(len, min(cap_len, lim_i)) << 16) | lim_i);
                                    }
                                }
                            }
#ifdef TAMP_PROF
                            if (rep) n16 = n16_first;
                            }
#endif
                        } else if constexpr (!LAZY) {
                            const uint32_t Ws = WSCAN ? WSCAN : W;  // (see the template parameter)
#ifdef TAMP_PROF
                            // instruction-count experiments: run the (idempotent) loop twice, count the difference
                            const uint32_t sl0 = sl;
                            uint32_t n16_first = 0;
                            for (uint32_t rep = 0; rep < ((a.dbg & 0x100u) ? 2u : 1u); rep++) {
                            if (rep) { n16_first = n16; sl = sl0; e_next = PACKED ? ent[sl] : (uint32_t)ent16[sl]; }
#endif
                            while (sl < s_hi) {
#ifdef TAMP_PROF
                                niter++;
#endif
                                const uint32_t e = e_next;
                                sl++;
                                e_next = PACKED ? ent[sl] : (uint32_t)ent16[sl];  // one past the range at the end: harmless
                                // position-only entries: everything is "deep", the byte compare decides
                                const uint32_t x = PACKED ? (e ^ pk) >> 16 : 0u;
                                const uint32_t c = e & 0xFFFFu;
                                const uint32_t d = c - q;              // distance from the oldest window byte
                                const uint32_t i = (e + e_wp) & (Ws - 1);  // window index (payload bits masked off)
                                // in the window and the same bigram.  (Index W-1 cannot start a match: its limit
                                // W - i = 1 rejects it below.)
                                if (d <= Ws - 2 && (x & ((1u << kRem) - 1)) == 0) {
                                    const uint32_t t = Ws - d;  // bytes before the candidate reaches the newest byte
                                    uint32_t len = (x & (0xFFu << kRem)) ? 2u : 3u;
#ifdef TAMP_PROF
                                    // (0x8000: instruction-count experiment -- the loop without its 16-byte compares, i.e.
                                    // what any scheme that takes the deep compares elsewhere leaves behind; results are wrong)
                                    if ((x >> kRem) == 0 && !(a.dbg & 0x8000u)) {
                                        len = prefix_len16(ebuf, c, P);
                                        if (a.dbg & 0x1000000u) {  // (round 5: the compare once more -- same result -- for its exact count)
                                            asm volatile("" ::: "memory");
                                            len = max(len, prefix_len16(ebuf, c, P));
                                        }
                                    } else if ((x >> kRem) == 0) len = 4u;
#else
                                    if ((x >> kRem) == 0) len = prefix_len16(ebuf, c, P);  // next two bytes agree too
#endif
                                    // The candidate's first t bytes lie in front of the newest window byte, where the
                                    // buffer IS the ring: a common prefix shorter than t is exact whatever follows.  Only
                                    // a candidate that agrees all the way to the newest byte (periodic input: rare) goes
                                    // on with the OLDEST window bytes and is resolved after the loop.  (Its payload and
                                    // the 16-byte compare look at buffer bytes behind the newest one there, but any
                                    // length they report is then >= t.)
                                    if (t < 16 && len >= t) {
                                        wrapmask |= 1u << t;
                                    } else {
                                        if constexpr (RUNS) n16 += len >> 4;
                                        // key = length << 16 | (W - index): longest, then lowest index.  W - i is also
                                        // the limit "may not run past index W-1"; a clipped length of 1 (index W-1)
                                        // yields a key below every real match and is read as "no match" later.
                                        const uint32_t lim_i = Ws - i;
                                        key = max(key, (min(len, min(cap_len, lim_i)) << 16) | lim_i);
                                    }
                                }
                            }
#ifdef TAMP_PROF
                            if (rep) n16 = n16_first;
                            }
#endif
                        } else
                        while (sl < s_hi) {
#ifdef TAMP_PROF
                            niter++;
#endif
                            const uint32_t e = e_next;
                            sl++;
                            e_next = PACKED ? ent[sl] : (uint32_t)ent16[sl];  // one past the range at the end: harmless
                            const uint32_t c = e & 0xFFFFu;
                            // position-only entries: everything is "deep", the byte compare decides
                            const uint32_t x = PACKED ? (e ^ pk) >> 16 : 0u;
                            const uint32_t i = (e_wp + c) & mask;  // window index of the candidate
                            // in the window, same bigram, and not index W-1 (which cannot start a match)
                            const bool same = (x & ((1u << kRem) - 1)) == 0 && i != mask;
                            const bool ok = c >= q && c <= chi && same;
                            const bool okB = cap_lenB && c >= q && c + 1 <= chi && same;  // window [q-1, q+W-1); c = q-1 apart
                            const uint32_t lim = min(cap_len, W - i);  // may not run past index W-1
                            const uint32_t t = q + W - c;              // bytes before the candidate reaches the newest byte
                            uint32_t len = (x & (0xFFu << kRem)) ? 2u : 3u;
                            if ((ok || okB) && (x >> kRem) == 0 && t > 16) len = prefix_len16(ebuf, c, P);  // next two bytes agree too
                            if (ok) {
                                if (t < 16) {
                                    wrapmask |= 1u << t;  // runs past the newest window byte: resolved after the loop
                                } else {
                                    const uint32_t la = t == 16 && (x >> kRem) == 0 ? prefix_len16(ebuf, c, P) : len;
                                    const uint32_t l2 = min(la, lim);
                                    const uint32_t k = (l2 << 16) | (W - i);
                                    if (l2 >= 2 && k > key) key = k;
                                }
                            }
                            if (okB) {
                                if (t <= 16) {
                                    wrapmaskB |= 1u << (t - 1);
                                } else {
                                    const uint32_t l2 = min(len, min(cap_lenB, W - i));
                                    const uint32_t k = (l2 << 16) | (0xFFFFu - i);
                                    if (l2 >= 2 && k > keyB) keyB = k;
                                }
                            }
                        }
                        // Candidates in the last 15 window positions run past the newest byte, where the ring continues
                        // with the OLDEST byte; t = 1 (the newest byte itself) pairs with the oldest one and is therefore
                        // not in the index at all: test its first byte here.
                        // (a match there needs the oldest byte to equal the pattern's second byte as well)
                        if (prevb == (P[0] & 0xFFu) && ebuf[q] == ((P[0] >> 8) & 0xFFu)) wrapmask |= 2u;
#ifdef TAMP_PROF
                        const uint32_t wrapmask0 = wrapmask;
                        for (uint32_t rep = 0; rep < ((a.dbg & 0x200u) ? 2u : 1u); rep++) {
                        wrapmask = wrapmask0;
#endif
                        while (wrapmask) {
                            const uint32_t t = (uint32_t)__builtin_ctz(wrapmask);
                            wrapmask &= wrapmask - 1;
                            const uint32_t c = q + W - t;
                            const uint32_t i = (e_wp + c) & mask;
                            if (i == mask) continue;
                            const uint32_t lw = prefix_len_wrapped16(ebuf, c, t, W, P);
                            if constexpr (RUNS) wrapbest = max(wrapbest, lw);
                            const uint32_t len = min(lw, min(cap_len, W - i));
                            const uint32_t k = (len << 16) | (W - i);
                            if (len >= 2 && k > key) key = k;
                        }
#ifdef TAMP_PROF
                        }
#endif
                        if constexpr (RUNS) {
                            // Extended matches (compressor.c:437-468, 636-644): a first match longer than min+11 starts
                            // a continuation that ends at "the candidate at or above the first match's index with the
                            // longest common prefix with the input, capped by the window end and min+131"
                            // (Walk::ext_search).  Every such candidate shares the first match's bytes.  A first match
                            // shorter than the 16-byte ring cannot grow at all (no candidate got past it inside the
                            // ring); a 16-byte one that was the only 16-byte hit of the scan has no rival.  Both are
                            // flagged (bit 6 of blen): the walk then just counts the common prefix at that index instead
                            // of searching the window.  Left to the search: several 16-byte hits, hits that run past the
                            // newest window byte, patterns that start inside a listed run.
                            const uint32_t len0 = key >> 16;
                            if (ext && len0 > minp + 11 && !(nruns && P[0] == rep) && (len0 < 16 || (n16 == 1 && wrapbest < 16)))
                                sole_ext = true;
                        }
                    }
                    if (cap_lenB) {
                        {   // oldest byte of the earlier window: outside this query's bracket, test it directly
                            const uint32_t c = q - 1, i = (e_wp + c) & mask;
                            if (i != mask) {
                                const uint32_t l2 = min(prefix_len16(ebuf, c, P), min(cap_lenB, W - i));
                                const uint32_t k = (l2 << 16) | (0xFFFFu - i);
                                if (l2 >= 2 && k > keyB) keyB = k;
                            }
                        }
                        if (ebuf[q + W - 2] == (P[0] & 0xFFu)) wrapmaskB |= 2u;  // newest byte of that window pairs with its oldest
                        while (wrapmaskB) {
                            const uint32_t t = (uint32_t)__builtin_ctz(wrapmaskB);
                            wrapmaskB &= wrapmaskB - 1;
                            const uint32_t c = q - 1 + W - t;
                            const uint32_t i = (e_wp + c) & mask;
                            if (i == mask) continue;
                            const uint32_t l2 = min(prefix_len_wrapped16(ebuf, c, t, W, P), min(cap_lenB, W - i));
                            const uint32_t k = (l2 << 16) | (0xFFFFu - i);
                            if (l2 >= 2 && k > keyB) keyB = k;
                        }
                    }
                    if (lazy && q >= 1) {
                        blen2[q - 1] = (uint8_t)(keyB >> 16);
                        bidx2[q - 1] = (uint16_t)(0xFFFFu - (keyB & 0xFFFFu));
                    }
                    const uint32_t len = key >> 16;
                    // positions where poll_extended_handling does more than fall through (compressor.c:470-503)
                    // (lazy matching: which match a step uses depends on the walk's state, so the length condition is
                    // tested there and only the byte condition is recorded here)
                    bool slow = false;
                    if (ext) {
                        const uint32_t prev = prevb, b0 = P[0] & 0xFFu, b1 = (P[0] >> 8) & 0xFFu;
                        slow = (prev == b0 && (b1 == b0 || R == 1)) || (!lazy && len > minp + 11);
                    }
                    // kSegPartial: no parse step without a full ring -- the chain of plain steps stops where the call ends
                    if (partial && leftq < kRing) slow = true;
                    blen[q] = (uint8_t)(len | (slow ? 0x80u : 0u) | (RUNS && sole_ext ? 0x40u : 0u));
                    bidx[q] = (uint16_t)(W - (key & 0xFFFFu));
                    TAMP_FINE(f2);
                }
                if constexpr (RUNS) {
                    // Second pass, only in epochs with listed runs: interior positions of those runs are not in the
                    // index.  They only matter to a pattern that starts with the run's byte twice.  Let rq = the
                    // pattern's own leading run and, for a run [a, b), rc = b - c the run remainder at candidate c.
                    // rc > rq: the match is exactly rq bytes, for all such c alike; rc < rq: exactly rc bytes, shrinking
                    // as c grows; rc == rq: the one candidate that can go past the run.  With "longest, then lowest
                    // window index" and the limit W - index, the best interior candidate of the run inside the window
                    // is one of: the first one, the one at window index 0, b - rq, b - rq + 1.  Those (at most) four
                    // are compared like any index entry and merged into the result of the first pass (each thread
                    // revisits its own queries).  Interior positions within 15 bytes of the newest window byte run into
                    // the oldest ones like their indexed neighbours (prefix_len_wrapped16).
#ifdef TAMP_PROF
                    if (nruns && !(a.dbg & 0x4000u))  // (0x4000: instruction-count experiments without the second pass)
                    TAMP_REPEAT(0x2000000u) {         // (0x2000000: twice -- the second time it finds its own results -- for its count)
#else
                    if (nruns) {
#endif
                        for (uint32_t j = tid; j < nq; j += nt) {
                            const uint32_t q = sorted[j];
                            const uint32_t b01 = lds_u32_unaligned(ebuf, W + q);
                            const uint32_t x = b01 & 0xFFu;
                            if (((b01 >> 8) & 0xFFu) != x) continue;
                            // (only bytes that HAVE a listed run: "ll", "ee", "ss" start a pattern somewhere in nearly every
                            // wavefront of queries, and one such lane used to take all 64 through the loop over the runs)
                            if (!((rxset[x >> 5] >> (x & 31u)) & 1u)) continue;
                            const uint32_t leftq = n - (e_p0 + q);
                            const uint32_t R = leftq < kRing ? leftq : kRing;
                            if (R < minp) continue;
                            uint32_t P[4];
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) P[jj] = lds_u32_unaligned(ebuf, W + q + 4 * jj);
                            const uint32_t rep = x * 0x01010101u;
                            uint32_t rq = 16;
#pragma unroll
                            for (int jj = 3; jj >= 0; jj--) {
                                const uint32_t xx = P[jj] ^ rep;
                                if (xx) rq = 4u * (uint32_t)jj + ((uint32_t)__builtin_ctz(xx) >> 3);
                            }
                            if (ext && ebuf[W + q - 1] == x && (min(rq, 7u) >= 7u || rq >= R)) continue;  // the RLE path owns it (no scan above either)
                            const uint32_t cap_len = R < maxp ? R : maxp;
                            const uint32_t sv = blen[q];
                            if ((sv & 0x1Fu) == kDeferred) continue;  // (the walk asks)
                            uint32_t key = (sv & 0x1Fu) ? (((sv & 0x1Fu) << 16) | (W - (uint32_t)bidx[q])) : 0u;
                            const uint32_t key0 = key;
                            bool hit16 = false;  // a 16-byte hit the first pass did not see: a rival for an extended match
                            {   // wrap zone: the run-interior positions t = 2..15 bytes before the window's end (straight from the
                                // bitmap of positions the index leaves out: sixteen bits at buffer position q + W - 16) that hold x
                                const uint32_t wz = q + W - 16, sh = wz & 31u;
                                uint32_t rb16 = rbits[wz >> 5] >> sh;
                                if (sh > 16) rb16 |= rbits[(wz >> 5) + 1] << (32u - sh);
                                uint32_t wrapmask = (__builtin_bitreverse32(rb16) >> 15) & 0xFFFCu;  // bit k <-> t = 16 - k
                                while (wrapmask) {
                                    const uint32_t t = (uint32_t)__builtin_ctz(wrapmask);
                                    wrapmask &= wrapmask - 1;
                                    const uint32_t c = q + W - t;
                                    if (ebuf[c] != x) continue;  // (the interior of a run of another byte)
                                    const uint32_t i = (e_wp + c) & mask;
                                    if (i == mask) continue;
                                    // (in ring terms such a position is not "four equal bytes": past the newest byte
                                    // come the oldest ones, so it can match anything -- including all 16 bytes)
                                    const uint32_t lw = prefix_len_wrapped16(ebuf, c, t, W, P);
                                    hit16 = hit16 || lw >= 16;
                                    const uint32_t len = min(lw, min(cap_len, W - i));
                                    const uint32_t k = (len << 16) | (W - i);
                                    if (len >= 2 && k > key) key = k;
                                }
                            }
                            const uint32_t cz = q + ((0u - e_wp - q) & mask);  // the window position with index 0
                            // Per listed run of the pattern's byte, interior candidates lo..hi (inside the window, in
                            // front of its last 15 bytes).  With cs = b - rq, the candidate whose run remainder equals the
                            // pattern's leading run:
                            //   below cs   the match is exactly rq bytes wherever it starts -> the lowest window index
                            //              decides: index 0 (cz) when it lies in the range, else the first one;
                            //   cs itself  goes on behind the run: the two bytes behind the run (kept with the run record)
                            //              against the pattern's two bytes behind ITS run decide whether the buffer has
                            //              to be looked at at all;
                            //   above cs   rc = b - c bytes, shrinking: the first one, or index 0 behind it (a larger
                            //              limit W - index can outweigh the shorter run when the first one is clipped).
                            // All of it is arithmetic on the run record, which is wave-uniform (one LDS read per run and
                            // wavefront, kept on the scalar unit); round 4: Python sources 70 k -> ~25 k VALU
                            // instructions per 4 KiB stream in this pass.
                            const uint32_t qhi = q + W - 16;
                            const uint32_t pb2 = lds_u32_unaligned(ebuf, W + q + (rq < 16 ? rq : 0u)) & 0xFFFFu;
#pragma unroll 1
                            for (uint32_t k = 0; k < nruns; k++) {
                                const uint32_t rv = Walk::uni(runs[k]), rx = Walk::uni(runsx[k]);
                                const uint32_t ra = rv & 0xFFFFu, rb = rv >> 16;
                                const uint32_t lo = max(ra + 1, q), hi = min(rb - 4, qhi);
                                if ((rx & 0xFFu) != x || lo > hi) continue;
                                const int32_t cs = (int32_t)rb - (int32_t)rq;
                                const bool cz_in = cz >= lo && cz <= hi;
                                {   // below cs
                                    const int32_t hiA = min((int32_t)hi, cs - 1);
                                    if ((int32_t)lo <= hiA) {
                                        const uint32_t c = cz_in && (int32_t)cz <= hiA ? cz : lo;
                                        const uint32_t lim_i = W - ((c + e_wp) & mask);
                                        key = max(key, (min(rq, min(cap_len, lim_i)) << 16) | lim_i);
                                    }
                                }
                                if (cs >= (int32_t)lo && cs <= (int32_t)hi) {  // cs itself
                                    const uint32_t x2 = (rx >> 8) ^ pb2;
                                    uint32_t len = rq;
                                    if (rq < 16 && !(x2 & 0xFFu)) len = (x2 >> 8) ? rq + 1 : prefix_len16(ebuf, (uint32_t)cs, P);
                                    const uint32_t lim_i = W - (((uint32_t)cs + e_wp) & mask);
                                    key = max(key, (min(len, min(cap_len, lim_i)) << 16) | lim_i);
                                }
                                {   // above cs
                                    const int32_t c1 = max((int32_t)lo, cs + 1);
                                    if (c1 <= (int32_t)hi) {
                                        const uint32_t lim_i = W - (((uint32_t)c1 + e_wp) & mask);
                                        key = max(key, (min(rb - (uint32_t)c1, min(cap_len, lim_i)) << 16) | lim_i);
                                        if (cz_in && (int32_t)cz > c1) key = max(key, (min(rb - cz, cap_len) << 16) | W);
                                    }
                                }
                            }
                            if (key != key0 || (hit16 && (sv & 0x40u))) {
                                const uint32_t len = key >> 16;
                                const bool slow = (sv & 0x80u) || (ext && len > minp + 11);
                                blen[q] = (uint8_t)(len | (slow ? 0x80u : 0u));
                                bidx[q] = (uint16_t)(W - (key & 0xFFFFu));
                            }
                        }
                    }
                }
                __syncthreads();
                __builtin_amdgcn_s_setprio(kPrioShort);
                TAMP_FINE(f3);
#ifdef TAMP_PROF
                pt[6] += f0, pt[7] += f1, pt[8] += f2, pt[10] += niter;
#endif
                if constexpr (RUNS && PACKED) {
                    // Tokens of the extended format that need no state machine.  A position is flagged slow when
                    // poll_extended_handling does more than fall through there (compressor.c:437-525).  Two of those
                    // cases are decided by the bytes and the match tables alone whenever the walk arrives in the clean
                    // state "everything consumed has been written" -- the only state in which it consults the tables:
                    //  * a run of 2..8 bytes equal to the previous byte that ends inside the input: the RLE token of
                    //    compressor.c:342-359 (or, for runs of 2..6, the ordinary match when that is longer, :490-503);
                    //    8 bytes at most are written for a run (:352-358), so such a token writes what it consumes;
                    //  * a first match longer than min+11 without a rival (bit 6, see the match phase): the extended
                    //    match ends where the common prefix of that window position and the input ends (:297-333,
                    //    377-415), provided it stays in front of the newest window byte.
                    // Both become plain steps of xcnt[q] bytes (bit 5 of blen; length field 1 = RLE token) that the
                    // jump tables chase and the emitter turns into bits, unless the window write would be clipped at the
                    // ring end (W - window_pos bytes, :355 / :404-410): those, long runs, runs reaching the end of the
                    // input and matches with rivals stay with the state machine.
                    if (ext && (kSlowCap * 8 + a_blk) <= (HB == 9 ? kHashBuckets : kBuckets) * 2) {
                        const uint32_t max_ext = minp + 11 + kExtExtraMax;
#ifdef TAMP_PROF
                        // (0x4000000: a dry run in front -- everything but the stores -- for the pass's instruction count)
                        for (uint32_t dry = (a.dbg & 0x4000000u) ? 1u : 0u; dry != 0xFFFFFFFFu; dry--)
#else
                        constexpr uint32_t dry = 0;
#endif
                        for (uint32_t q = e_pending + tid; q < nvalid; q += nt) {
                            const uint32_t sv = blen[q];
                            if (!(sv & 0x80u)) continue;
                            const uint32_t len = sv & 0x1Fu;
                            if (len == kDeferred) continue;  // (no match to weigh the run against: the state machine's)
                            const uint32_t leftq = n - (e_p0 + q);
                            if (partial && leftq < kRing) continue;  // (the call ends in front of this position)
                            const uint32_t wpq = (e_wp + q) & mask;  // window_pos when the walk arrives here clean
                            const uint32_t b4 = lds_u32_unaligned(ebuf, W + q - 1);  // previous byte, then the next three
                            const uint32_t prev = b4 & 0xFFu, b0 = (b4 >> 8) & 0xFFu, b1 = (b4 >> 16) & 0xFFu;
                            if (prev == b0) {
                                if (b1 != b0 || leftq < 2) continue;  // (a single byte at the very end of the input)
                                // run length from q, looked at up to 9 bytes
                                const uint32_t rep = b0 * 0x01010101u;
                                const uint32_t x0 = lds_u32_unaligned(ebuf, W + q + 2) ^ rep, x1 = lds_u32_unaligned(ebuf, W + q + 6) ^ rep;
                                const uint32_t r = 2u + (x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : 4u + (x1 ? (uint32_t)__builtin_ctz(x1) >> 3 : 4u));
                                if (r > kRleWindowMax || r >= leftq || wpq + r > W) continue;
                                if (r <= 6 && len > r) {  // the pattern wins: an ordinary match step (or an extended match: slow)
                                    if (len <= minp + 11 && !dry) blen[q] = (uint8_t)len;
                                    continue;
                                }
                                if (!dry) blen[q] = (uint8_t)(0x20u | 1u), xcnt[q] = (uint8_t)r;
                                continue;
                            }
                            if (!(sv & 0x40u)) continue;
                            const uint32_t idx = bidx[q];
                            const uint32_t off = (idx - wpq) & mask, t0 = W - off;  // candidate = ebuf[q + off ..], t0 bytes in front of the newest one
                            const uint32_t lim = min(min(W - idx, max_ext), min(leftq, t0));
                            uint32_t m = 0;
                            while (m < lim) {
                                const uint32_t x = lds_u32_unaligned(ebuf, q + off + m) ^ lds_u32_unaligned(ebuf, W + q + m);
                                if (x) {
                                    m += (uint32_t)__builtin_ctz(x) >> 3;
                                    break;
                                }
                                m += 4;
                            }
                            const uint32_t cnt = min(m, lim);
                            if (cnt >= t0 || cnt < len || wpq + cnt > W) continue;
                            // (kSegPartial: the reference takes this match a 16-byte ring at a time; the poll that emits
                            // the token must still have had a full ring)
                            if (partial && leftq < cnt + kRing) continue;
                            if (!dry) blen[q] = (uint8_t)(0x20u | len), xcnt[q] = (uint8_t)cnt;
                        }
                    }
                    __syncthreads();
                }
                }  // (!have_tables)
                for (uint32_t k = 4 + tid; k < 4 + a_blk / 2 && k < L.obuf_words; k += nt) obuf[k] = 0;  // scan starts out
                // Jump tables for the walk (the index is dead now, its space is reused).  Within each 64-position
                // block, lane = position: six rounds of pointer doubling over ds_bpermute give, for every position, where
                // the chain of plain steps starting there leaves the block (or the "slow" position it stops at) and how
                // many tokens it emits on the way.
                // Lazy matching (compressor.c:576-619): a step starts either fresh (state 0: best match A[p] =
                // blen/bidx) or from the match cached by the previous step's probe (state 1: B[p-1] = blen2/bidx2).  It
                // emits its match / literal, or -- when the probe B[p] beats it and does not overlap the byte about to
                // be written -- a literal, and hands B[p] to the next position.  That is a deterministic transition on
                // the 2 x nvalid states v = 2p + state: it is tabulated here and chased with the same pointer doubling.
                if constexpr (LAZY) {
                    for (uint32_t v = tid; v < nv + 128 && v < 2 * a_blk + 128; v += nt) {
                        uint32_t out = 0x80u;  // stop: the state machine takes this step
                        const uint32_t pp = v >> 1, s1 = v & 1u;
                        if (v < nv && pp + 1 < nvalid && !(s1 && pp == 0)) {
                            const uint32_t sva = blen[pp];
                            const uint32_t len = s1 ? (uint32_t)blen2[pp - 1] : (sva & 0x1Fu);
                            if (!(sva & 0x80u) && !(ext && len > minp + 11)) {
                                const uint32_t leftp = n - (e_p0 + pp);
                                const uint32_t R = leftp < kRing ? leftp : kRing;
                                bool defer = false;
                                if (len >= minp && len <= 8 && R > len + 2) {
                                    const uint32_t nlen = blen2[pp], nidx = bidx2[pp];
                                    const uint32_t wpos = (e_wp + pp) & mask;  // clean chain: everything consumed is written
                                    defer = nlen > len && (wpos < nidx || wpos >= nidx + nlen);  // validate_no_match_overlap, :185-188
                                }
                                out = defer ? 3u - s1 : 2u * (len >= minp ? len : 1u) - s1;  // v' - v
                            }
                        }
                        vstep[v] = (uint8_t)out;
                    }
                    __syncthreads();
                }
                TAMP_REPEAT(0x40000u)
                for (uint32_t b = wave * 64; b < nv; b += (nt >> 6) * 64) {
                    const uint32_t sv = steps[b + lane];  // sentinels (0x80) beyond the last state
                    const bool slowp = (sv & 0x80u) != 0;
                    uint32_t stepv = LAZY ? (sv & 0x7Fu) : (sv >= minp ? (sv & 0x1Fu) : 1u);
                    if constexpr (RUNS && PACKED) {
                        if (sv & 0x20u) stepv = xcnt[b + lane];  // RLE run / extended match settled above
                        // Step table for the token listing (round 4): every position leaves the bytes its step consumes
                        // in xcnt (settled tokens have theirs there already).  The listing runs in ONE wavefront with
                        // three waiting for it and what it costs is the instructions it issues: with one byte read per
                        // token instead of the step byte, its flag tests and a second read for settled tokens, the
                        // kernel is 2.2 % faster on the synthetic text and 1.6 % on prose
                        // (profiles/ab/r4_listing_step_table.log; two tokens per round trip on top: no further gain).
                        if (b + lane < nv) xcnt[b + lane] = (uint8_t)stepv;
                    }
                    // packed: target * 4 (the byte offset ds_bpermute wants) | count << 10 (| bits << 19 in block mode).  Round 6:
                    // a chain ENDS on a lane -- a slow position, or one whose step leaves the block -- and such a lane points at
                    // itself with a count of zero, so a round is a fixed point for it and needs no "finished" select: fetch the
                    // target's state, add the own count onto it (three VALU operations and the bpermute where there were six).
                    // What the end lane contributes -- its exit target and its own token, or itself and nothing -- is fetched
                    // once behind the six rounds.
                    constexpr uint32_t kCnt = (0xFFu << 10) | (BLOCKM ? 0xFFF80000u : 0u);
                    const uint32_t tgt = (uint32_t)lane + stepv;
                    const bool endp = slowp || tgt >= 64u;
                    uint32_t own = 1u << 10;  // this position's token: one, and in block mode its bits
                    if constexpr (BLOCKM) {
                        // (a 64-position block's tokens take 64 x 9 + 24 bits at most: literal 1 + 8, match prefix code + window bits
                        // per two positions or more)
                        const uint32_t lenv = sv & 0x1Fu;
                        own |= (lenv >= minp ? tok_nbits(lenv - minp) + wbits : lbits + 1u) << 19;
                    }
                    uint32_t st = endp ? ((uint32_t)lane << 2) : ((tgt << 2) | own);
                    const uint32_t endv = slowp ? ((uint32_t)lane << 2) : ((tgt << 2) | own);  // (read only from end lanes)
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const uint32_t o2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(st & 0x3FFu), (int)st);
                        st = o2 + (st & kCnt);
                    }
                    {
                        const uint32_t e = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(st & 0x3FFu), (int)endv);
                        st = e + (st & kCnt);
                    }
                    if constexpr (BLOCKM) {
                        if (a.block_pass == 1 && b + lane < nv) toklist[b + lane] = (uint16_t)(st >> 19);  // (pass 1 lists no tokens: the space is free)
                    }
                    st = ((st & 0x3FFu) >> 2) | (((st >> 10) & 0xFFu) << 8);  // target | count << 8 for the stores below
                    if (b + lane < nv) {
                        if constexpr (LAZY) {
                            jump16[b + lane] = (uint16_t)(b + (st & 0xFFu));
                            count8[b + lane] = (uint8_t)((st >> 8) & 0xFFu);
                        } else {
                            jc32[b + lane] = (b + (st & 0xFFu)) | (((st >> 8) & 0xFFu) << 16);
                        }
                    }
                }
                __syncthreads();
                TAMP_PROF_MARK(2);
            }
            if constexpr (BLOCKM) {
                if (a.block_pass == 1) {
                    // per entry offset (the previous block's last token reaches up to 14 bytes into this one): where the chain
                    // of tokens leaves the block and how many bits it takes -- sixteen dependent reads per lane, 15 lanes
                    if (tid < 16) {
                        uint32_t pos = tid, bits = 0;
                        bool ok = tid < 15;
                        while (ok && pos < nvalid) {
                            const uint32_t jc = jc32[pos], j = jc & 0xFFFFu;
                            if (j == pos) { ok = false; break; }  // (a position the state machine would take: none in the v1 format)
                            bits += toklist[pos];
                            pos = j;
                        }
                        a.blk_table[(size_t)s * 16 + tid] = ok ? ((pos - nvalid) | (bits << 4)) : 0xFFFFFFFFu;
                    }
                    if (a.blk_len) {  // the match results, for pass 3
                        uint8_t* const gl = a.blk_len + bpos;
                        uint16_t* const gi = a.blk_idx + bpos;
                        for (uint32_t k = tid * 4; k < nvalid; k += nt * 4) {
                            *reinterpret_cast<uint32_t*>(gl + k) = *reinterpret_cast<const uint32_t*>(blen + k);
                            *reinterpret_cast<uint2*>(gi + k) = *reinterpret_cast<const uint2*>(bidx + k);
                        }
                    }
                    break;
                }
                // Entry offset 15 is no offset: pass 1 writes 0xFFFFFFFF for a chain that met a position the state machine would
                // take (none in the v1 format), the scan follows that word to entry 15 and stays there (row 15 of every table
                // is 0xFFFFFFFF as well).  Should the invariant ever break, the stream ends with TAMP_ERROR instead of bytes.
                if (((uint32_t)binfo & 15u) == 15u) {
                    if (s == a.n_blocks - 1 && tid == 0) a.out_len[0] = 0, a.status[0] = kError;
                    break;
                }
                wk.rd = wk.wr = (uint32_t)binfo & 15u;  // pass 3: the walk starts where the previous block's last token ended
            }

            // ---------------- walk: wave 0 ----------------
            if (wave == 0) {
                // The walk is a chain of dependent steps in ONE wavefront while the other three of the workgroup wait for
                // it, and it shares its SIMD with five wavefronts of other workgroups: let the issue arbiter prefer it
                // (kPrioWalk; see kPrioScan for the scheme).
                __builtin_amdgcn_s_setprio(kPrioWalk);
#ifdef TAMP_PROF
                // (0x8000000: the walk's main loop TWICE from the same state, for its instruction count -- what it writes
                // to LDS is the same both times, positions matched on demand are marked "deferred" again in between;
                // with 0x10000000 the second run skips the token listing: the difference is the listing's count)
                const Walk wk_save = wk;
                uint32_t walk_reps = (a.dbg & 0x8000000u) ? 2u : 1u, ndef = 0;
                bool walk_second = false;
            walk_again:
#endif
                wk.nvalid = nvalid;
                wk.ntok = 0, wk.ns = 0;
                uint32_t act = 0, excess_tok = 0xFFFFFFFFu;
                uint32_t nqueued = 0;  // blocks whose tokens are still to be listed
                uint32_t segv = 0;     // lane k: first position | first token slot << 16 of queued block k
#ifdef TAMP_PROF
                unsigned long long dbg_list = 0, dbg_hop = 0, dbg_calls = 0;
#endif
                auto list_queued = [&]() {
#ifdef TAMP_PROF
                    const unsigned long long lt0 = __builtin_readcyclecounter();
                    dbg_calls++;
#endif
                    __builtin_amdgcn_wave_barrier();
#ifdef TAMP_PROF
                    if ((a.dbg & 0x10000000u) && walk_second) { nqueued = 0; return; }
                    // (0x200000: the listing without its chain of dependent reads -- valid positions, wrong tokens: the
                    // time the chain costs)
                    if ((a.dbg & 0x200000u) && (uint32_t)lane < nqueued) {
                        uint32_t pp = segv & 0xFFFFu, slot = segv >> 16;
                        for (uint32_t cleft = LAZY ? (uint32_t)count8[pp] : jc32[pp] >> 16; cleft; cleft--) toklist[slot++] = (uint16_t)pp++;
                    } else
#endif
                    if ((uint32_t)lane < nqueued) {
                        uint32_t pp = segv & 0xFFFFu;
                        uint32_t slot = segv >> 16;
                        for (uint32_t cleft = LAZY ? (uint32_t)count8[pp] : jc32[pp] >> 16; cleft; cleft--) {
                            toklist[slot++] = (uint16_t)pp;
                            if constexpr (RUNS && PACKED) {  // (the step table of the jump phase)
                                pp += xcnt[pp];
                                continue;
                            }
                            const uint32_t sv = steps[pp];
                            if (RUNS && PACKED && (sv & 0x20u))
                                pp += xcnt[pp];
                            else
                                pp += LAZY ? sv : (sv >= minp ? sv : 1u);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    nqueued = 0;
#ifdef TAMP_PROF
                    dbg_list += __builtin_readcyclecounter() - lt0;
#endif
                };
                for (;;) {
                    // (round 5: the walk's scalars are changed inside `if (wave == 0)` -- a divergent branch to the compiler --
                    // and come round the epoch loop as vector values: the hop loop and most of the state machine were
                    // v_cmp + s_and_saveexec + s_cbranch_execz on VGPR copies.  Pinned back to SGPRs once per iteration:
                    // 1.5 k -> 1.35 k VALU in the step's code, the hop loop 5 VALU + 1 LDS read a hop instead of 12 + three
                    // exec-mask regions; synthetic -0.9 %, prose -1.8 %, markup -2.7 %, Python sources -1 %.  Deriving `wave`
                    // itself from the scalar copy frees the branch as well but costs 33 more spilled SGPRs: no better.)
                    wk.ntok = Walk::uni(wk.ntok), wk.ns = Walk::uni(wk.ns), wk.rd = Walk::uni(wk.rd), wk.wr = Walk::uni(wk.wr);
                    wk.rle_count = Walk::uni(wk.rle_count), wk.ext_count = Walk::uni(wk.ext_count), wk.ext_pos = Walk::uni(wk.ext_pos);
                    nqueued = Walk::uni(nqueued), wk.wp_e = Walk::uni(wk.wp_e), w_p0 = Walk::uni(w_p0), wk.nvalid = Walk::uni(wk.nvalid);
                    if (wk.ntok + 72 > L.tokcap || wk.ns + 8 > kSlowCap) {
                        act = kActContinue;
                        break;
                    }
                    const bool clean = wk.wr == wk.rd && wk.rle_count == 0 && wk.ext_count == 0;
                    // (lazy: a match cached before this epoch began has no table entry -- the state machine takes that step)
                    if (clean && wk.rd < nvalid && !(LAZY && wk.lazy_valid && wk.rd == e_pending)) {
                        // Plain steps: hop from block to block through the jump tables (one dependent LDS read per 64
                        // positions / states), then let one lane per block list that block's tokens.
                        uint32_t pos = LAZY ? 2 * Walk::uni(wk.rd) + (wk.lazy_valid ? 1u : 0u) : Walk::uni(wk.rd);
                        // The blocks passed are queued (first position, first token slot); their tokens are listed
                        // later, many blocks at a time (list_queued): a listing pass costs as many dependent LDS
                        // round trips as the fullest block has tokens, however few blocks it serves.
                        uint32_t total = 0, nhop = 0;
#ifdef TAMP_PROF
                        const unsigned long long ht0 = __builtin_readcyclecounter();
#endif
                        while (pos < nv && nqueued < 64 && wk.ntok + total + 64 <= L.tokcap) {
                            uint32_t j, cpos;
                            if constexpr (LAZY) {
                                const uint32_t jv = jump16[pos], cv = count8[pos];  // both reads in flight: one LDS round trip
                                j = Walk::uni(jv), cpos = Walk::uni(cv);
                            } else {
                                const uint32_t jc = Walk::uni(jc32[pos]);
                                j = jc & 0xFFFFu, cpos = jc >> 16;
                            }
                            if (j == pos) break;  // a position the state machine has to look at
                            segv = (uint32_t)lane == nqueued ? (pos | ((wk.ntok + total) << 16)) : segv;  // lane k keeps block k
                            nqueued++, nhop++;
                            total += cpos;
                            pos = j;
                        }
#ifdef TAMP_PROF
                        dbg_hop += __builtin_readcyclecounter() - ht0;
#endif
                        if (nqueued > 64 - 34) list_queued();  // room for the next chain of hops (at most blk / 64 = 32 blocks)
                        wk.ntok += total;
                        if constexpr (LAZY) {
                            wk.rd = wk.wr = pos >> 1;
                            wk.lazy_valid = (pos & 1u) != 0;
                            if (wk.lazy_valid) {
                                wk.lazy_len = Walk::uni(blen2[wk.rd - 1]);
                                wk.lazy_idx = Walk::uni(bidx2[wk.rd - 1]);
                            }
                        } else {
                            wk.rd = wk.wr = pos;
                        }
                        if (nhop) continue;
                    }
                    const uint32_t p = w_p0 + wk.rd;
                    if (partial ? n - p >= kRing : p < n) {
                        const uint32_t pending = wk.rle_count + wk.ext_count;
                        int r = Walk::kStepRebase;
                        if (wk.rd <= cur_blk + pending) {
                            // (kSegPartial: exactly one poll -- the ring holds 16 bytes and nothing is known beyond it)
                            const uint32_t leftp = partial ? kRing : n - p;
#ifdef TAMP_PROF
                            const unsigned long long t0 = __builtin_readcyclecounter();
                            const uint32_t ec0 = wk.ext_count;
#endif
                            if constexpr (!LAZY) {
                                if (ext && wk.wr == wk.rd && wk.rd < nvalid && (Walk::uni(blen[wk.rd]) & 0x1Fu) == kDeferred) {
#ifdef TAMP_PROF
                                    if constexpr (RUNS) {
                                        if (lane == 0 && ndef < kRunCap) runs[ndef] = wk.rd;  // (the run list is dead during the walk)
                                        ndef++;
                                    }
#endif
                                    wk.best_on_demand(leftp < kRing ? leftp : kRing);  // (a position the match phase left out)
                                }
                            }
                            r = wk.template step<RUNS>(leftp < kRing ? leftp : kRing, leftp);
#ifdef TAMP_PROF
                            pt[11] += 1;
                            if (ec0) pt[5] += __builtin_readcyclecounter() - t0;  // time in extended-match continuation steps
                            else pt[9] += __builtin_readcyclecounter() - t0;      // other slow steps
                            if (ec0) pt[14] += 1;                         // extended matches that went through the search
                            if (!ec0 && wk.ntok >= 3 && wk.ns >= 3 && wk.last_ext_direct) pt[15] += 1;  // ... settled without one
                            wk.last_ext_direct = false;
#endif
                        }
                        if (r == Walk::kStepRebase) {
                            act = kActRebase;
                            break;
                        }
                        if (r == Walk::kStepExcess) {
                            excess_tok = wk.ntok;
                            act = kActDone;
                            break;
                        }
                    } else if (!partial && ext && wk.rle_count >= 1) {  // compressor.c:748-763
                        if (wk.rle_count == 1) {
                            const uint32_t c = wk.win((wk.wp() - 1) & mask);  // uniform (readfirstlane inside)
                            wk.put((1u << lbits) | c, lbits + 1u);
                            wk.append(1, wk.wr + 1 == wk.rd, [&](uint32_t) { return c; });
                        } else {
                            wk.emit_rle(wk.rle_count);
                        }
                        wk.rle_count = 0;
                    } else if (!partial && ext && wk.ext_count) {  // compressor.c:764-766
                        wk.emit_ext();
                    } else {
                        if (st_io && (a.seg_flags & kSegSave)) {  // hand the window back in ring order
                            for (uint32_t i = lane; i < W; i += kWave) st_io[i] = (uint8_t)wk.win_l(i);
                            if (lane == 0) {
                                st_io[W] = (uint8_t)wk.wp();
                                st_io[W + 1] = (uint8_t)(wk.wp() >> 8);
                                // what stays pending when the call ends without a flush (zero after a drain)
                                st_io[W + 3] = (uint8_t)wk.rle_count, st_io[W + 4] = (uint8_t)wk.ext_count;
                                st_io[W + 6] = (uint8_t)wk.ext_pos, st_io[W + 7] = (uint8_t)(wk.ext_pos >> 8);
                                for (uint32_t k = 0; k < 4; k++) st_io[W + 9 + k] = (uint8_t)(p >> (8 * k));
                            }
                        }
                        act = kActDone;
                        break;
                    }
                }
                list_queued();
#ifdef TAMP_PROF
                if (--walk_reps) {
                    if constexpr (RUNS) {
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0)
                            for (uint32_t k = 0; k < ndef && k < kRunCap; k++) blen[runs[k]] = (uint8_t)(0x80u | kDeferred);
                        __builtin_amdgcn_wave_barrier();
                    }
                    ndef = 0;
                    wk = wk_save;
                    walk_second = true;
                    goto walk_again;
                }
                if (a.dbg & 0x2000u) wk.dbg_lag_rle += (uint32_t)dbg_hop, wk.dbg_lag_ext += (uint32_t)dbg_list, wk.dbg_lag_rle_short += (uint32_t)dbg_calls;
#endif
                if (act == kActRebase) {
                    // drop the lag, keep the bytes a pending RLE run / extended match has consumed but not
                    // written (oracle/tamp_model.c m_epoch_begin)
                    const uint32_t pending = wk.rle_count + wk.ext_count;
                    const uint32_t shift = wk.wr;
                    // broke inside the block (lag) -> halve it, 512 positions at least; ran off its end -> double it
                    // (measured on 16,384 x 4 KiB: quarter / 256 prose 12.75, Python 5.72 GB/s; halve / 512: 13.64, 5.76; halve / 1024: 13.15,
                    // 5.42; never shrink: 11.62, 4.81)
                    const bool broke = wk.wr + pending != wk.rd;  // bytes were consumed that will never be written
                    uint32_t nb2 = broke ? max(cur_blk >> TAMP_BRK_SHIFT, (uint32_t)TAMP_BRK_MIN) : min(cur_blk << 1, a_blk);
                    nb2 = min(nb2, a_blk);
                    if (lane == 0) ctl[cBlk] = nb2;
                    // The cut was a guess: when the walk arrives at the end of a shortened block without a lag, a match
                    // carried it through the run and the cut only cost an index build -- this stream's runs have to be
                    // twice as long from now on.  (Source code: indentation repeats the line above; prose: rules and
                    // table borders mostly do not.)
                    if (nvalid < nplan && !broke && wk.rd >= nvalid && lane == 0)
                        ctl[cCutThr] = min(2u * (uint32_t)ctl[cCutThr], 64u);
                    wk.wp_e = wk.wp();
                    w_p0 += wk.rd - pending;
                    wk.wr = 0;
                    wk.rd = pending;
                    if (lane == 0) {
                        ctl[cShift] = shift;
                        ctl[cP0] = w_p0;
                        ctl[cPending] = pending;
                        ctl[cWp] = wk.wp_e;
                    }
                }
                if (lane == 0) {
                    ctl[cAct] = act;
                    ctl[cNtok] = wk.ntok;
                    ctl[cExcess] = excess_tok;
                    ctl[Walk::kCoopCmd] = Walk::kCmdEnd;
                }
                __builtin_amdgcn_s_setprio(kPrioShort);
            } else if (LOOP && tid == 3 * kWave) {
                // the other wavefronts wait for the walk: the last one fetches the workgroup's next claim meanwhile, if this
                // stream is the last of the current one (a fetch from the one counter all workgroups share takes
                // microseconds; behind the stream, nothing of the next one could start before it)
                if (claim[0] == claim[1] && claim[3] == kNoClaim) claim[3] = atomicAdd(a.work_counter, a.claim);
            }
            // (256-thread workgroups: wavefronts 1-3 serve the walk's searches until it posts the end; the barrier that
            // ends their service is the one wavefront 0 executes here)
            if (wk.coop && wave != 0)
                wk.serve(wave);
            else
                __syncthreads();
            TAMP_PROF_MARK(3);

            // ---------------- emit: token list -> bits (all threads) ----------------
#ifdef TAMP_PROF
            uint32_t emit_reps = (a.dbg & 0x80000u) ? 2u : 1u;
        emit_again:
#endif
            asm volatile("" : "+v"(tid));
            lane = (int)(tid & (kWave - 1)), wave = tid >> 6, wk.lane = lane;  // (re-derived: see above)
            uint32_t act = Walk::uni(ctl[cAct]);
            const uint32_t ntok = Walk::uni(ctl[cNtok]);
            const uint32_t K = (ntok + nt - 1) >> nt_log2;
            const uint32_t k0 = min(tid * K, ntok), k1 = min(k0 + K, ntok);
            auto token = [&](uint32_t k, uint32_t& v, uint32_t& nb) -> bool {  // false: literal with excess bits
                const uint32_t e = toklist[k];
                if (e & 0x8000u) {
                    v = stok[2 * (e & 0x7FFFu)];
                    nb = stok[2 * (e & 0x7FFFu) + 1];
                    return true;
                }
                uint32_t pos = e, len, idx;
                if (LAZY) {  // e = 2 * position + state; the step deferred (literal now) iff it leads to a cached state
                    pos = e >> 1;
                    const bool cached = (e & 1u) != 0;
                    const bool deferred = ((e + vstep[e]) & 1u) != 0;
                    len = deferred ? 0u : (cached ? (uint32_t)blen2[pos - 1] : (blen[pos] & 0x1Fu));
                    idx = cached ? (uint32_t)bidx2[pos - 1] : (uint32_t)bidx[pos];
                } else {
                    len = blen[pos] & 0x1Fu, idx = bidx[pos];
                    if constexpr (RUNS && PACKED) {
                        if (blen[pos] & 0x20u) {  // tokens settled by the match phase: xcnt[pos] bytes
                            const uint32_t cnt = xcnt[pos];
                            if (len == 1) {  // write_rle_token, compressor.c:342-359: symbol 12, then count - 2 (4 trailing bits)
                                const uint32_t val = cnt - 2, ci = val >> 4;
                                const uint32_t hn = tok_nbits(ci) - 1 + 4;
                                v = ((uint32_t)codetab[kSymRle] << hn) | ((uint32_t)codetab[ci] << 4) | (val & 15u);
                                nb = tok_nbits(kSymRle) + hn;
                            } else {  // write_extended_match_token, :377-415: symbol 13, size - min - 12 (3 trailing bits), position
                                const uint32_t val = cnt - minp - 12, ci = val >> 3;
                                const uint32_t hn = tok_nbits(ci) - 1 + 3;
                                v = ((((uint32_t)codetab[kSymExt] << hn) | ((uint32_t)codetab[ci] << 3) | (val & 7u)) << wbits) | idx;
                                nb = tok_nbits(kSymExt) + hn + wbits;
                            }
                            return true;
                        }
                    }
                }
                if (len < minp) {  // compressor.c:625-632
                    const uint32_t c = ebuf[W + pos];
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
                        atomicMin((uint32_t*)&ctl[cExcess], k);
                        break;
                    }
                __syncthreads();
            }
            const uint32_t limit = min((uint32_t)ctl[cExcess], ntok);
            const bool excess = ctl[cExcess] != 0xFFFFFFFFu;
            uint32_t mybits = 0;
#ifndef TAMP_EMIT_TWICE
            // (round 6: a thread's tokens -- three for a 1,024-position block of text -- are put together once and kept in registers for the
            // scatter below instead of being put together a second time there: synthetic 4.92 -> 4.86 ms, real text -0.3 .. -1 %)
            const bool tok_cached = K <= 4;
            uint32_t cv[4] = {0, 0, 0, 0}, cn[4] = {0, 0, 0, 0};
            if (tok_cached) {
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t k = k0 + j;
                    if (k < k1 && k < limit) token(k, cv[j], cn[j]);
                    mybits += cn[j];
                }
            } else
#endif
            for (uint32_t k = k0; k < k1 && k < limit; k++) {
                uint32_t v, nb;
                token(k, v, nb);
                mybits += nb;
            }
            const uint32_t incl = wave_scan_add(mybits);
            if (lane == kWave - 1) ctl[cWave + wave] = incl;
            __syncthreads();
            uint32_t o = carry + incl - mybits, segbits = 0;
            for (uint32_t w2 = 0; w2 < (nt >> 6); w2++) {
                const uint32_t wt = ctl[cWave + w2];
                if (w2 < wave) o += wt;
                segbits += wt;
            }
            {  // MSb-first scatter of this thread's contiguous run of tokens
                uint32_t wi = o >> 5, ph = o & 31, fill = 0;
                uint64_t acc = 0;
                auto put_bits = [&](uint32_t v, uint32_t nb) {
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
                };
#ifndef TAMP_EMIT_TWICE
                if (tok_cached) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++)
                        if (cn[j]) put_bits(cv[j], cn[j]);
                } else
#endif
                for (uint32_t k = k0; k < k1 && k < limit; k++) {
                    uint32_t v, nb;
                    token(k, v, nb);
                    put_bits(v, nb);
                }
                if (fill) {
                    const uint32_t w = ((uint32_t)acc & ((1u << fill) - 1)) << (32 - ph - fill);
                    atomicOr(&obuf[wi], __builtin_bswap32(w));
                }
            }
            __syncthreads();
            uint32_t tot = carry + segbits;  // bits now in obuf
            if (excess) act = kActDone;
            if (act == kActDone && !excess && (a.seg_flags & kSegFlushToken)) {
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
            // flush whole words (or, at the end, the zero-padded / truncated byte count) to HBM
            uint32_t nbytes;
            if (act == kActDone)
                nbytes = (excess || partial) ? (tot >> 3) : ((tot + 7) >> 3);  // compressor.c:629-631 / :799-807
            else
                nbytes = (tot >> 5) << 2;
            if (act == kActDone && st_io && (a.seg_flags & kSegSave) && tid == 0) {
                // the bits of the last, incomplete byte stay with the stream (partial_flush writes whole bytes only,
                // compressor.c:65-75); after a drain the byte was padded and written: nothing is carried
                const uint32_t nb = (partial && !excess) ? (tot & 7u) : 0u;
                st_io[W + 5] = (uint8_t)nb;
                st_io[W + 16] = 0, st_io[W + 17] = 0, st_io[W + 18] = 0;
                st_io[W + 19] = nb ? reinterpret_cast<const uint8_t*>(obuf)[tot >> 3] : 0;
            }
            if constexpr (BLOCKM) {
                // Block mode: the bit buffer's word 0 is word (bit position >> 5) of the stream's output; this block owns the
                // bytes [c0, c1) of it -- the first and the last one possibly together with a neighbour, whose bits are zero
                // here.  Aligned dwords of the destination: the first and the last are ORed in atomically (the output was
                // zeroed before the launch), the ones in between are stored.  Nothing is written at or behind `cap`.
                const uint8_t* ob = reinterpret_cast<const uint8_t*>(obuf);
                const unsigned long long base = ((binfo >> 4) >> 5) << 2;  // output byte index of ob[0]
                const uint32_t c0 = s == 0 ? 0u : carry >> 3, c1 = (tot + 7) >> 3;
                const uint32_t c1c = base + c1 <= cap ? c1 : (base >= cap ? 0u : (uint32_t)(cap - base));
                if (c1c > c0) {
                    uint8_t* const P = gout + base;
                    const uintptr_t A0 = reinterpret_cast<uintptr_t>(P + c0) & ~(uintptr_t)3, A1 = reinterpret_cast<uintptr_t>(P + c1c - 1) & ~(uintptr_t)3;
                    const uint32_t ndw = (uint32_t)((A1 - A0) >> 2) + 1;
                    for (uint32_t j = tid; j < ndw; j += nt) {
                        // (an explicitly GLOBAL pointer: built from an integer it would be a generic one, and the store / atomic FLAT)
                        typedef __attribute__((address_space(1))) uint32_t GlobalWord;
                        GlobalWord* const dst = (GlobalWord*)(A0 + 4 * (uintptr_t)j);
                        const int32_t i0 = (int32_t)((intptr_t)(A0 + 4 * (uintptr_t)j) - reinterpret_cast<intptr_t>(P));  // ob index of its first byte
                        uint32_t v = 0;
                        if (i0 >= (int32_t)c0 && i0 + 4 <= (int32_t)c1c) {
                            v = lds_u32_unaligned(ob, (uint32_t)i0);
                        } else {
                            for (int32_t x = 0; x < 4; x++)
                                if (i0 + x >= (int32_t)c0 && i0 + x < (int32_t)c1c) v |= (uint32_t)ob[i0 + x] << (8 * x);
                        }
                        if (j == 0 || j == ndw - 1) __hip_atomic_fetch_or(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else *dst = v;
                    }
                }
                if (act == kActDone && tid == 0) {
                    const unsigned long long total_bytes = base + c1;
                    a.out_len[0] = total_bytes < cap ? (uint32_t)total_bytes : cap;
                    a.status[0] = total_bytes > cap ? kOutputFull : kOk;
                }
                break;  // (one epoch per block)
            }
            {   // HBM stores are whole aligned dwords whatever the slab's byte alignment: a few head bytes, then
                // dwords funnel-shifted out of the bit buffer, then the tail bytes
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
#ifdef TAMP_PROF
            if (--emit_reps) {  // (ORs the same bits into the same words and stores the same bytes)
                __syncthreads();
                goto emit_again;
            }
#endif
            if (act == kActDone) {
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
            TAMP_PROF_MARK(4);
            need_match = act == kActRebase;
            if (act == kActRebase) {
                const uint32_t shift = Walk::uni(ctl[cShift]);
                e_p0 = Walk::uni(ctl[cP0]);
                e_pending = Walk::uni(ctl[cPending]);
                e_wp = Walk::uni(ctl[cWp]);
                cur_blk = Walk::uni(ctl[cBlk]);
                // re-base: ebuf[0..W) <- ebuf[shift..shift+W) (moving left, chunked)
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
            __syncthreads();
        }
#ifdef TAMP_PROF
        if (tid == 0 && a.prof) {
            if (a.dbg & 0x3000u) pt[6] = wk.dbg_lag_rle, pt[7] = wk.dbg_lag_ext, pt[8] = wk.dbg_lag_rle_short;  // lag causes instead of the fine timers
            for (int i = 0; i < 16; i++) atomicAdd(&a.prof[i], pt[i]);
        }
#endif
        if constexpr (LOOP) {
            // the next stream of this workgroup's current claim, or the first of the next claim of a.claim consecutive
            // streams (1 for long streams; short ones are claimed sixteen at a time: fewer fetches from the one counter).
            // The next claim was fetched during the walk (above), or is now (one-wavefront workgroups, streams without a
            // walk).  The claim lives in LDS: nothing of it stays in registers across the stream.
            if (tid_k == 0) {
                uint32_t nx = claim[0];
                if (nx == claim[1]) {
                    nx = claim[3];
                    if (nx == kNoClaim) nx = atomicAdd(a.work_counter, a.claim);
                    claim[1] = nx + a.claim;
                    claim[3] = kNoClaim;
                }
                claim[0] = nx + 1;
                claim[2] = nx;
            }
        }
        __syncthreads();  // ctl / LDS reuse by the next stream; the next stream's number
        if constexpr (!LOOP) break;
    }
}

}  // namespace tamp_amd
