// tamp_decompress_split_kernel.hpp -- batch `.tamp` decoder for gfx950 in two kernels: parse, then resolve.
//
// Replaces, per stream, tamp_decompressor_init(conf=NULL) + tamp_decompressor_decompress
// (tamp/_c_src/tamp/decompressor.c:331-347,371-578), like tamp_decompress_kernel.hpp -- same statuses, sizes and consumed
// counts -- but splits the work where the dependencies are:
//
//   * PARSE (tamp_decode_parse_kernel, one lane per stream, no window, no LDS): the bit-serial part.  It is the
//     reference's token loop (bit buffer, refill rule, prefix code, RLE / extended-match payloads, FLUSH, the out-of-bounds
//     rule, partial tokens when the output fills up: decompressor.c:357-365,431-575) with every data movement taken out:
//     what it leaves behind per token is a 32-bit record (kind, bytes produced, window offset or literal), per stream the
//     final status / size / consumed count, and a short list of the tokens that wrote fewer bytes to the window than they
//     produced (RLE runs over 8 bytes, tokens clipped at the ring end: decompressor.c:140-173,229-272).  With no window
//     to keep per lane the kernel is bound by instruction issue, not by LDS capacity or cache-resident windows as the
//     lane-per-stream decoders are (13-16 % VALU busy, tools/dec_pmc2.sh).
//   * RESOLVE (tamp_decode_resolve_kernel, one workgroup per stream): the data part, parallel over the stream's bytes.
//     Every byte ever written to the window has a virtual position v (the dictionary sits at v = -W .. -1); ring index i
//     holds, when V bytes have been written, the byte with v = V - 1 - ((V - 1 - i) mod W).  A copy token therefore names,
//     for each byte it produces, an EARLIER output byte or a dictionary byte -- sources are read in the window as it was
//     before the token (decompressor.c:564-572, tamp_window_copy common.c:58-86: memmove semantics), so the pointers run
//     strictly backwards.  The records are expanded to one pointer per output byte, the pointers are resolved by pointer
//     jumping (each round halves every chain), and the finished bytes leave as coalesced dwords.  No window exists at all,
//     so any window size costs the same.
//
// Streams the two kernels do not cover are flagged by PARSE and decoded by the lane / wave decoders afterwards
// (DecompressArgs::only_flagged): a dictionary reset inside the stream (double FLUSH, decompressor.c:501-514), more
// tokens or lagging tokens than the scratch slots hold.  Scalar model: the oracle's decoder (oracle/tamp_oracle.c); parity
// and status / consumed semantics are tested against it for every decoder (tests/test_gpu_parity.py, tools/fuzz_gpu.py).
#pragma once
#include <type_traits>
#include "tamp_common.hpp"
#include "tamp_decompress_kernel.hpp"

namespace tamp_amd {

constexpr uint32_t kSplitMaxLag = 48;      // lagging tokens listed per stream (more: the stream is left to the lane decoders)
constexpr uint32_t kSplitWaveMaxOut = 2048;  // out_cap up to which RESOLVE runs one wavefront per stream (2 KiB: 1.25 against 1.31 ms for 131,072 streams; 4 KiB: 1.95 against 1.47)
constexpr uint32_t kSplitMaxOut = 16384;   // bytes of output RESOLVE keeps in LDS (out_cap above: not a split-decoder batch)
__host__ __device__ constexpr uint32_t split_resolve_lds(uint32_t maxcap) {
    // bytes + 16, one u16 pointer per byte, lag list, control words
    return ((maxcap + 15u) & ~15u) * 3u + 16u + kSplitMaxLag * 8u + 64u;
}

// record = kind | out_len << 2 | arg << 10;  arg = literal byte, or the window offset of a copy
enum : uint32_t { kRecLit = 0, kRecCopy = 1, kRecFill = 2, kRecCopyExt = 3 };
// meta = ntok | wbits-8 << 20 | dict_sel << 23 | nlag << 25 | fallback << 31
constexpr uint32_t kMetaFallback = 1u << 31;

struct SplitArgs {
    DecompressArgs d;        // the batch (d.only_flagged is not used by these kernels)
    uint32_t* recs;          // n_streams x tokcap records
    uint32_t* meta;          // n_streams
    uint32_t* lag;           // n_streams x kSplitMaxLag x 2: (Oend | Vend << 16), cumulative lag
    uint8_t* flagged;        // n_streams: 1 = left to the lane / wave decoders
    uint32_t* flagged_count; // how many of them are set (the leftover launch returns at once when none is)
    uint32_t tokcap;         // records per stream
    uint32_t maxcap;         // largest out_cap of the batch (sizes RESOLVE's LDS)
    uint32_t first;          // first stream of this slice
    uint32_t count;          // streams in this slice
    uint32_t spw;            // PARSE: streams per wavefront (16, 32 or 64).  The parse is a serial chain per stream and
                             // bound by latency, not issue, until every SIMD holds several waves: small batches spread
                             // their streams over more, partly filled waves
};

// ---------------------------------------------------------------------------------------------------------------
// PARSE: the reference's token loop without the data (compare tamp_decompress_kernel.hpp, exact loop)
// ---------------------------------------------------------------------------------------------------------------
// per lane in LDS: a 64-byte input ring and a stage of 20 records that leaves for HBM 16 records (64 bytes) at a time --
// sixty-four lanes each storing one dword to a slot of its own cost the memory pipeline a transaction per lane and token
// (the parse ran at a third of its instruction rate however many waves were resident); odd dword stride per lane
// (+ 4 bytes behind the ring that mirror its first dword: a bit window is two aligned dwords from anywhere in the ring)
constexpr uint32_t kParseRing = 64, kParseStage = 20, kParseLane = 64 + 4 + 4 * kParseStage + 8;  // 156 B: odd dword stride
__host__ __device__ constexpr uint32_t split_parse_lds(uint32_t threads) { return 128u + threads * kParseLane; }

__global__ void __launch_bounds__(256) tamp_decode_parse_kernel(SplitArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DecompressArgs& a = sa.d;
    uint8_t* const lut = smem;  // prefix-code LUT: index = the 7 bits after the leading 1 -> (extra bits << 4) | symbol
    for (uint32_t v = threadIdx.x; v < 128; v += blockDim.x) {
        const uint64_t codes_lo = 0x2b2624140b080300ull, codes_hi = 0x00ab27aa9594544bull, nbits = 0x979998877765532ull;
        uint32_t entry = 0;
        for (int sy = 1; sy < 15; sy++) {
            const uint32_t l = (uint32_t)((nbits >> (4 * sy)) & 15) - 1u;  // code length without the flag: 2..8
            const uint32_t code = (uint32_t)((sy < 8 ? codes_lo >> (8 * sy) : codes_hi >> (8 * (sy - 8))) & 0xFF);
            if ((code & ((1u << (l - 1)) - 1)) == (v >> (7 - (l - 1)))) entry = ((l - 1) << 4) | (uint32_t)sy;
        }
        lut[v] = (uint8_t)entry;
    }
    __syncthreads();
    const uint32_t k = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * sa.spw + (threadIdx.x & (kWave - 1));
    const bool live = (threadIdx.x & (kWave - 1)) < sa.spw && k < sa.count;
    const uint32_t s = sa.first + (live ? k : 0u);
    const uint8_t* const in = a.in + a.in_off[s];
    const uint32_t n = live ? a.in_len[s] : 0u;
    const uint32_t cap = a.out_cap[s];
    uint32_t* const rec = sa.recs + (size_t)(live ? k : 0u) * sa.tokcap;
    uint32_t* const lag = sa.lag + (size_t)(live ? k : 0u) * kSplitMaxLag * 2;
    uint8_t* const inr = smem + 128 + threadIdx.x * kParseLane;
    uint32_t* const rb = reinterpret_cast<uint32_t*>(inr + kParseRing + 4);  // staged records
    uint32_t nflushed = 0, nstage = 0;                                    // records in HBM / in the stage
    auto flush16 = [&]() {  // the stage's first 16 records -> HBM as four 16-byte stores; the rest moves to the front
        if (nflushed + 16 <= sa.tokcap) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) st16(reinterpret_cast<uint8_t*>(rec + nflushed + 4 * q), ld16(reinterpret_cast<const uint8_t*>(rb + 4 * q)));
        }
        nflushed += 16;
        nstage -= 16;
        st16(reinterpret_cast<uint8_t*>(rb), ld16(reinterpret_cast<const uint8_t*>(rb + 16)));  // (the up to four records behind)
    };
    uint32_t ip = 0, op = 0, nlag = 0, cumlag = 0;  // (records so far = nflushed + nstage; bytes written = op - cumlag)
    uint32_t wbits = 8, dict_sel = 2;
    bool fallback = false;
    int res = kInputExhausted;

    do {
        if (n > kMaxDecodeIn) { res = kBadArgument; break; }
        if (a.max_wbits < 8 || a.max_wbits > 15) { res = kInvalidConf; break; }  // decompressor.c:336
        if (n == 0) break;
        const uint32_t h0 = in[0];
        const uint32_t hs = 1 + (h0 & 1);
        if (n < hs) { ip = 1; break; }  // decompressor.c:405-410
        if (hs == 2 && in[1]) { res = kInvalidConf; break; }
        ip = hs;
        wbits = ((h0 >> 5) & 7) + 8;
        const uint32_t lbits = ((h0 >> 3) & 3) + 5;
        const bool custom = (h0 >> 2) & 1, extended = (h0 >> 1) & 1, dreset = h0 & 1;
        if (wbits > a.max_wbits) { res = kInvalidConf; break; }  // decompressor.c:311
        const uint32_t W = 1u << wbits, mask = W - 1;
        const uint32_t minp = (uint32_t)min_pattern_size((int)wbits, (int)lbits);
        dict_sel = (!extended || lbits >= 7) ? 2u : (lbits == 6 ? 1u : 0u);  // decompressor.c:318-319
        if (custom) {
            if (!a.dict || a.dict_len < W) { res = kInvalidConf; break; }
            dict_sel = 3;
        }
        if (cap > sa.maxcap) fallback = true;  // (cannot happen: maxcap is the batch maximum)

        // one record; `written` = bytes of it that enter the window
        // (FAST: from the fast loop, where the stage is emptied every four tokens and cannot fill up, and every token has bytes)
        auto put_any = [&](uint32_t kind, uint32_t olen, uint32_t arg, uint32_t written, auto fast_loop) {
            if (!fast_loop.value && olen == 0) return;
            fallback = fallback | (nflushed + nstage >= sa.tokcap);
            rb[nstage++] = kind | (olen << 2) | (arg << 10);
            if (!fast_loop.value && nstage == kParseStage) flush16();
            op += olen;
            if (written < olen) {  // a lag: later window offsets name bytes further back in the output
                cumlag += olen - written;
                if (nlag < kSplitMaxLag) {
                    lag[2 * nlag] = (op & 0xFFFFu) | (((op - cumlag) & 0xFFFFu) << 16);
                    lag[2 * nlag + 1] = cumlag;
                } else {
                    fallback = true;
                }
                nlag++;
            }
        };
        auto put = [&](uint32_t kind, uint32_t olen, uint32_t arg, uint32_t written) {
            put_any(kind, olen, arg, written, std::integral_constant<bool, false>{});
        };

        uint32_t bb = 0, nb = 0, stage = 0, ns = 0;
        bool last_flush = false;
        auto refill = [&]() {  // decompressor.c:357-365; bytes reach `stage` a dword at a time
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

        // Two loops alternate.  FAST: straight-line decode from a 32-bit bit window read afresh per token (the token decode of the lane
        // decoders' bulk path, tamp_decompress_kernel.hpp, without its data movement), input through a 64-byte ring per lane
        // that 16-byte global loads refill at points common to the wave.  It declines -- at a token boundary, nothing
        // consumed -- whatever it does not cover: the last ~32 input bytes, FLUSH, a token the output has no room for, an
        // out-of-bounds offset, a dry ring.  EXACT: the reference's own loop, rebuilt from the bit position (including the
        // refill cursor that decides the consumed count), takes two tokens and hands back, or finishes the stream.
        // Short messages (round 4: a compressed 256-byte telemetry message is ~35 bytes): the WHOLE stream is put into the
        // ring once -- dwords while they lie inside the stream, then bytes: nothing is read behind its end -- and the fast
        // loop runs on it without any further input; only the tokens inside the last four bytes go to the exact loop.
        // (Without this such batches ran the exact loop alone: 0.13 ms per 262,144 messages of parse.)
        const bool whole = n <= kParseRing;
        bool ring_loaded = false;
        const bool use_fast = whole ? n >= hs + 8 : n >= hs + 160;
        for (;;) {
        bool resume = false;
        uint32_t budget = 0xFFFFFFFFu;
        if (use_fast) {
            uint32_t T = 8 * ip - nb;  // bits consumed from the start of the stream
            const uint32_t sp = whole ? 0u : T >> 3;
            bool fast = whole ? (T >> 3) + 4 <= n : sp + 32 <= n;
            const uint32_t sp0 = sp;  // stream byte x lives at inr[(x - sp0) & 63]
            // 25+ bits of the stream from bit position t, MSb first: two aligned dwords of the ring, funnel-shifted
            auto window = [&](uint32_t t) -> uint32_t {
                const uint32_t o = ((t >> 3) - sp0) & (kParseRing - 1);
                const uint32_t* w = reinterpret_cast<const uint32_t*>(inr + (o & ~3u));
                return __builtin_bswap32(__builtin_amdgcn_alignbyte(w[1], w[0], o & 3u)) << (t & 7);
            };
            B16 cb = {{0, 0, 0, 0}};
            bool cb_valid = false;
            uint32_t fill = sp, ld_off = sp;  // stream bytes [.., fill) are in the ring; next chunk to load
            if (fast && whole) {
                if (!ring_loaded) {
                    uint32_t x = 0;
                    for (; x + 4 <= n; x += 4) {
                        uint32_t v;
                        __builtin_memcpy(&v, in + x, 4);
                        *reinterpret_cast<uint32_t*>(inr + x) = v;
                    }
                    uint32_t tail = 0;
                    for (uint32_t y = x; y < n; y++) tail |= (uint32_t)in[y] << (8 * (y - x));
                    *reinterpret_cast<uint32_t*>(inr + x) = tail;  // (x <= 64: the mirror dword at most)
                    ring_loaded = true;
                }
                fill = ld_off = n;  // (nothing more to load: the I/O points only empty the record stage)
                budget = 2;
            } else if (fast) {
                const B16 c0 = ld16(in + sp);
                st16(inr, c0);
                st16(inr + 16, ld16(in + sp + 16));
                *reinterpret_cast<uint32_t*>(inr + kParseRing) = c0.w[0];
                fill = ld_off = sp + 32;
                budget = 2;
            }
            const uint32_t T_in = T;
            uint32_t T_mark = T;  // bit position at the reference's most recent refill
            while (__ballot(fast)) {
                if (fast) {  // ---- I/O point: every fourth step, the same step for the whole wave ----
                    if (nstage >= 16) flush16();
                    if (cb_valid && fill + 16 - (T >> 3) <= kParseRing) {  // (nothing at or behind the read position is overwritten)
                        const uint32_t ro = (fill - sp0) & (kParseRing - 1);
                        st16(inr + ro, cb);
                        if (ro == 0) *reinterpret_cast<uint32_t*>(inr + kParseRing) = cb.w[0];
                        fill += 16;
                        cb_valid = false;
                    }
                    if (!cb_valid && ld_off + 16 <= n) {
                        cb = ld16(in + ld_off);
                        ld_off += 16;
                        cb_valid = true;
                    }
                }
#pragma unroll 1
                for (uint32_t quad = 0; quad < 4; quad++)
                if (fast) {  // (one exit, state committed in one place)
                const uint32_t T0 = T;
                uint32_t Tl = T0;  // bit position inside the token
                bool ok = (T0 >> 3) + 4 <= fill;  // a dry ring (the end of the input, mostly): the exact loop takes over
                uint32_t mark = T0;  // the reference refills at the top of every token (decompressor.c:357-365,431-445)
                const uint32_t wp = (op - cumlag) & mask, room = cap - op;  // (window_pos = bytes written mod W on a fresh decoder)
                // literal and plain match decoded side by side and selected (one branch instead of a tree of them: the
                // values a branch tree assigns on different paths meet in register copies); a stale window of a dry ring
                // decodes to garbage that `ok` discards
                const uint32_t win = window(T0);
                const bool is_lit = (win >> 31) != 0;  // decompressor.c:466-482
                const uint32_t e0 = lut[(win >> 23) & 0x7F];
                const bool coded = ((win >> 30) & 1) != 0;
                const uint32_t sym = coded ? (e0 & 15) : 0u;
                const uint32_t used_m = coded ? 2 + (e0 >> 4) : 2u;
                const uint32_t tok_m = sym + minp;  // plain match, decompressor.c:529-572 (9 + 15 bits at most)
                const uint32_t arg_m = (win << used_m) >> (32 - wbits);
                uint32_t used = is_lit ? 1 + lbits : used_m + wbits;
                uint32_t tok = is_lit ? 1u : tok_m;
                uint32_t wl = tok;
                uint32_t kind = is_lit ? (uint32_t)kRecLit : (uint32_t)kRecCopy;
                uint32_t arg = is_lit ? (win << 1) >> (32 - lbits) : arg_m;
                {   // (bitwise, not short-circuit: one select instead of a tree of exec-mask regions)
                    const bool ok_l = room >= 1, ok_m = (arg_m + tok_m <= W) & (tok_m <= room);
                    ok = ok & (is_lit ? ok_l : ok_m);
                }
                const bool special = !is_lit & ((sym == kSymFlush) | (extended & (sym >= kSymRle)));
                if (special) {
                    // RLE / extended match, decompressor.c:114-273 (FLUSH: left to the exact loop).  Straight-line as well:
                    // windows read from a dry ring are garbage that `ok` discards.
                    Tl += used_m;
                    const bool have2 = (sym != kSymFlush) & ((Tl >> 3) + 4 <= fill);
                    const uint32_t w2 = window(Tl);
                    const uint32_t trailing = sym == kSymRle ? 4u : 3u;
                    const bool coded2 = (w2 >> 31) != 0;
                    const uint32_t e2 = lut[(w2 >> 24) & 0x7F];
                    const uint32_t h = coded2 ? (e2 & 15) : 0u;
                    uint32_t u = coded2 ? 1 + (e2 >> 4) : 1u;
                    const uint32_t value = (h << trailing) + ((w2 << u) >> (32 - trailing));
                    u += trailing;
                    const bool is_rle = sym == kSymRle;
                    // the offset of an extended match gets a window of its own (13 + 15 bits would not fit one)
                    const bool have3 = ((Tl + u) >> 3) + 4 <= fill;
                    const uint32_t arg_x = window(Tl + u) >> (32 - wbits);
                    // ... and the reference refills once more in front of the offset if its buffer (25..32 bits after the
                    // top-of-token refill) no longer holds `wbits` bits (decompressor.c:447-456)
                    const uint32_t nb_top = 8 * (((T0 + 24) >> 3) + 1) - T0;
                    mark = (!is_rle & (nb_top - (Tl - T0) - u < wbits)) ? Tl + u : mark;
                    tok = is_rle ? value + 2 : value + minp + 12;
                    wl = is_rle ? min(min(tok, kRleWindowMax), W - wp) : min(tok, W - wp);
                    kind = is_rle ? (uint32_t)kRecFill : (uint32_t)kRecCopyExt;
                    arg = is_rle ? 0u : arg_x;
                    used = is_rle ? u : u + wbits;
                    ok = have2 & (tok <= room) & (is_rle | (have3 & (arg_x + tok <= W)));
                }
                if (ok) {
                    T = Tl + used;
                    T_mark = mark;
                    put_any(kind, tok, arg, wl, std::integral_constant<bool, true>{});
                } else {
                    fast = false;  // (nothing consumed: the exact loop reads the token again from T)
                }
                }
            }
            if (T != T_in) {  // the reference's buffer at this token boundary: everything its last refill pulled in
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
        }

        for (;;) {  // decompressor.c:431-575
            if (use_fast && budget-- == 0) {  // back to the fast loop (it declines by itself near the end of the input)
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
                put(kRecLit, 1, c, 1);
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
                if (dreset && last_flush) fallback = true;  // dictionary reset inside the stream: lane / wave decoders
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
                const uint32_t wp = (op - cumlag) & mask, room = cap - op;  // (window_pos = bytes written mod W on a fresh decoder)
                if (sym == kSymRle) {  // decompressor.c:140-173
                    const uint32_t count = value + 2;
                    const uint32_t w = count <= room ? count : room;
                    put(kRecFill, w, 0, min(w, min(min(count, kRleWindowMax), W - wp)));
                    if (w < count) { res = kOutputFull; break; }
                } else {  // decompressor.c:229-272
                    if (off >= W || off + match_len > W) { res = kOob; break; }
                    const uint32_t w = match_len <= room ? match_len : room;
                    put(kRecCopyExt, w, off, min(w, W - wp));  // up to the end of the buffer, no wrap
                    if (w < match_len) { res = kOutputFull; break; }
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
                put(kRecCopy, room, off, room);
                res = kOutputFull;
                break;
            }
            bb = b2 << wbits;
            nb = n2 - wbits;
            put(kRecCopy, match_len, off, match_len);
        }
        if (!resume) break;
        }  // fast / exact alternation
    } while (false);

    if (!live) return;
    for (uint32_t i = 0; i < nstage; i++)
        if (nflushed + i < sa.tokcap) rec[nflushed + i] = rb[i];
    const uint32_t ntok = nflushed + nstage;
    if (ntok > 0xFFFFFu || op > 0xFFFFu) fallback = true;
    a.out_len[s] = op;
    a.status[s] = (int8_t)res;
    if (a.in_consumed) a.in_consumed[s] = ip;
    sa.meta[k] = (ntok & 0xFFFFFu) | ((wbits - 8) << 20) | (dict_sel << 23) | ((nlag < 63 ? nlag : 63u) << 25) | (fallback ? kMetaFallback : 0u);
    sa.flagged[s] = fallback ? 1 : 0;
    if (fallback) atomicAdd(sa.flagged_count, 1u);
}

// ---------------------------------------------------------------------------------------------------------------
// RESOLVE: records -> one pointer per output byte -> pointer jumping -> bytes
// ---------------------------------------------------------------------------------------------------------------
// NT = threads per stream: 256 (one workgroup per stream) or 64 (one WAVEFRONT per stream, four streams per workgroup, for
// batches of short messages -- out_cap up to 1 KiB: a workgroup per 256-byte message is mostly barriers and idle lanes, and
// the split decoder used to lose to the lane decoders by a factor of four there).  With NT = 64 nothing crosses a
// wavefront: the barriers become wave barriers, the cross-wave sums fall away, each wavefront has its own slice of LDS.
// BPT = consecutive output bytes a thread expands in the byte pass.  Rounds 2-3: 16 (4,096 bytes per round and workgroup); round 4:
// FOUR -- 65,536 x 4 KiB 1.57 -> 1.47 ms, and a 256-byte message keeps all 64 lanes of its wavefront busy instead of 16
// (1 M messages: 2.11 -> 1.29 ms); 1 / 2 / 8 measured: 1.68 / 1.58 / 1.45 ms at 4 KiB, 8 loses a quarter on short messages.
template <uint32_t NT, uint32_t BPT = 4>
__global__ void __launch_bounds__(256) tamp_decode_resolve_kernel(SplitArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    const DecompressArgs& a = sa.d;
    static_assert(NT == 256 || NT == 64, "a workgroup or a wavefront per stream");
    static_assert(BPT == 16 || BPT == 8 || BPT == 4 || BPT == 2 || BPT == 1, "marks are read as one aligned group");
    // (the wavefront's number through v_readfirstlane: what follows from it -- the stream's record count, window, lags, sizes --
    // is then scalar to the compiler for the wavefront-per-stream build as well, instead of vector loads and exec-mask loops)
    const uint32_t k = NT == 256 ? blockIdx.x : blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (NT == 64 && k >= sa.count) return;  // (whole wavefronts: no workgroup barrier below)
    uint8_t* const smem = NT == 256 ? smem_all : smem_all + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * split_resolve_lds(sa.maxcap);
    auto sync = [&]() {
        if constexpr (NT == 256) {
            __syncthreads();
        } else {  // one wavefront: its LDS operations complete in order; keep the compiler from moving them across
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    auto sync_or = [&](int v) -> bool {
        if constexpr (NT == 256) return __syncthreads_or(v) != 0;
        sync();
        return __ballot(v != 0) != 0;
    };
    const uint32_t s = sa.first + k;
    const uint32_t meta = sa.meta[k];
    if (meta & kMetaFallback) return;  // decoded by the lane / wave decoders afterwards
    const uint32_t n_out = a.out_len[s];
    if (n_out == 0) return;
    const uint32_t ntok = meta & 0xFFFFFu, wbits = 8 + ((meta >> 20) & 7), dict_sel = (meta >> 23) & 3, nlag = (meta >> 25) & 63;
    const uint32_t W = 1u << wbits, mask = W - 1;
    const uint8_t* const dict = dict_sel == 3 ? a.dict : a.seed_dicts + ((size_t)dict_sel << 15);
    const uint32_t* const rec = sa.recs + (size_t)k * sa.tokcap;

    constexpr uint32_t nt = NT;  // (a constant keeps divisions by it shifts)
    const uint32_t tid = threadIdx.x & (NT - 1), lane = tid & (kWave - 1), wave = tid >> 6;
    const uint32_t capa = align_up(sa.maxcap, 16);
    uint8_t* const outb = smem;                                             // capa bytes (+16)
    uint16_t* const src = reinterpret_cast<uint16_t*>(smem + capa + 16);    // capa entries: src[p] == p <=> outb[p] is final
    uint32_t* const lagl = reinterpret_cast<uint32_t*>(smem + capa + 16 + 2 * capa);  // kSplitMaxLag x 2
    // 16 control words; an explicit LDS pointer (volatile accesses through a generic pointer compile to FLAT loads)
    typedef __attribute__((address_space(3))) volatile uint32_t LdsCtl;
    LdsCtl* const ctl = (LdsCtl*)(lagl + 2 * kSplitMaxLag);

    for (uint32_t i = tid; i < 2 * nlag; i += nt) lagl[i] = sa.lag[(size_t)k * kSplitMaxLag * 2 + i];
    sync();

    // lag before the token that starts at output position O (all lagging tokens that END at or before O)
    auto lag_before_out = [&](uint32_t O) -> uint32_t {
        uint32_t L = 0;
        for (uint32_t i = 0; i < nlag; i++) {
            if ((lagl[2 * i] & 0xFFFFu) <= O) L = lagl[2 * i + 1];
        }
        return L;
    };
    // output position of the byte with virtual position v (v >= 0): v + the lag of the lagging tokens written before it
    auto out_of_virtual = [&](uint32_t v) -> uint32_t {
        uint32_t L = 0;
        for (uint32_t i = 0; i < nlag; i++) {
            if ((lagl[2 * i] >> 16) <= v) L = lagl[2 * i + 1];
        }
        return v + L;
    };
    // ---- expansion, parallel over the bytes ----
    // Token pass: the records' sizes are prefix-summed (256 tokens per step) and every token leaves its number + 1 at the
    // output position where it starts (in `src`, which holds nothing else yet).  Byte pass: every thread takes 16
    // consecutive output bytes; the token of its first byte is the last mark at or before it (a max-scan over the
    // threads), the marks inside its own 16 entries switch tokens on the way.  Each byte becomes either a final byte
    // (literal, dictionary) or a pointer to an earlier output byte.
    for (uint32_t i = tid; i < capa / 2; i += nt) reinterpret_cast<uint32_t*>(src)[i] = 0;
    sync();
    {   // every thread takes a run of consecutive tokens: one prefix sum over the workgroup
        // (round 4: eight per thread as two 16-byte loads whenever the stream has at most 8 x NT tokens -- one round trip to L2 per
        // pass instead of one per token: 618 instructions used to take 0.17 ms here)
        const bool wide = ntok <= 8 * nt && sa.tokcap >= 8 * nt;  // (the loads stay inside this stream's slot)
        const uint32_t K = wide ? 8u : (ntok + nt - 1) / nt;
        const uint32_t j0 = min(tid * K, ntok), j1 = min(j0 + K, ntok);
        uint32_t sum = 0;
        if (wide) {
            // (threads whose eight records lie behind the stream's last token do not load them: round 4 read 8 x NT records
            // per stream whatever it held -- 8 KB for ~6 KB of records, a quarter of RESOLVE's fetch traffic)
            B16 ra = {{0, 0, 0, 0}}, rb2 = {{0, 0, 0, 0}};
            if (tid * 8 < ntok) ra = ld16(reinterpret_cast<const uint8_t*>(rec + tid * 8));
            if (tid * 8 + 4 < ntok) rb2 = ld16(reinterpret_cast<const uint8_t*>(rec + tid * 8 + 4));
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) {
                const uint32_t r = e < 4 ? ra.w[e & 3] : rb2.w[e & 3];
                sum += tid * 8 + e < ntok ? (r >> 2) & 0xFFu : 0u;
            }
        } else {
            for (uint32_t j = j0; j < j1; j++) sum += (rec[j] >> 2) & 0xFFu;
        }
        const uint32_t incl = wave_scan_add(sum);
        if (lane == kWave - 1) ctl[wave] = incl;
        sync();
        uint32_t O = incl - sum;
        for (uint32_t w2 = 0; w2 < wave; w2++) O += ctl[w2];
        if (wide) {
            B16 ra = {{0, 0, 0, 0}}, rb2 = {{0, 0, 0, 0}};  // (from L1 / L2)
            if (tid * 8 < ntok) ra = ld16(reinterpret_cast<const uint8_t*>(rec + tid * 8));
            if (tid * 8 + 4 < ntok) rb2 = ld16(reinterpret_cast<const uint8_t*>(rec + tid * 8 + 4));
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) {
                const uint32_t r = e < 4 ? ra.w[e & 3] : rb2.w[e & 3];
                const uint32_t olen = tid * 8 + e < ntok ? (r >> 2) & 0xFFu : 0u;
                if (olen) src[O] = (uint16_t)(tid * 8 + e + 1);
                O += olen;
            }
        } else {
            for (uint32_t j = j0; j < j1; j++) {  // (the records come from L1 the second time)
                const uint32_t olen = (rec[j] >> 2) & 0xFFu;
                if (olen) src[O] = (uint16_t)(j + 1);
                O += olen;
            }
        }
    }
    sync();
#if defined(TAMP_SPLIT_STOP) && TAMP_SPLIT_STOP == 1
    return;
#endif
    for (uint32_t r0 = 0, carry = 0; r0 < n_out; r0 += BPT * nt) {  // 4,096 bytes per round (NT = 256, BPT = 16)
        const uint32_t p0 = r0 + BPT * tid;
        uint32_t h[BPT >= 2 ? BPT / 2 : 1];  // this thread's marks
        if constexpr (BPT == 16) {
            uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
            if (p0 < capa) lo = *reinterpret_cast<const uint4*>(src + p0), hi = *reinterpret_cast<const uint4*>(src + p0 + 8);
            h[0] = lo.x, h[1] = lo.y, h[2] = lo.z, h[3] = lo.w, h[4] = hi.x, h[5] = hi.y, h[6] = hi.z, h[7] = hi.w;
        } else if constexpr (BPT == 8) {
            uint4 lo = make_uint4(0, 0, 0, 0);
            if (p0 < capa) lo = *reinterpret_cast<const uint4*>(src + p0);
            h[0] = lo.x, h[1] = lo.y, h[2] = lo.z, h[3] = lo.w;
        } else if constexpr (BPT == 4) {
            uint2 lo = make_uint2(0, 0);
            if (p0 < capa) lo = *reinterpret_cast<const uint2*>(src + p0);
            h[0] = lo.x, h[1] = lo.y;
        } else if constexpr (BPT == 2) {
            h[0] = p0 < capa ? *reinterpret_cast<const uint32_t*>(src + p0) : 0u;
        } else {
            h[0] = p0 < capa ? (uint32_t)src[p0] : 0u;
        }
        uint32_t last = 0;  // position + 1 of this thread's last mark
#pragma unroll
        for (uint32_t i = 0; i < BPT; i++)
            if ((h[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) last = p0 + i + 1;
        const uint32_t inc = wave_scan_max(last);  // inclusive max-scan over the wave
        if (lane == kWave - 1) ctl[8 + wave] = inc;
        sync();
        // last mark in front of this thread's bytes (position + 1): the scan shifted by one lane (DPP wave_shr:1, 0 into lane 0)
        uint32_t head = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xF, 0xF, false);
        for (uint32_t w2 = 0; w2 < wave; w2++) head = max(head, (uint32_t)ctl[8 + w2]);
        const uint32_t carry_in = carry;
        head = max(head, carry);
        uint32_t wgmax = carry;
        for (uint32_t w2 = 0; w2 < (nt >> 6); w2++) wgmax = max(wgmax, (uint32_t)ctl[8 + w2]);
        carry = wgmax;
        // token of the byte in front of p0 (it spans into p0 unless p0 is marked).  Marks of earlier rounds have been
        // overwritten with pointers: the last token of the previous round was handed over in ctl[12].
        uint32_t jcur = 0;
        if (head) jcur = (head == carry_in && r0) ? (uint32_t)ctl[12] : (uint32_t)src[head - 1] - 1u;
        uint32_t hpos = head ? head - 1 : 0u;
        sync();  // every mark has been read: `src` may be overwritten with pointers now
        if (p0 < n_out) {
            // (a rolled loop over the thread's own LDS entries: unrolled over register copies it was 1,700 VALU
            // instructions of straight-line code for the 16 bytes)
            uint32_t kind = 0, arg = 0, Vj = 0;
            const uint32_t pend = min(p0 + BPT, n_out);
            if (head) {  // the token that reaches into this thread's bytes (a mark at p0 replaces it at once)
                const uint32_t r = rec[jcur];
                kind = r & 3u, arg = r >> 10;
                Vj = nlag ? hpos - lag_before_out(hpos) : hpos;
            }
#pragma unroll
            for (uint32_t i = 0; i < BPT; i++) {
                const uint32_t p = p0 + i;
                if (p >= pend) break;
                // (four bytes per thread: unrolled, with the marks this thread read before the barrier; sixteen: rolled over LDS)
                const uint32_t m = BPT <= 4 ? (h[i >> 1] >> (16 * (i & 1))) & 0xFFFFu : (uint32_t)src[p];
                if (m) {  // a token starts here (about five times per 16 bytes: the records sit in L1 / L2)
                    jcur = m - 1, hpos = p;
                    const uint32_t r = rec[jcur];
                    kind = r & 3u, arg = r >> 10;
                    Vj = nlag ? hpos - lag_before_out(hpos) : hpos;
                }
                // (selects instead of a branch per token kind: the kinds alternate from byte to byte and lane to lane)
                const bool lit = kind == kRecLit;
                const uint32_t idx = kind == kRecFill ? ((Vj - 1) & mask) : arg + (p - hpos);  // ring index read
                const uint32_t back = (Vj - 1 - idx) & mask;  // 0 = newest ... W-1 = oldest
                const bool from_dict = !lit && back >= Vj;    // never written: the dictionary
                uint32_t v = Vj - 1 - back;
                if (nlag && !lit && !from_dict) v = out_of_virtual(v);
                const uint32_t sp = (lit || from_dict) ? p : v;
                uint32_t byte = lit ? arg : 0u;
                if (from_dict) byte = dict[idx & mask];  // (& mask: groups of one long stream arrive with rotated offsets)
                src[p] = (uint16_t)sp;
                outb[p] = (uint8_t)byte;
            }
            // (the thread that owns the last byte of a full round also sees the token that reaches its end)
            if (tid == nt - 1) ctl[12] = jcur;  // the token that reaches the end of this round
        }
        sync();
    }

#if defined(TAMP_SPLIT_STOP) && TAMP_SPLIT_STOP == 2
    return;
#endif
    // ---- pointer jumping: every round halves the chains; a byte is final when it points at itself.  Each thread keeps
    // working on its own unresolved bytes (one mask per 4,096 bytes), so a round costs what is still unresolved.  The bytes
    // of a thread are spread out (byte t, t + 256, t + 512, ... of each 4,096): sixteen neighbours are all literals or all
    // copies, and lanes in lock step wait for the lane with the most work -- spread out, the counts are nearly equal. ----
    static_assert(kSplitMaxOut <= 4 * 16 * 256, "four masks per thread");
    uint32_t um0 = 0, um1 = 0, um2 = 0, um3 = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        uint32_t um = 0;
        if (q * 16 * nt < n_out) {
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) {
                const uint32_t p = q * 16 * nt + i * nt + tid;
                um |= ((p < n_out && src[p] != p) ? 1u : 0u) << i;
            }
        }
        if (q == 0) um0 = um;
        else if (q == 1) um1 = um;
        else if (q == 2) um2 = um;
        else um3 = um;
    }
    for (uint32_t round = 0; round < 17; round++) {
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            uint32_t um = q == 0 ? um0 : (q == 1 ? um1 : (q == 2 ? um2 : um3));
            const uint32_t p0 = q * 16 * nt + tid;
            for (uint32_t m = um; m;) {
                const uint32_t i = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                const uint32_t p = p0 + i * nt;
                // three hops per pass (round 4; rounds 2-3: two): a chain shrinks to a third instead of a half, 37 % fewer
                // passes for one more LDS read each (four hops: the same 0.64 ms).  s2 is final when src[s2] == s2 -- which also holds when s1 is (s2 = s1).
                const uint32_t s1 = src[p];
                const uint32_t s2 = src[s1];
                const uint32_t s3 = src[s2];
                // (program order matters twice: the byte is read after its "final" mark was seen, and written before
                // this position's own mark -- the LDS serves every wave's operations in order)
                asm volatile("" ::: "memory");
                if (s3 == s2) {
                    outb[p] = outb[s2];
                    asm volatile("" ::: "memory");
                    src[p] = (uint16_t)p;
                    um &= ~(1u << i);
                } else {
                    src[p] = (uint16_t)s3;
                }
            }
            if (q == 0) um0 = um;
            else if (q == 1) um1 = um;
            else if (q == 2) um2 = um;
            else um3 = um;
        }
        if (!sync_or((int)(um0 | um1 | um2 | um3))) break;
    }

#if defined(TAMP_SPLIT_STOP) && TAMP_SPLIT_STOP == 3
    return;
#endif
    // ---- out: aligned dwords, byte head / tail ----
    uint8_t* const out = a.out + a.out_off[s];
    const uint32_t head = min((uint32_t)((4 - (reinterpret_cast<uintptr_t>(out) & 3)) & 3), n_out);
    if (tid < head) out[tid] = outb[tid];
    const uint32_t ndw = (n_out - head) >> 2;
    uint32_t* const out32 = reinterpret_cast<uint32_t*>(out + head);
    for (uint32_t i = tid; i < ndw; i += nt) out32[i] = lds_u32_unaligned(outb, head + 4 * i);
    for (uint32_t i = head + 4 * ndw + tid; i < n_out; i += nt) out[i] = outb[i];
}

}  // namespace tamp_amd
