/*
 * tamp_oracle.c -- CPU restatement of the tamp hot path.  TEST INFRASTRUCTURE ONLY
 * (see tamp_oracle.h for who may call this and how it is pinned).
 *
 * The reference (tamp/_c_src/tamp/) is a streaming state machine around a 16-byte
 * input ring and a 32-bit bit accumulator.  This file restates the same arithmetic
 * for the one-shot case the batch codec serves: the stream is an array, the
 * "ring" at input position p is in[p .. p+R) with R = min(16, n-p), and tokens are
 * appended to a flat bit string.  Every function names the reference lines it follows.
 */
#include "tamp_oracle.h"

#include <string.h>

/* ------------------------------------------------------------------------- */
/* Static tables                                                             */
/* ------------------------------------------------------------------------- */

/* Match-length prefix code, flag bit excluded from `code`, included in `nbits`
 * (compressor.c:33-36).  Index 12 = RLE, 13 = extended match, 14 = FLUSH. */
static const uint8_t k_code[15] = {0x00, 0x03, 0x08, 0x0b, 0x14, 0x24, 0x26, 0x2b, 0x4b, 0x54, 0x94, 0x95, 0xaa, 0x27, 0xab};
static const uint8_t k_nbits[15] = {2, 3, 5, 5, 6, 7, 7, 7, 8, 8, 9, 9, 9, 7, 9};

enum { SYM_RLE = 12, SYM_EXT = 13, SYM_FLUSH = 14 };
enum { RING = 16, RLE_MAX = 241, RLE_WINDOW_MAX = 8, EXT_EXTRA_MAX = 120 };

static oracle_token_cb g_cb;
static void *g_cb_user;
void oracle_set_token_cb(oracle_token_cb cb, void *user) {
    g_cb = cb;
    g_cb_user = user;
}
#define TRACE(kind, pos, len, idx)                                        \
    do {                                                                  \
        if (g_cb) g_cb(g_cb_user, (kind), (pos), (unsigned)(len), (unsigned)(idx)); \
    } while (0)

/* ------------------------------------------------------------------------- */
/* a1/a2: dictionary seed and minimum pattern size                            */
/* ------------------------------------------------------------------------- */

/* common.c:28-52 -- xorshift32 from 3758097560, one draw per 8 bytes, nibble k picks
 * from a 16-entry table chosen by `literal` (common.c:18-25). */
void oracle_initialize_dictionary(uint8_t *buf, size_t size, uint8_t literal) {
    static const char text16[] = " etaoinshrdlcumw";
    static const uint8_t markup16[16] = {' ', 0, '0', 'e', 'i', '>', 't', 'o', '<', 'a', 'n', 's', '\n', 'r', '/', '.'};
    uint8_t table[16];
    for (int k = 0; k < 16; k++) {
        if (literal <= 5)
            table[k] = (uint8_t)text16[k] & 0x1F;
        else if (literal == 6)
            table[k] = (uint8_t)text16[k] & 0x3F;
        else
            table[k] = markup16[k];
    }
    uint32_t s = 3758097560u, draw = 0;
    for (size_t i = 0; i < size; i++) {
        if ((i & 7) == 0) {
            s ^= s << 13;
            s ^= s >> 17;
            s ^= s << 5;
            draw = s;
        }
        buf[i] = table[draw & 15];
        draw >>= 4;
    }
}

/* common.c:54-56 */
int oracle_min_pattern_size(uint8_t window, uint8_t literal) { return 2 + (window > 10 + 2 * (literal - 5)); }

/* ------------------------------------------------------------------------- */
/* Bit emitter                                                               */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint8_t *out;
    size_t cap;
    size_t nbytes;  /* whole bytes stored */
    uint64_t acc;   /* pending bits, right-aligned */
    unsigned nacc;  /* number of pending bits (< 8 after drain) */
    int overflow;
} BitSink;

/* write_to_bit_buffer + partial_flush (compressor.c:49-52, 65-75), collapsed:
 * append MSb-first and drain whole bytes at once. */
static void put_bits(BitSink *s, uint32_t v, unsigned n) {
    s->acc = (s->acc << n) | (uint64_t)v;
    s->nacc += n;
    while (s->nacc >= 8) {
        uint8_t b = (uint8_t)(s->acc >> (s->nacc - 8));
        if (s->nbytes < s->cap)
            s->out[s->nbytes] = b;
        else
            s->overflow = 1;
        s->nbytes++;
        s->nacc -= 8;
    }
    s->acc &= ((uint64_t)1 << s->nacc) - 1;
}

/* ------------------------------------------------------------------------- */
/* Compressor                                                                */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *in;
    size_t n;
    size_t p; /* next unconsumed input position */
    uint8_t *win;
    uint32_t W, mask;
    uint32_t wp; /* window write cursor */
    unsigned wbits, lbits, minp;
    int extended, lazy;
    unsigned rle_count;
    unsigned ext_count, ext_pos;
    int lazy_idx; /* -1 = none */
    unsigned lazy_len;
    BitSink bs;
} Enc;

static unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

/* find_best_match: exhaustive statement of compressor.c:113-172 /
 * compressor_find_match_desktop.c:82-167 / fuzz/esp32_host/differential.cpp:51-67.
 * Candidates are window indices 0..W-2 whose first two bytes equal the pattern's; a
 * match may not run past index W-1; longest wins, ties keep the lowest index. */
static void best_match(const Enc *e, size_t p, unsigned R, unsigned *idx_out, unsigned *len_out) {
    *len_out = 0;
    *idx_out = 0;
    if (R < e->minp) return;
    const unsigned maxp = e->extended ? e->minp + 11 + EXT_EXTRA_MAX : e->minp + 13; /* compressor.c:12-19 */
    const unsigned cap = umin(R, maxp);
    const uint8_t *pat = e->in + p;
    for (uint32_t i = 0; i + 1 < e->W; i++) {
        if (e->win[i] != pat[0] || e->win[i + 1] != pat[1]) continue;
        unsigned len = 2;
        while (len < cap && i + len < e->W && e->win[i + len] == pat[len]) len++;
        if (len > *len_out) {
            *len_out = len;
            *idx_out = i;
            if (len == cap) return;
        }
    }
}

/* find_extended_match (compressor.c:297-333): look for window[pos..pos+cnt) followed by
 * the next ring bytes, scanning candidates upward from `pos`; first-longest wins. */
static void ext_match_search(const Enc *e, size_t p, unsigned R, unsigned pos, unsigned cnt, unsigned *npos,
                             unsigned *ncnt) {
    *ncnt = 0;
    *npos = pos;
    const unsigned maxp = umin(cnt + R, e->minp + 11 + EXT_EXTRA_MAX);
    const uint8_t nextb = e->in[p];
    for (uint32_t c = pos; c + cnt + 1 <= e->W; c++) {
        if (e->win[c + cnt] != nextb) continue;
        if (memcmp(e->win + c, e->win + pos, cnt) != 0) continue;
        const unsigned cmax = umin(maxp, e->W - c);
        unsigned len = cnt + 1;
        while (len < cmax && e->win[c + len] == e->in[p + len - cnt]) len++;
        if (len > *ncnt) {
            *ncnt = len;
            *npos = c;
            if (len == maxp) return;
        }
    }
}

static uint8_t last_window_byte(const Enc *e) { return e->win[(e->wp - 1) & e->mask]; } /* compressor.c:270-273 */

/* write_extended_huffman (compressor.c:257-263) */
static void put_exthuff(Enc *e, unsigned value, unsigned trailing) {
    unsigned ci = value >> trailing;
    put_bits(&e->bs, ((uint32_t)k_code[ci] << trailing) | (value & ((1u << trailing) - 1)), (k_nbits[ci] - 1) + trailing);
}

/* write_rle_token (compressor.c:342-359) */
static void emit_rle(Enc *e, unsigned count, size_t run_start) {
    uint8_t sym = last_window_byte(e);
    put_bits(&e->bs, k_code[SYM_RLE], k_nbits[SYM_RLE]);
    put_exthuff(e, count - 2, 4);
    unsigned room = e->W - e->wp;
    unsigned w = umin(umin(count, RLE_WINDOW_MAX), room);
    for (unsigned i = 0; i < w; i++) {
        e->win[e->wp] = sym;
        e->wp = (e->wp + 1) & e->mask;
    }
    TRACE(2, run_start, count, 0);
}

/* tamp_window_copy (common.c:58-86): ring[wp..] <- ring[off..off+n), destination wraps,
 * memmove semantics (bytes are read before being overwritten). */
static void window_copy(uint8_t *win, uint32_t *wp, uint32_t off, unsigned n, uint32_t mask) {
    uint8_t tmp[256];
    memcpy(tmp, win + off, n);
    for (unsigned i = 0; i < n; i++) {
        win[*wp] = tmp[i];
        *wp = (*wp + 1) & mask;
    }
}

/* write_extended_match_token (compressor.c:377-415) */
static void emit_ext(Enc *e) {
    const unsigned count = e->ext_count, pos = e->ext_pos;
    put_bits(&e->bs, k_code[SYM_EXT], k_nbits[SYM_EXT]);
    put_exthuff(e, count - e->minp - 12, 3);
    put_bits(&e->bs, pos, e->wbits);
    unsigned room = e->W - e->wp;
    window_copy(e->win, &e->wp, pos, umin(count, room), e->mask);
    e->ext_count = 0;
    TRACE(3, e->p - count, count, pos);
}

static void emit_literal_raw(Enc *e, uint8_t c) {
    put_bits(&e->bs, (1u << e->lbits) | c, e->lbits + 1);
    e->win[e->wp] = c;
    e->wp = (e->wp + 1) & e->mask;
}

/* One parse step = tamp_compressor_poll (compressor.c:532-660) with the ring being
 * in[p .. p+R).  Returns ORACLE_OK or ORACLE_EXCESS_BITS. */
static int enc_step(Enc *e, unsigned R) {
    if (R == 0) return ORACLE_OK;
    unsigned idx = 0, len = 0;

    if (e->extended) { /* poll_extended_handling (compressor.c:437-525) */
        if (e->ext_count) {
            const unsigned max_ext = e->minp + 11 + EXT_EXTRA_MAX;
            while (R > 0) {
                if (e->ext_pos + e->ext_count >= e->W || e->ext_count >= max_ext) {
                    emit_ext(e);
                    e->lazy_idx = -1;
                    return ORACLE_OK;
                }
                unsigned npos, ncnt;
                ext_match_search(e, e->p, R, e->ext_pos, e->ext_count, &npos, &ncnt);
                if (ncnt > e->ext_count) {
                    unsigned extra = ncnt - e->ext_count;
                    e->ext_pos = npos;
                    e->ext_count = ncnt;
                    e->p += extra;
                    R -= extra;
                    continue;
                }
                emit_ext(e);
                e->lazy_idx = -1;
                return ORACLE_OK;
            }
            e->lazy_idx = -1;
            return ORACLE_OK;
        }

        const uint8_t last = last_window_byte(e);
        unsigned avail = 0;
        while (avail < R && e->rle_count + avail < RLE_MAX && e->in[e->p + avail] == last) avail++;
        const unsigned total = e->rle_count + avail;
        const int ended = (avail < R) || (total >= RLE_MAX);
        int handled = 0;
        if (!ended && total > 0) {
            e->rle_count = total;
            e->p += avail;
            handled = 1;
        } else if (total >= 2) {
            int use_pattern = 0;
            if (total == avail && total <= 6) { /* compressor.c:490-503 */
                best_match(e, e->p, R, &idx, &len);
                if (len > total) {
                    e->rle_count = 0;
                    use_pattern = 1;
                } else {
                    len = 0;
                }
            }
            if (!use_pattern) {
                size_t run_start = e->p - e->rle_count;
                e->p += avail;
                emit_rle(e, total, run_start);
                e->rle_count = 0;
                handled = 1;
            }
        } else if (e->rle_count == 1) { /* compressor.c:512-523 */
            emit_literal_raw(e, last);
            TRACE(0, e->p - 1, 1, last);
            e->rle_count = 0;
            handled = 1;
        }
        if (handled) {
            e->lazy_idx = -1;
            return ORACLE_OK;
        }
    }

    if (e->lazy) { /* compressor.c:576-619 */
        if (e->lazy_idx >= 0) {
            idx = (unsigned)e->lazy_idx;
            len = e->lazy_len;
            e->lazy_idx = -1;
        } else if (len == 0) {
            best_match(e, e->p, R, &idx, &len);
        }
        if (len >= e->minp && len <= 8 && R > len + 2) {
            unsigned nidx, nlen;
            best_match(e, e->p + 1, R - 1, &nidx, &nlen);
            /* validate_no_match_overlap (compressor.c:185-188) */
            if (nlen > len && (e->wp < nidx || e->wp >= nidx + nlen)) {
                e->lazy_idx = (int)nidx;
                e->lazy_len = nlen;
                len = 0;
            } else {
                e->lazy_idx = -1;
            }
        } else {
            e->lazy_idx = -1;
        }
    } else if (len == 0) {
        best_match(e, e->p, R, &idx, &len);
    }

    if (len < e->minp) { /* literal, compressor.c:625-632 */
        uint8_t c = e->in[e->p];
        if (c >> e->lbits) return ORACLE_EXCESS_BITS;
        put_bits(&e->bs, (1u << e->lbits) | c, e->lbits + 1);
        TRACE(0, e->p, 1, c);
        len = 1;
    } else {
        if (e->extended && len > e->minp + 11) { /* compressor.c:636-644 */
            e->ext_count = len;
            e->ext_pos = idx;
            e->p += len;
            return ORACLE_OK;
        }
        put_bits(&e->bs, ((uint32_t)k_code[len - e->minp] << e->wbits) | idx, k_nbits[len - e->minp] + e->wbits);
        TRACE(1, e->p, len, idx);
    }
    for (unsigned i = 0; i < len; i++) { /* compressor.c:651-657 */
        e->win[e->wp] = e->in[e->p + i];
        e->wp = (e->wp + 1) & e->mask;
    }
    e->p += len;
    return ORACLE_OK;
}

/* Core of every compress entry point: one SEGMENT = lead bits (header, append marker or nothing), the
 * reference's sink/poll loop over `in`, then tamp_compressor_flush(write_token).  `win` is the caller's window
 * (already seeded or carried over), *wp its write cursor. */
static int compress_core(const OracleConf *conf, uint8_t *win, uint32_t *wp, uint32_t lead, unsigned lead_bits,
                         const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, int flush_token,
                         int *token_written) {
    static uint8_t dummy;
    Enc e;
    memset(&e, 0, sizeof e);
    e.in = in ? in : &dummy;
    e.n = n;
    e.win = win;
    e.W = 1u << conf->window;
    e.mask = e.W - 1;
    e.wp = *wp & e.mask;
    e.wbits = conf->window;
    e.lbits = conf->literal;
    e.minp = (unsigned)oracle_min_pattern_size(conf->window, conf->literal);
    e.extended = conf->extended != 0;
    e.lazy = conf->lazy_matching != 0;
    e.lazy_idx = -1;
    e.bs.out = out;
    e.bs.cap = cap;
    if (token_written) *token_written = 0;
    if (lead_bits) put_bits(&e.bs, lead, lead_bits);

    int res = ORACLE_OK;
    /* tamp_compressor_compress_cb (compressor.c:681-722): refill the ring, step only while it is full. */
    size_t sunk = 0;
    while (sunk < n) {
        sunk = (e.p + RING < n) ? e.p + RING : n;
        if (sunk - e.p == RING) {
            size_t whole_before = e.bs.nbytes;
            res = enc_step(&e, RING);
            if (res != ORACLE_OK) {
                if (out_len) *out_len = whole_before < cap ? whole_before : cap;
                return res;
            }
        }
    }
    /* tamp_compressor_flush (compressor.c:728-810): drain with shrinking look-ahead. */
    for (;;) {
        if (e.p < n) {
            size_t whole_before = e.bs.nbytes;
            res = enc_step(&e, (unsigned)(n - e.p));
            if (res != ORACLE_OK) {
                if (out_len) *out_len = whole_before < cap ? whole_before : cap;
                return res;
            }
        } else if (e.extended && e.rle_count >= 1) {
            if (e.rle_count == 1) {
                uint8_t c = last_window_byte(&e);
                emit_literal_raw(&e, c);
                TRACE(0, e.p - 1, 1, c);
            } else {
                emit_rle(&e, e.rle_count, e.p - e.rle_count);
            }
            e.rle_count = 0;
        } else if (e.extended && e.ext_count) {
            emit_ext(&e);
        } else {
            break;
        }
    }
    /* FLUSH token only when bits are pending or the stream is a dictionary_reset one (compressor.c:784-794);
     * the caller has already folded last_was_flush into flush_token. */
    if (flush_token && (e.bs.nacc || conf->dictionary_reset)) {
        put_bits(&e.bs, k_code[SYM_FLUSH], k_nbits[SYM_FLUSH]);
        if (token_written) *token_written = 1;
    }
    if (e.bs.nacc) put_bits(&e.bs, 0, 8 - e.bs.nacc); /* zero-pad the last byte, compressor.c:799-807 */
    if (out_len) *out_len = e.bs.nbytes < cap ? e.bs.nbytes : cap;
    *wp = e.wp;
    return e.bs.overflow ? ORACLE_OUTPUT_FULL : ORACLE_OK;
}

static uint32_t header_bits(const OracleConf *conf, unsigned *nbits) { /* compressor.c:236-241 */
    uint32_t h = ((uint32_t)(conf->window - 8) << 5) | ((uint32_t)(conf->literal - 5) << 3) |
                 ((uint32_t)(conf->use_custom_dictionary != 0) << 2) | ((uint32_t)(conf->extended != 0) << 1) |
                 (uint32_t)(conf->dictionary_reset != 0);
    *nbits = conf->dictionary_reset ? 16 : 8;
    return conf->dictionary_reset ? h << 8 : h;
}

int oracle_compress(const OracleConf *conf, const uint8_t *dict, const uint8_t *in, size_t n, uint8_t *out,
                    size_t cap, size_t *out_len) {
    if (out_len) *out_len = 0;
    if (conf->window < 8 || conf->window > 15) return ORACLE_INVALID_CONF; /* compressor.c:208-209 */
    if (conf->literal < 5 || conf->literal > 8) return ORACLE_INVALID_CONF;
    if (conf->use_custom_dictionary && !dict) return ORACLE_INVALID_CONF;
    uint8_t win[1 << 15];
    const uint32_t W = 1u << conf->window;
    if (conf->use_custom_dictionary)
        memcpy(win, dict, W);
    else
        oracle_initialize_dictionary(win, W, conf->extended ? conf->literal : 8); /* compressor.c:224-225 */
    unsigned hb;
    uint32_t h = header_bits(conf, &hb), wp = 0;
    return compress_core(conf, win, &wp, h, hb, in, n, out, cap, out_len, 0, NULL);
}

int oracle_compress_segment(const OracleConf *conf, int emit_header, int append_marker, int resume, int flush_token,
                            uint8_t *window_state, uint16_t *window_pos, const uint8_t *in, size_t n, uint8_t *out,
                            size_t cap, size_t *out_len, int *token_written) {
    if (out_len) *out_len = 0;
    if (conf->window < 8 || conf->window > 15) return ORACLE_INVALID_CONF;
    if (conf->literal < 5 || conf->literal > 8) return ORACLE_INVALID_CONF;
    const uint32_t W = 1u << conf->window;
    uint32_t wp = *window_pos;
    if (!resume) {
        wp = 0;
        if (!conf->use_custom_dictionary)
            oracle_initialize_dictionary(window_state, W, conf->extended ? conf->literal : 8);
    }
    uint32_t lead = 0;
    unsigned lead_bits = 0;
    if (append_marker) /* compressor.c:227-235: FLUSH padded to 16 bits */
        lead = (uint32_t)k_code[SYM_FLUSH] << 7, lead_bits = 16;
    else if (emit_header)
        lead = header_bits(conf, &lead_bits);
    int res = compress_core(conf, window_state, &wp, lead, lead_bits, in, n, out, cap, out_len, flush_token,
                            token_written);
    *window_pos = (uint16_t)wp;
    return res;
}

/* ------------------------------------------------------------------------- */
/* Decompressor                                                              */
/* ------------------------------------------------------------------------- */

/* Prefix-code reader for the symbol after the 0 flag (decompressor.c:52-104).  The
 * reference uses a 128-entry LUT; here the code is walked from the (code, nbits)
 * table above.  Returns symbol 0..14, or -1 if `avail` bits are not enough. */
static int read_symbol(uint32_t bits_left_aligned, unsigned avail, unsigned *used) {
    if (avail < 1) return -1;
    if ((bits_left_aligned >> 31) == 0) {
        *used = 1;
        return 0;
    }
    /* The reference first consumes the leading 1, then indexes a LUT by the next 7 bits and
     * only afterwards checks that enough bits were present (decompressor.c:86-92).  Because no
     * code word is a prefix of another, matching against zero-filled bits gives the same symbol
     * and the same "enough bits?" outcome. */
    for (int s = 1; s < 15; s++) {
        unsigned nb = k_nbits[s] - 1u;
        if ((bits_left_aligned >> (32 - nb)) == k_code[s]) {
            if (avail < nb) return -1;
            *used = nb;
            return s;
        }
    }
    return -1; /* unreachable: the code is complete */
}

int oracle_decompress(const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_len, uint8_t max_window_bits,
                      uint8_t *out, size_t cap, size_t *out_len, size_t *in_consumed) {
    size_t ip = 0, op = 0;
    int res;
#define DONE(code)                      \
    do {                                \
        res = (code);                   \
        goto finish;                    \
    } while (0)

    if (max_window_bits < 8 || max_window_bits > 15) DONE(ORACLE_INVALID_CONF); /* decompressor.c:336 */

    /* read_header (decompressor.c:276-297) */
    if (n == 0) DONE(ORACLE_INPUT_EXHAUSTED);
    {
        size_t hs = 1 + (in[0] & 1);
        if (n < hs) {
            ip = 1; /* first byte is stashed (decompressor.c:405-410) */
            DONE(ORACLE_INPUT_EXHAUSTED);
        }
        if (hs == 2 && in[1]) DONE(ORACLE_INVALID_CONF);
        ip = hs;
    }
    const unsigned wbits = ((in[0] >> 5) & 7) + 8, lbits = ((in[0] >> 3) & 3) + 5;
    const int custom = (in[0] >> 2) & 1, extended = (in[0] >> 1) & 1, dreset = in[0] & 1;
    if (wbits > max_window_bits) DONE(ORACLE_INVALID_CONF); /* decompressor.c:311 */
    const uint32_t W = 1u << wbits, mask = W - 1;
    const unsigned minp = (unsigned)oracle_min_pattern_size((uint8_t)wbits, (uint8_t)lbits);

    uint8_t win[1 << 15];
    if (custom) {
        if (!dict || dict_len < W) DONE(ORACLE_INVALID_CONF);
        memcpy(win, dict, W);
    } else {
        oracle_initialize_dictionary(win, W, extended ? (uint8_t)lbits : 8); /* decompressor.c:318-319 */
    }

    uint32_t bb = 0;     /* left-aligned bit buffer */
    unsigned nb = 0;     /* bits in bb */
    uint32_t wp = 0;
    int last_flush = 0;

#define REFILL()                                   \
    while (ip < n && nb <= 24) {                   \
        nb += 8;                                   \
        bb |= (uint32_t)in[ip++] << (32 - nb);     \
    }
#define TAKE(k) (bb <<= (k), nb -= (k))

    /* main loop, decompressor.c:431-575 (one-shot: no resume state is ever live at the loop head) */
    for (;;) {
        if (!(ip < n || nb)) DONE(ORACLE_INPUT_EXHAUSTED);
        if (op == cap) DONE(ORACLE_OUTPUT_FULL);
        REFILL();
        if (nb == 0) DONE(ORACLE_INPUT_EXHAUSTED);

        if (bb >> 31) { /* literal, decompressor.c:466-482 */
            last_flush = 0;
            if (nb < 1 + lbits) DONE(ORACLE_INPUT_EXHAUSTED);
            TAKE(1);
            uint8_t c = (uint8_t)(bb >> (32 - lbits));
            TAKE(lbits);
            out[op++] = c;
            win[wp] = c;
            wp = (wp + 1) & mask;
            continue;
        }

        /* token: decode on copies so that a short read leaves the state untouched (decompressor.c:485-498) */
        uint32_t b2 = bb << 1;
        unsigned n2 = nb - 1, used;
        int sym = read_symbol(b2, n2, &used);
        if (sym < 0) DONE(ORACLE_INPUT_EXHAUSTED);
        b2 <<= used;
        n2 -= used;

        if (sym == SYM_FLUSH) { /* decompressor.c:501-514 */
            bb = b2 << (n2 & 7);
            nb = n2 & ~7u;
            if (dreset && last_flush) {
                wp = 0;
                oracle_initialize_dictionary(win, W, extended ? (uint8_t)lbits : 8);
            }
            last_flush = 1;
            continue;
        }
        last_flush = 0;

        if (extended && sym >= SYM_RLE) {
            /* Symbol bits are committed before the payload is read (decompressor.c:521-526). */
            bb = b2;
            nb = n2;
            unsigned trailing = (sym == SYM_RLE) ? 4u : 3u;
            unsigned value = 0, match_len = 0, off = 0;
            int got = 0; /* 0 nothing, 1 have value (ext only), 2 have everything */
            /* decode_rle / decode_extended_match (decompressor.c:114-273).  When bits run short the
             * reference returns INPUT_EXHAUSTED to the loop, which refills and retries; it gives up only
             * when the refill adds nothing and the input is spent (decompressor.c:447-456). */
            for (;;) {
                if (got == 0) {
                    uint32_t b3 = bb;
                    unsigned n3 = nb, u3;
                    int hs = (n3 >= 1 + trailing) ? read_symbol(b3, n3, &u3) : -1;
                    if (hs >= 0 && n3 - u3 < trailing) hs = -1;
                    if (hs >= 0) {
                        b3 <<= u3;
                        n3 -= u3;
                        value = ((unsigned)hs << trailing) + (b3 >> (32 - trailing));
                        b3 <<= trailing;
                        n3 -= trailing;
                        bb = b3;
                        nb = n3;
                        got = (sym == SYM_RLE) ? 2 : 1;
                        if (sym == SYM_EXT) match_len = value + minp + 12;
                    }
                }
                if (got == 1) {
                    if (nb >= wbits) {
                        off = bb >> (32 - wbits);
                        TAKE(wbits);
                        got = 2;
                    }
                }
                if (got == 2) break;
                unsigned before = nb;
                REFILL();
                if (nb == before && ip == n) DONE(ORACLE_INPUT_EXHAUSTED);
                if (op == cap) DONE(ORACLE_OUTPUT_FULL); /* loop head check after `continue` */
            }
            if (sym == SYM_RLE) { /* decode_rle body */
                unsigned count = value + 2;
                uint8_t c = win[(wp - 1) & mask];
                size_t room = cap - op;
                unsigned w = count <= room ? count : (unsigned)room;
                memset(out + op, c, w);
                op += w;
                unsigned ww = umin(umin(count, RLE_WINDOW_MAX), W - wp);
                for (unsigned i = 0; i < ww; i++) win[wp++] = c;
                wp &= mask;
                if (w < count) DONE(ORACLE_OUTPUT_FULL);
            } else { /* decode_extended_match body */
                if (off >= W || off + match_len > W) DONE(ORACLE_OOB); /* decompressor.c:232-236 */
                size_t room = cap - op;
                unsigned w = match_len <= room ? match_len : (unsigned)room;
                memcpy(out + op, win + off, w);
                op += w;
                if (w < match_len) DONE(ORACLE_OUTPUT_FULL); /* window only updated on a complete token */
                window_copy(win, &wp, off, umin(match_len, W - wp), mask);
            }
            continue;
        }

        /* plain match, decompressor.c:529-572 */
        if (n2 < wbits) DONE(ORACLE_INPUT_EXHAUSTED);
        unsigned match_len = (unsigned)sym + minp;
        uint32_t off = b2 >> (32 - wbits);
        if (off >= W || off + match_len > W) DONE(ORACLE_OOB); /* decompressor.c:540-544 */
        size_t room = cap - op;
        if (match_len > room) { /* partial copy, token not consumed (decompressor.c:553-557) */
            memcpy(out + op, win + off, room);
            op += room;
            DONE(ORACLE_OUTPUT_FULL);
        }
        bb = b2 << wbits;
        nb = n2 - wbits;
        memcpy(out + op, win + off, match_len);
        op += match_len;
        window_copy(win, &wp, off, match_len, mask);
    }

finish:
    if (out_len) *out_len = op;
    if (in_consumed) *in_consumed = ip;
    return res;
#undef DONE
#undef REFILL
#undef TAKE
}

/* ---------------------------------------------------------------------------------------------
 * Resumable decoder: the reference's TampDecompressor object as a value (decompressor.h:13-57) plus the
 * caller's window buffer, advanced by one tamp_decompressor_decompress call at a time with whatever input and
 * output room that call has (decompressor.c:371-578).  Test infrastructure, like the rest of this file.
 * --------------------------------------------------------------------------------------------- */
enum { DS_CONFIGURED = 1, DS_HEADER_STASHED = 2, DS_LAST_WAS_FLUSH = 4 };
enum { TS_NONE = 0, TS_RLE = 1, TS_EXT_FRESH = 2, TS_EXT_HAVE_SIZE = 3 }; /* decompressor.c:39-42 */

static int decoder_configure(OracleDecoder *d, uint8_t *window, unsigned wbits, unsigned lbits, int custom, int extended,
                             int dreset) {
    /* tamp_decompressor_populate_from_conf, decompressor.c:304-329 */
    if (wbits < 8 || wbits > 15 || lbits < 5 || lbits > 8 || wbits > d->window_bits_max) return ORACLE_INVALID_CONF;
    if (!custom) oracle_initialize_dictionary(window, (size_t)1 << wbits, extended ? (uint8_t)lbits : 8);
    d->conf = (uint8_t)(((wbits - 8) << 5) | ((lbits - 5) << 3) | (custom << 2) | (extended << 1) | dreset);
    d->flags |= DS_CONFIGURED;
    return ORACLE_OK;
}

int oracle_decoder_init(OracleDecoder *d, uint8_t *window, const OracleConf *conf, uint8_t window_bits_max) {
    if (window_bits_max < 8 || window_bits_max > 15) return ORACLE_INVALID_CONF; /* decompressor.c:336 */
    memset(d, 0, sizeof *d);
    d->window_bits_max = window_bits_max;
    if (!conf) return ORACLE_OK;
    return decoder_configure(d, window, conf->window, conf->literal, conf->use_custom_dictionary, conf->extended,
                             conf->dictionary_reset);
}

int oracle_decoder_call(OracleDecoder *d, uint8_t *win, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                        size_t *written, size_t *consumed) {
    size_t ip = 0, op = 0;
    int res = ORACLE_INPUT_EXHAUSTED;
#define DONE(code)     \
    do {               \
        res = (code);  \
        goto finish;   \
    } while (0)

    if (!(d->flags & DS_CONFIGURED)) { /* decompressor.c:389-429 */
        unsigned h0;
        if (d->flags & DS_HEADER_STASHED) {
            h0 = d->skip_bytes; /* the stash shares storage with skip_bytes */
            if (n == 0) DONE(ORACLE_INPUT_EXHAUSTED);
            if (in[0]) DONE(ORACLE_INVALID_CONF);
            ip = 1;
        } else {
            if (n == 0) DONE(ORACLE_INPUT_EXHAUSTED);
            h0 = in[0];
            if ((h0 & 1) && n < 2) {
                d->skip_bytes = (uint8_t)h0;
                d->flags |= DS_HEADER_STASHED;
                ip = 1;
                DONE(ORACLE_INPUT_EXHAUSTED);
            }
            if ((h0 & 1) && in[1]) DONE(ORACLE_INVALID_CONF);
            ip = 1 + (h0 & 1);
        }
        int rc = decoder_configure(d, win, ((h0 >> 5) & 7) + 8, ((h0 >> 3) & 3) + 5, (h0 >> 2) & 1, (h0 >> 1) & 1, h0 & 1);
        if (rc != ORACLE_OK) DONE(rc);
        d->skip_bytes = 0;
        d->flags &= (uint8_t)~DS_HEADER_STASHED;
    }
    {
        const unsigned wbits = ((d->conf >> 5) & 7) + 8, lbits = ((d->conf >> 3) & 3) + 5;
        const int extended = (d->conf >> 1) & 1, dreset = d->conf & 1;
        const uint32_t W = 1u << wbits, mask = W - 1;
        const unsigned minp = (unsigned)oracle_min_pattern_size((uint8_t)wbits, (uint8_t)lbits);
        uint32_t bb = d->bit_buffer;
        unsigned nb = d->bit_buffer_pos;
        uint32_t wp = d->window_pos;
#define REFILL()                               \
    while (ip < n && nb <= 24) {               \
        nb += 8;                               \
        bb |= (uint32_t)in[ip++] << (32 - nb); \
    }
#define SAVE() (d->bit_buffer = bb, d->bit_buffer_pos = (uint8_t)nb, d->window_pos = (uint16_t)wp)

        while (ip < n || nb || d->token_state) { /* decompressor.c:431 */
            if (op == cap) {
                SAVE();
                DONE(ORACLE_OUTPUT_FULL);
            }
            REFILL();

            if (d->token_state) { /* decompressor.c:441-462, with decode_rle / decode_extended_match inline */
            dispatch:;
                const int rle = d->token_state == TS_RLE;
                const unsigned trailing = rle ? 4u : 3u;
                unsigned skip = d->skip_bytes, count, off = 0;
                int starved = 0;
                if (skip) {
                    count = rle ? d->pending_window_offset : d->pending_match_size;
                    off = d->pending_window_offset;
                } else {
                    if (d->token_state == TS_EXT_HAVE_SIZE) {
                        count = d->pending_match_size;
                    } else {
                        unsigned used = 0;
                        int hs = (nb >= 1 + trailing) ? read_symbol(bb, nb, &used) : -1;
                        if (hs >= 0 && nb - used < trailing) hs = -1;
                        if (hs < 0) {
                            starved = 1;
                            count = 0;
                        } else {
                            bb <<= used, nb -= used;
                            count = ((unsigned)hs << trailing) + (bb >> (32 - trailing));
                            bb <<= trailing, nb -= trailing;
                            count += rle ? 2u : minp + 12u;
                        }
                    }
                    if (!starved && !rle) {
                        if (nb < wbits) { /* size known, offset not yet: decompressor.c:215-222 */
                            d->token_state = TS_EXT_HAVE_SIZE;
                            d->pending_match_size = (uint16_t)count;
                            starved = 1;
                        } else {
                            off = bb >> (32 - wbits);
                            bb <<= wbits, nb -= wbits;
                        }
                    }
                }
                if (starved) { /* decompressor.c:447-456 */
                    const unsigned before = nb;
                    REFILL();
                    if (nb == before && ip == n) {
                        SAVE();
                        DONE(ORACLE_INPUT_EXHAUSTED);
                    }
                    continue;
                }
                if (!rle && (off >= W || off + count > W)) {
                    SAVE();
                    DONE(ORACLE_OOB);
                }
                const unsigned remaining = count - skip;
                const size_t room = cap - op;
                unsigned w;
                if (remaining > room) { /* partial: remember where to pick up */
                    w = (unsigned)room;
                    d->skip_bytes = (uint8_t)(skip + w);
                    d->token_state = rle ? TS_RLE : TS_EXT_HAVE_SIZE;
                    d->pending_window_offset = (uint16_t)(rle ? count : off);
                    if (!rle) d->pending_match_size = (uint16_t)count;
                } else {
                    w = remaining;
                    d->skip_bytes = 0;
                    d->token_state = TS_NONE;
                }
                if (rle) {
                    memset(out + op, win[(wp - 1) & mask], w);
                    if (skip == 0) { /* window: first piece only, at most 8 bytes, no wrap */
                        const uint8_t c = win[(wp - 1) & mask];
                        unsigned ww = umin(umin(count, RLE_WINDOW_MAX), W - wp);
                        for (unsigned i = 0; i < ww; i++) win[wp++] = c;
                        wp &= mask;
                    }
                } else {
                    memcpy(out + op, win + off + skip, w);
                    if (d->token_state == TS_NONE) window_copy(win, &wp, off, umin(count, W - wp), mask);
                }
                op += w;
                if (d->token_state != TS_NONE) {
                    SAVE();
                    DONE(ORACLE_OUTPUT_FULL);
                }
                continue;
            }

            if (nb == 0) {
                SAVE();
                DONE(ORACLE_INPUT_EXHAUSTED);
            }
            if (bb >> 31) { /* literal */
                d->flags &= (uint8_t)~DS_LAST_WAS_FLUSH;
                if (nb < 1 + lbits) {
                    SAVE();
                    DONE(ORACLE_INPUT_EXHAUSTED);
                }
                const uint8_t c = (uint8_t)((bb << 1) >> (32 - lbits));
                bb <<= 1 + lbits, nb -= 1 + lbits;
                out[op++] = c;
                win[wp] = c;
                wp = (wp + 1) & mask;
                continue;
            }
            uint32_t b2 = bb << 1;
            unsigned n2 = nb - 1, used = 0;
            const int sym = read_symbol(b2, n2, &used);
            if (sym < 0) {
                SAVE();
                DONE(ORACLE_INPUT_EXHAUSTED);
            }
            b2 <<= used, n2 -= used;
            if (sym == SYM_FLUSH) {
                bb = b2 << (n2 & 7);
                nb = n2 & ~7u;
                if (dreset && (d->flags & DS_LAST_WAS_FLUSH)) {
                    wp = 0;
                    oracle_initialize_dictionary(win, W, extended ? (uint8_t)lbits : 8);
                }
                d->flags |= DS_LAST_WAS_FLUSH;
                continue;
            }
            d->flags &= (uint8_t)~DS_LAST_WAS_FLUSH;
            if (extended && sym >= SYM_RLE) { /* symbol committed, payload handled at the loop head */
                bb = b2, nb = n2;
                d->token_state = (uint8_t)(sym == SYM_RLE ? TS_RLE : TS_EXT_FRESH);
                /* straight into the dispatch (decompressor.c:521-527): no refill in between, so a token that ends
                 * the call (output full) leaves the input cursor where it was */
                goto dispatch;
            }
            if (n2 < wbits) {
                SAVE();
                DONE(ORACLE_INPUT_EXHAUSTED);
            }
            const unsigned match_len = (unsigned)sym + minp;
            const uint32_t off = b2 >> (32 - wbits);
            if (off >= W || off + match_len > W) {
                SAVE();
                DONE(ORACLE_OOB);
            }
            const unsigned skip = d->skip_bytes;
            unsigned w = match_len - skip;
            const size_t room = cap - op;
            if (w > room) { /* token stays in the bit buffer; the next call decodes it again and skips */
                w = (unsigned)room;
                d->skip_bytes = (uint8_t)(skip + w);
            } else {
                d->skip_bytes = 0;
                bb = b2 << wbits;
                nb = n2 - wbits;
            }
            memcpy(out + op, win + off + skip, w);
            op += w;
            if (d->skip_bytes == 0) window_copy(win, &wp, off, match_len, mask);
        }
        SAVE();
    }
finish:
    if (written) *written = op;
    if (consumed) *consumed = ip;
    return res;
#undef DONE
#undef REFILL
#undef SAVE
}
