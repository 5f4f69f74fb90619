/*
 * tamp_model.c -- scalar model of the DATA STRUCTURES the HIP compressor uses.  TEST ONLY.
 *
 * tamp_oracle.c restates the reference's ring window literally.  The HIP kernel does not keep a
 * ring: it keeps the *linear history* E' = dictionary ++ (every byte ever written to the window),
 * of which the live window is always the last W bytes, and it computes find_best_match for a
 * whole block of input positions at once under the speculation "every consumed byte has been
 * written" (true for literals and plain matches; RLE / extended-match tokens may write fewer
 * bytes than they consume -- compressor.c:352-358,404-410 -- which ends the speculation and
 * starts a new epoch).  This file is that algorithm in plain scalar C so that the design can be
 * checked against the oracle on the CPU before any kernel runs; tests/test_model_epoch.py fuzzes
 * model == oracle.  See DESIGN.md section 3 for the derivation.
 *
 * Buffer coordinates (one epoch):
 *   ebuf[0 .. W)            the window at epoch start, oldest byte first
 *   ebuf[W + k]             input[p0 + k]
 *   window index of ebuf[c] = (wp_e + c) mod W        (wp_e = window_pos at epoch start)
 *   wr = bytes written, rd = bytes consumed since epoch start; live window = ebuf[wr .. wr+W)
 *   speculative window for input position q (needs wr == rd == q):  ebuf[q .. q+W)
 */
#include <stdlib.h>

#include "tamp_oracle.c"

typedef struct {
    const uint8_t *in;
    size_t n;
    uint32_t W, mask;
    unsigned wbits, lbits, minp, maxp;
    int extended;
    uint32_t blk;
    uint8_t *ebuf;   /* W + blk + RING + PEND_MAX (+ slack) */
    uint8_t *blen;   /* blk */
    uint16_t *bidx;  /* blk */
    size_t p0;       /* input position of ebuf[W] */
    uint32_t wp_e;   /* window_pos at epoch start */
    uint32_t nvalid; /* positions with a precomputed best */
    uint32_t wr, rd;
    unsigned rle_count, ext_count, ext_pos;
    BitSink bs;
    unsigned n_epochs;
} Model;

static inline uint32_t m_wp(const Model *m) { return (m->wp_e + m->wr) & m->mask; }
/* byte at window index i of the LIVE window */
static inline uint8_t m_win(const Model *m, uint32_t i) { return m->ebuf[m->wr + ((i - m_wp(m)) & m->mask)]; }
static inline uint8_t m_in(const Model *m, uint32_t k) { return m->ebuf[m->W + m->rd + k]; } /* ring byte k */
static inline unsigned m_R(const Model *m) {
    size_t left = m->n - (m->p0 + m->rd);
    return left < RING ? (unsigned)left : RING;
}
static inline void m_append(Model *m, uint8_t b) {
    m->ebuf[m->W + m->wr] = b;
    m->wr++;
}

/* The parallel phase of the kernel: best match for every position of the block at once. */
static void m_match_phase(Model *m) {
    const uint32_t W = m->W;
    for (uint32_t q = 0; q < m->nvalid; q++) {
        size_t left = m->n - (m->p0 + q);
        unsigned R = left < RING ? (unsigned)left : RING;
        unsigned best = 0, besti = 0;
        if (R >= m->minp) {
            unsigned cap = umin(R, m->maxp);
            const uint8_t *pat = m->ebuf + W + q;
            uint32_t key = 0;
            for (uint32_t c = q; c < q + W; c++) {
                uint32_t i = (m->wp_e + c) & m->mask;
                if (i == W - 1) continue;
                unsigned lim = umin(cap, W - i), len = 0;
                while (len < lim) {
                    uint32_t s = c + len;
                    if (s >= q + W) s -= W; /* ran past the newest byte: the ring continues with the oldest */
                    if (m->ebuf[s] != pat[len]) break;
                    len++;
                }
                if (len < 2) continue;
                uint32_t k = ((uint32_t)len << 16) | (0xFFFFu - i);
                if (k > key) key = k;
            }
            best = key >> 16;
            besti = 0xFFFFu - (key & 0xFFFFu);
        }
        m->blen[q] = (uint8_t)best;
        m->bidx[q] = (uint16_t)besti;
    }
}

enum { PEND_MAX = 256 }; /* >= RLE_MAX and >= the longest extended match: bytes consumed but not yet written */

static void m_epoch_begin(Model *m) {
    /* Re-base: the live window becomes ebuf[0..W) and the input is reloaded behind it.  Bytes that a
     * pending RLE run / extended match has consumed but not yet written stay in front of the read
     * cursor (they become window bytes when the token is finally emitted); bytes that a finished
     * token consumed without writing (the lag) are dropped for good. */
    const uint32_t pending = m->rle_count + m->ext_count;
    memmove(m->ebuf, m->ebuf + m->wr, m->W);
    m->wp_e = m_wp(m);
    m->p0 += m->rd - pending;
    m->wr = 0;
    m->rd = pending;
    size_t left = m->n - m->p0;
    size_t room = (size_t)m->blk + RING + PEND_MAX;
    size_t nload = left < room ? left : room;
    memcpy(m->ebuf + m->W, m->in + m->p0, nload);
    memset(m->ebuf + m->W + nload, 0, room + 16 - nload);
    m->nvalid = left < m->blk ? (uint32_t)left : m->blk;
    m_match_phase(m);
    m->n_epochs++;
}

static void m_best(Model *m, unsigned *idx, unsigned *len) {
    if (m->wr != m->rd || m->rd >= m->nvalid) m_epoch_begin(m);
    *len = m->blen[m->rd];
    *idx = m->bidx[m->rd];
}

static void m_put_exthuff(Model *m, unsigned value, unsigned trailing) {
    unsigned ci = value >> trailing;
    put_bits(&m->bs, ((uint32_t)k_code[ci] << trailing) | (value & ((1u << trailing) - 1)), (k_nbits[ci] - 1) + trailing);
}

static void m_emit_rle(Model *m, unsigned count) {
    uint8_t sym = m_win(m, (m_wp(m) - 1) & m->mask);
    put_bits(&m->bs, k_code[SYM_RLE], k_nbits[SYM_RLE]);
    m_put_exthuff(m, count - 2, 4);
    unsigned w = umin(umin(count, RLE_WINDOW_MAX), m->W - m_wp(m));
    for (unsigned i = 0; i < w; i++) m_append(m, sym);
    TRACE(2, m->p0 + m->rd - count, count, 0);
}

static void m_emit_ext(Model *m) {
    const unsigned count = m->ext_count, pos = m->ext_pos;
    put_bits(&m->bs, k_code[SYM_EXT], k_nbits[SYM_EXT]);
    m_put_exthuff(m, count - m->minp - 12, 3);
    put_bits(&m->bs, pos, m->wbits);
    unsigned w = umin(count, m->W - m_wp(m));
    /* source indices are taken in the pre-token window; appends land beyond it, so no overlap */
    const uint32_t wr0 = m->wr, wp0 = m_wp(m);
    for (unsigned i = 0; i < w; i++) m_append(m, m->ebuf[wr0 + ((pos + i - wp0) & m->mask)]);
    m->ext_count = 0;
    TRACE(3, m->p0 + m->rd - count, count, pos);
}

static void m_ext_search(const Model *m, unsigned R, unsigned pos, unsigned cnt, unsigned *npos, unsigned *ncnt) {
    *ncnt = 0;
    *npos = pos;
    const unsigned maxp = umin(cnt + R, m->minp + 11 + EXT_EXTRA_MAX);
    const uint8_t nextb = m_in(m, 0);
    for (uint32_t c = pos; c + cnt + 1 <= m->W; c++) {
        if (m_win(m, c + cnt) != nextb) continue;
        unsigned i = 0;
        while (i < cnt && m_win(m, c + i) == m_win(m, pos + i)) i++;
        if (i < cnt) continue;
        const unsigned cmax = umin(maxp, m->W - c);
        unsigned len = cnt + 1;
        while (len < cmax && m_win(m, c + len) == m_in(m, len - cnt)) len++;
        if (len > *ncnt) {
            *ncnt = len;
            *npos = c;
            if (len == maxp) return;
        }
    }
}

/* One parse step; same decisions as enc_step() in tamp_oracle.c, different window representation. */
static int m_step(Model *m, unsigned R) {
    if (R == 0) return ORACLE_OK;
    unsigned idx = 0, len = 0;
    if (m->extended) {
        if (m->ext_count) {
            const unsigned max_ext = m->minp + 11 + EXT_EXTRA_MAX;
            while (R > 0) {
                if (m->ext_pos + m->ext_count >= m->W || m->ext_count >= max_ext) {
                    m_emit_ext(m);
                    return ORACLE_OK;
                }
                unsigned npos, ncnt;
                m_ext_search(m, R, m->ext_pos, m->ext_count, &npos, &ncnt);
                if (ncnt > m->ext_count) {
                    unsigned extra = ncnt - m->ext_count;
                    m->ext_pos = npos;
                    m->ext_count = ncnt;
                    m->rd += extra;
                    R -= extra;
                    continue;
                }
                m_emit_ext(m);
                return ORACLE_OK;
            }
            return ORACLE_OK;
        }
        const uint8_t last = m_win(m, (m_wp(m) - 1) & m->mask);
        unsigned avail = 0;
        while (avail < R && m->rle_count + avail < RLE_MAX && m_in(m, avail) == last) avail++;
        const unsigned total = m->rle_count + avail;
        const int ended = (avail < R) || (total >= RLE_MAX);
        if (!ended && total > 0) {
            m->rle_count = total;
            m->rd += avail;
            return ORACLE_OK;
        }
        if (total >= 2) {
            int use_pattern = 0;
            if (total == avail && total <= 6) {
                m_best(m, &idx, &len);
                if (len > total)
                    use_pattern = 1;
                else
                    len = 0;
            }
            if (!use_pattern) {
                m->rd += avail;
                m_emit_rle(m, total);
                m->rle_count = 0;
                return ORACLE_OK;
            }
        } else if (m->rle_count == 1) {
            put_bits(&m->bs, (1u << m->lbits) | last, m->lbits + 1);
            m_append(m, last);
            m->rle_count = 0;
            return ORACLE_OK;
        }
    }
    if (len == 0) m_best(m, &idx, &len);
    if (len < m->minp) {
        uint8_t c = m_in(m, 0);
        if (c >> m->lbits) return ORACLE_EXCESS_BITS;
        put_bits(&m->bs, (1u << m->lbits) | c, m->lbits + 1);
        TRACE(0, m->p0 + m->rd, 1, c);
        len = 1;
    } else {
        if (m->extended && len > m->minp + 11) {
            m->ext_count = len;
            m->ext_pos = idx;
            m->rd += len;
            return ORACLE_OK;
        }
        put_bits(&m->bs, ((uint32_t)k_code[len - m->minp] << m->wbits) | idx, k_nbits[len - m->minp] + m->wbits);
        TRACE(1, m->p0 + m->rd, len, idx);
    }
    for (unsigned i = 0; i < len; i++) m_append(m, m_in(m, i)); /* identity when wr == rd */
    m->rd += len;
    return ORACLE_OK;
}

/* Same contract as oracle_compress (lazy matching not modelled); `blk` = epoch block in positions. */
int model_compress(const OracleConf *conf, const uint8_t *dict, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                   size_t *out_len, uint32_t blk, unsigned *n_epochs) {
    static uint8_t dummy;
    if (out_len) *out_len = 0;
    if (conf->window < 8 || conf->window > 15 || conf->literal < 5 || conf->literal > 8) return ORACLE_INVALID_CONF;
    if (conf->use_custom_dictionary && !dict) return ORACLE_INVALID_CONF;
    Model m;
    memset(&m, 0, sizeof m);
    m.in = in ? in : &dummy;
    m.n = n;
    m.W = 1u << conf->window;
    m.mask = m.W - 1;
    m.wbits = conf->window;
    m.lbits = conf->literal;
    m.minp = (unsigned)oracle_min_pattern_size(conf->window, conf->literal);
    m.extended = conf->extended != 0;
    m.maxp = m.extended ? m.minp + 11 + EXT_EXTRA_MAX : m.minp + 13;
    m.blk = blk;
    m.ebuf = (uint8_t *)calloc(m.W + blk + RING + PEND_MAX + 64, 1);
    m.blen = (uint8_t *)calloc(blk, 1);
    m.bidx = (uint16_t *)calloc(blk, 2);
    m.bs.out = out;
    m.bs.cap = cap;
    if (conf->use_custom_dictionary)
        memcpy(m.ebuf, dict, m.W);
    else
        oracle_initialize_dictionary(m.ebuf, m.W, conf->extended ? conf->literal : 8);
    put_bits(&m.bs,
             ((uint32_t)(conf->window - 8) << 5) | ((uint32_t)(conf->literal - 5) << 3) |
                 ((uint32_t)(conf->use_custom_dictionary != 0) << 2) | ((uint32_t)(conf->extended != 0) << 1) |
                 (uint32_t)(conf->dictionary_reset != 0),
             8);
    if (conf->dictionary_reset) put_bits(&m.bs, 0, 8);
    m_epoch_begin(&m); /* wr = rd = 0 -> plain load + first match phase */

    int res = ORACLE_OK;
    size_t whole = 0;
    for (;;) {
        size_t p = m.p0 + m.rd;
        /* steps happen with a full ring while input remains, then with the shrinking tail */
        if (p < n) {
            /* a step may look at ring bytes beyond the loaded block: re-base first (never in the
             * middle of the speculation: only the RLE/ext paths run with wr != rd) */
            if (m.rd > m.blk + m.rle_count + m.ext_count) m_epoch_begin(&m);
            whole = m.bs.nbytes;
            res = m_step(&m, m_R(&m));
            if (res != ORACLE_OK) break;
        } else if (m.extended && m.rle_count >= 1) {
            if (m.rle_count == 1) {
                uint8_t c = m_win(&m, (m_wp(&m) - 1) & m.mask);
                put_bits(&m.bs, (1u << m.lbits) | c, m.lbits + 1);
                m_append(&m, c);
            } else {
                m_emit_rle(&m, m.rle_count);
            }
            m.rle_count = 0;
        } else if (m.extended && m.ext_count) {
            m_emit_ext(&m);
        } else {
            break;
        }
    }
    if (res == ORACLE_OK) {
        if (m.bs.nacc) put_bits(&m.bs, 0, 8 - m.bs.nacc);
        whole = m.bs.nbytes;
        if (m.bs.overflow) res = ORACLE_OUTPUT_FULL;
    }
    if (out_len) *out_len = whole < cap ? whole : cap;
    if (n_epochs) *n_epochs = m.n_epochs;
    free(m.ebuf);
    free(m.blen);
    free(m.bidx);
    return res;
}
