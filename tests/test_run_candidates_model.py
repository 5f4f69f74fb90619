"""CPU tier: the selection rule of the run-aware build (tamp_compress_kernel.hpp second pass).

Interior positions of a long run are not in the bigram index; the kernel claims that, for "longest, then lowest
window index, never past index W-1" (compressor_find_match_desktop.c:43,59,65-69,118,159), the best interior candidate
of a run inside the window is always one of four positions: the first one, the one at window index 0, b - rq and
b - rq + 1 (rq = the pattern's own leading run).  This restates the kernel's key function in Python and checks the
claim against an exhaustive scan of every interior position, on random run-heavy buffers."""
import random


def lcp(buf, c, q, cap):
    n = 0
    while n < cap and buf[c + n] == buf[q + n]:
        n += 1
    return n


def key_of(buf, c, qpos, W, wp_e, cap_len):
    """The match phase's reduction key for candidate c against the pattern at buffer position qpos = W + q."""
    i = (c + wp_e) & (W - 1)          # window index of buffer position c
    lim_i = W - i                     # may not run past index W-1
    ln = lcp(buf, c, qpos, 16)
    return (min(ln, cap_len, lim_i) << 16) | lim_i


def test_four_candidates_per_run_suffice():
    rng = random.Random(5)
    checked = 0
    for it in range(300):
        W = rng.choice([64, 128, 256])
        blk = rng.choice([64, 128])
        n = W + blk + 48
        # run-heavy buffer over a tiny alphabet
        buf = []
        while len(buf) < n:
            x = rng.choice(b"ab ")
            buf += [x] * rng.choice([1, 1, 2, 3, 5, 8, 9, 13, 20, 40])
        buf = buf[:n] + [0] * 32
        wp_e = rng.randrange(W)
        # maximal runs of 8+ bytes [a, b), ends capped as in the kernel
        NE = W + blk
        runs = []
        c = 0
        while c < NE:
            e = c
            while e < NE + 16 and buf[e] == buf[c]:
                e += 1
            if e - c >= 8:
                runs.append((c, e))
            c = e
        for q in range(blk):
            qpos = W + q
            x = buf[qpos]
            if buf[qpos + 1] != x:
                continue                                  # only patterns starting x x meet interior positions
            rq = 0
            while rq < 16 and buf[qpos + rq] == x:
                rq += 1
            R = rng.choice([16, 16, 16, rng.randrange(2, 17)])   # look-ahead cap (stream tail)
            cap_len = R
            cz = q + ((-wp_e - q) & (W - 1))               # the window position with index 0
            for (a, b) in runs:
                if buf[a] != x:
                    continue
                lo, hi = max(a + 1, q), min(b - 4, q + W - 16)   # interior positions, linear part of the window
                if lo > hi:
                    continue
                best_all = max(key_of(buf, cc, qpos, W, wp_e, cap_len) for cc in range(lo, hi + 1))
                cs = b - rq
                sel = [cc for cc in (lo, cz, cs, cs + 1) if lo <= cc <= hi]
                best_sel = max(key_of(buf, cc, qpos, W, wp_e, cap_len) for cc in sel)
                assert best_sel == best_all, (it, W, q, a, b, rq, R, wp_e, lo, hi, sel)
                checked += 1
    assert checked > 5000
