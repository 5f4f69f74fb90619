#!/usr/bin/env python3
"""Host-side search for the 16-bit multiplier of the bigram mix (tamp_compress_kernel.hpp: mix16).  The index keys its
buckets by the top HB bits of (bigram * M) mod 2^16 (2,048 until round 6, 1,024 for the W = 2^10 build since) -- any odd M is a bijection, so exactness does not depend on it,
only how many foreign bigrams share a query's bucket.  For every candidate M this replays the kernel's brackets (tile-ordered
buckets, 256-position tiles) on epoch buffers cut from the synthetic text and the frozen corpora and reports
entries scanned per query and lock-step iterations per 64 queries after the sort by bracket length.
usage: hash_search.py [coarse_epochs_per_corpus] [top_n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamp_amd  # noqa: E402
from tamp_amd import workloads as wl  # noqa: E402

W, TILE = 1024, 256
BLK, HB = int(os.environ.get('BLK', '1024')), int(os.environ.get('HB', '10'))  # (round 6: 1,024 positions, 1,024 buckets)
CUR = int(os.environ.get('CUR', '46437'))
DICT = np.frombuffer(bytes(tamp_amd.initialize_dictionary(W)), dtype=np.uint8)


def epochs(rows):
    out = []
    for r in rows:
        hist = np.concatenate([DICT, r]).astype(np.uint32)
        for e0 in range(0, len(r), BLK):
            nv = min(BLK, len(r) - e0)
            buf = hist[e0:e0 + W + nv + 1]
            if len(buf) < W + nv + 1:
                buf = np.concatenate([buf, np.zeros(W + nv + 1 - len(buf), dtype=np.uint32)])
            big = buf[:W + nv] | (buf[1:W + nv + 1] << 8)
            out.append((big, nv))
    return out


def cost(eps, M, hb=HB):
    scanned = lock = nq = 0
    for big, nv in eps:
        NE = W + nv
        h = ((big * M) & 0xFFFF) >> (16 - hb)
        tile = np.arange(NE) // TILE
        nt = (NE + TILE - 1) // TILE
        C = np.bincount(h * (nt + 1) + tile + 1, minlength=(1 << hb) * (nt + 1)).reshape(1 << hb, nt + 1)
        P = np.cumsum(C, axis=1)  # P[b, t] = entries of bucket b in tiles < t
        q = np.arange(nv)
        bq = h[W + q]
        L = P[bq, (W + q) // TILE + 1] - P[bq, q // TILE]
        scanned += int(L.sum())
        s = np.sort(np.minimum(L, 10 ** 6))[::-1]
        pad = (-len(s)) % 64
        s = np.concatenate([s, np.zeros(pad, dtype=s.dtype)]).reshape(-1, 64)
        lock += int(s.max(axis=1).sum())
        nq += nv
    return scanned / nq, lock * 64 / nq


def main():
    ncoarse = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    topn = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    corp = {
        "synth": wl.synth_text(24, 4096),
        "prose": wl.tile_rows(wl.real_text("prose"), 768)[::19][:24],
        "python": wl.tile_rows(wl.real_text("python"), 768)[::19][:24],
        "markup": wl.tile_rows(wl.real_text("markup"), 768)[::19][:24],
    }
    eps = {k: epochs(v) for k, v in corp.items()}
    base = {k: cost(v, CUR) for k, v in eps.items()}
    print(f"current M={CUR} (HB={HB}, BLK={BLK}):", {k: (round(a, 2), round(b, 2)) for k, (a, b) in base.items()})
    # coarse pass: every odd multiplier on a few epochs per corpus
    small = {k: v[:ncoarse] for k, v in eps.items()}
    b0 = {k: cost(v, CUR)[1] for k, v in small.items()}
    res = []
    for M in range(1, 65536, 2):
        sc = 0.0
        for k, v in small.items():
            sc += cost(v, M)[1] / b0[k]
        res.append((sc / len(small), M))
    res.sort()
    print("coarse best:", [(round(s, 4), m) for s, m in res[:topn]])
    fine = []
    for s, M in res[:topn * 4]:
        r = {k: cost(v, M) for k, v in eps.items()}
        fine.append((sum(r[k][1] / base[k][1] for k in r) / len(r), M, r))
    fine.sort(key=lambda t: t[0])
    for sc, M, r in fine[:topn]:
        print(f"M={M} (0x{M:04x}) rel {sc:.4f}:", {k: (round(a, 2), round(b, 2)) for k, (a, b) in r.items()})


if __name__ == "__main__":
    main()
