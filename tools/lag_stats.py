#!/usr/bin/env python3
"""CPU-side study of the tokens that write fewer bytes than they consume ("lags"): parses the
reference's own output token by token (format: SURVEY.md appendix A) and reports, per 4 KiB chunk of a corpus, how many
RLE tokens over 8 bytes / extended matches clipped at the ring end occur, and how many of the RLE lags start right
behind the first byte of their run (the arrival a predicted-gap scheme would have to guess).  Test-side tool: it uses
the oracle to produce the streams.  usage: lag_stats.py [prose|python|synth] [chunks]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.checker import Oracle, Ref  # noqa: E402
from tamp_amd import workloads as wl  # noqa: E402

CODES = [0x00, 0x03, 0x08, 0x0b, 0x14, 0x24, 0x26, 0x2b, 0x4b, 0x54, 0x94, 0x95, 0xaa, 0x27, 0xab]
NBITS = [2, 3, 5, 5, 6, 7, 7, 7, 8, 8, 9, 9, 9, 7, 9]
LUT = {(NBITS[i] - 1, CODES[i]): i for i in range(15)}


class Bits:
    def __init__(self, b):
        self.b, self.p, self.n = b, 0, len(b) * 8

    def get(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | ((self.b[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def huff(self):
        v = 0
        for k in range(1, 9):
            v = (v << 1) | self.get(1)
            if (k, v) in LUT:
                return LUT[(k, v)]
        raise ValueError("bad code")


def tokens(comp, n_out, W=1024, wbits=10, minp=2):
    """-> list of (kind, input position, consumed, written, window_pos before)"""
    bs = Bits(comp)
    bs.get(8)
    pos, wp, out = 0, 0, []
    while pos < n_out and bs.p + 9 <= bs.n + 8:
        if bs.get(1):
            bs.get(8)
            out.append(("lit", pos, 1, 1, wp))
            pos += 1
            wp = (wp + 1) & (W - 1)
            continue
        s = bs.huff()
        if s == 14:
            break
        if s == 12:
            cnt = ((bs.huff() << 4) | bs.get(4)) + 2
            w = min(cnt, 8, W - wp)
            out.append(("rle", pos, cnt, w, wp))
        elif s == 13:
            cnt = ((bs.huff() << 3) | bs.get(3)) + minp + 12
            bs.get(wbits)
            w = min(cnt, W - wp)
            out.append(("ext", pos, cnt, w, wp))
        else:
            cnt = s + minp
            bs.get(wbits)
            w = cnt
            out.append(("match", pos, cnt, w, wp))
        pos += cnt
        wp = (wp + w) & (W - 1)
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "prose"
    nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    if name == "synth":
        rows = wl.synth_text(nchunks, 4096)
    else:
        rows = wl.tile_rows(wl.real_text(name), nchunks)
    n = rows.shape[0]
    off, ln = wl.csr_for_fixed(n, 4096)
    try:
        orc = Ref()
    except Exception:
        orc = Oracle()
    res = orc.compress_batch(rows.reshape(-1), off, ln, window=10, literal=8, extended=True, threads=8)
    tot = dict(rle_lag=0, rle_head=0, rle_clip_only=0, ext_lag=0, tokens=0, rle=0, ext=0, lag_bytes=0)
    runlens = {}
    per_chunk = []
    for i in range(n):
        data = rows[i].tobytes()
        toks = tokens(res.stream(i), 4096)
        assert sum(t[2] for t in toks) == 4096, (i, sum(t[2] for t in toks))
        lags = 0
        for kind, pos, cnt, w, wp in toks:
            tot["tokens"] += 1
            if kind == "rle":
                tot["rle"] += 1
            if kind == "ext":
                tot["ext"] += 1
            if w < cnt:
                lags += 1
                tot["lag_bytes"] += cnt - w
                if kind == "rle":
                    tot["rle_lag"] += 1
                    x = data[pos]
                    # arrival right behind the head of the run: byte pos-1 is x (it is: the RLE rule), pos-2 is not
                    head = pos >= 1 and data[pos - 1] == x and (pos < 2 or data[pos - 2] != x)
                    if head:
                        tot["rle_head"] += 1
                    if cnt <= 8:
                        tot["rle_clip_only"] += 1
                    b = min(cnt, 40)
                    runlens[b] = runlens.get(b, 0) + 1
                else:
                    tot["ext_lag"] += 1
        per_chunk.append(lags)
    pc = np.array(per_chunk)
    print(f"{name}: {n} chunks, tokens/chunk {tot['tokens'] / n:.0f}, rle/chunk {tot['rle'] / n:.2f}, ext/chunk {tot['ext'] / n:.2f}")
    print(f"  lags/chunk {pc.mean():.2f} (max {pc.max()}, chunks without {np.mean(pc == 0):.2f}); RLE lags {tot['rle_lag'] / n:.2f} "
          f"(arrival behind the run's head {tot['rle_head'] / max(tot['rle_lag'], 1):.2f}, clipped short runs {tot['rle_clip_only'] / n:.3f}), "
          f"clipped extended matches {tot['ext_lag'] / n:.2f}, bytes deleted/chunk {tot['lag_bytes'] / n:.1f}")
    print("  RLE lag run lengths:", " ".join(f"{k}:{v}" for k, v in sorted(runlens.items())))


if __name__ == "__main__":
    main()
