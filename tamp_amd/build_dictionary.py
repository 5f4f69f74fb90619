"""``python -m tamp_amd build-dictionary``: a custom dictionary for a corpus of short messages.

Replaces the reference's offline tool (``tamp/cli/build_dictionary.py:706-927`` with the kernels of
``tamp/_c_build_dictionary.pyx``).  Kept: the command's options, the output contract -- the file holds only the
*effective* bytes, which ``compress`` / ``decompress --dictionary`` put at the END of a seeded window
(``tamp/cli/main.py:90-105``, here ``cli.load_dictionary``) -- the packing rule (what appears late in a message and saves
most per byte sits furthest right, where the window is overwritten last: ``build_dictionary.py:617-673``), the
compression-versus-size table with its knee, and the sweep over the trim threshold when none is given
(``:426-490``).  The reference itself says its output may change between releases (``:744-746``): there is no byte-exact bar
here, the bar is "the dictionary helps, measurably, on the corpus".

Different, because this package is a batch engine: every "compress the whole corpus with dictionary D" -- one per fill
level of the table, one per trim threshold of the sweep, the baseline and the final check; the reference loops
``tamp.compress`` over the samples for each (``:464-477,517-528``) -- is ONE ``tamp_amd.compress_batch`` launch with a
shared custom dictionary (the shape of BASELINE configs[4]).  Candidate mining is a numpy n-gram census over the heads of
the samples followed by a lazy greedy cover; it is host code, its own design, and not on the codec path.
"""
from __future__ import annotations

import heapq
import sys
from pathlib import Path
from typing import Callable, Iterable, Optional, Sequence

import numpy as np

# match-length prefix code, lengths including the flag bit (format fact: compressor.c:33-36); symbols 12 / 13 / 14 are
# RLE / extended match / FLUSH
_CODE_BITS = (2, 3, 5, 5, 6, 7, 7, 7, 8, 8, 9, 9, 9, 7, 9)
_LENGTH_LADDER = (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 24, 28, 32, 40, 48, 64)


def bits_saved(length: int, min_pattern: int, window: int, literal: int, extended: bool) -> float:
    """Bits a match of ``length`` bytes saves over ``length`` literals: a plain token costs its prefix code + ``window``
    offset bits (lengths up to min+13; up to min+11 in the extended format), an extended match costs symbol 13, the
    prefix-coded high part of ``length - min - 12`` with three trailing bits, and the offset (compressor.c:257-263,377-415)."""
    lit = length * (1 + literal)
    k = length - min_pattern
    best = float("inf")
    if 0 <= k <= (11 if extended else 13):
        best = _CODE_BITS[k] + window
    if extended and k >= 12:
        hi = (k - 12) >> 3
        if hi < 15:
            best = min(best, _CODE_BITS[13] + (_CODE_BITS[hi] - 1) + 3 + window)
    return max(0.0, lit - best) if best != float("inf") else 0.0


def read_corpus(path, delimiter: Optional[str] = "\n") -> list:
    """Samples of a corpus: every non-empty file of a directory (sorted by name), or one file cut at ``delimiter``."""
    p = Path(path)
    if p.is_dir():
        out = [f.read_bytes() for f in sorted(q for q in p.iterdir() if q.is_file())]
        out = [s for s in out if s]
        if not out:
            raise ValueError(f"No files found in {p}")
        return out
    if p.is_file():
        sep = (delimiter if delimiter is not None else "\n").encode()
        return [s for s in p.read_bytes().split(sep) if s]
    raise ValueError(f"Input path does not exist: {p}")


class _Census:
    """The first ``W`` bytes of every sample laid end to end, and rolling hashes of all substrings for a ladder of lengths.
    Only a message's first ``W`` bytes can reach the dictionary at all: byte i of the window is overwritten after i bytes."""

    def __init__(self, corpus: Sequence[bytes], window: int, max_bytes: int = 2 << 20):
        W = 1 << window
        heads = [bytes(s[:W]) for s in corpus]
        total = sum(len(h) for h in heads)
        if total > max_bytes:  # an even sample of the messages: the census is statistics, not an index
            step = -(-total // max_bytes)
            heads = heads[::step]
        self.W = W
        self.buf = np.frombuffer(b"".join(heads), dtype=np.uint8)
        lens = np.array([len(h) for h in heads], dtype=np.int64)
        self.start = np.repeat(np.cumsum(lens) - lens, lens)        # start of the sample a byte belongs to
        self.room = np.repeat(np.cumsum(lens), lens) - np.arange(len(self.buf))  # bytes left in its sample
        self.nsamples = len(heads)
        self._h = {}

    def hashes(self, L: int) -> np.ndarray:
        """64-bit polynomial hash of buf[i : i + L] for every i (garbage where the substring leaves its sample)."""
        if L not in self._h:
            h = np.zeros(len(self.buf), dtype=np.uint64)
            b = self.buf.astype(np.uint64)
            n = len(b)
            for j in range(L):
                h[: n - j] = h[: n - j] * np.uint64(0x9E3779B97F4A7C15) + b[j:] + np.uint64(1)
            self._h = {L: h}  # (one length at a time: callers work through the ladder length by length)
        return self._h[L]

    def occurrences(self, entry: bytes) -> np.ndarray:
        L = len(entry)
        e = np.frombuffer(entry, dtype=np.uint8)
        h = self.hashes(L)
        key = 0
        for c in entry:
            key = (key * 0x9E3779B97F4A7C15 + c + 1) & 0xFFFFFFFFFFFFFFFF
        idx = np.flatnonzero((h == np.uint64(key)) & (self.room >= L))
        if len(idx):  # (hash collisions: compare the bytes)
            ok = np.ones(len(idx), dtype=bool)
            for j in range(L):
                ok &= self.buf[idx + j] == e[j]
            idx = idx[ok]
        return idx

    def candidates(self, L: int, top: int) -> list:
        """The ``top`` most frequent substrings of length ``L`` that occur at least twice."""
        h = self.hashes(L)
        valid = np.flatnonzero(self.room >= L)
        if len(valid) == 0:
            return []
        u, first, cnt = np.unique(h[valid], return_index=True, return_counts=True)
        keep = np.flatnonzero(cnt >= 2)
        keep = keep[np.argsort(cnt[keep])[::-1][:top]]
        return [self.buf[valid[first[k]] : valid[first[k]] + L].tobytes() for k in keep]


def mine_entries(corpus: Sequence[bytes], window: int, literal: int, extended: bool, trim_threshold: int,
                 capacity: Optional[int] = None) -> list:
    """-> [(entry, bits saved on the corpus, Q3 of the relative position where it ends)], chosen by a lazy greedy cover:
    take the candidate that explains the most still-unexplained bytes per bit, mark its occurrences explained, repeat until
    ``capacity`` bytes are chosen.  A candidate contained in a chosen entry is dropped (the longer one serves its matches)."""
    import tamp_amd

    minp = tamp_amd.compute_min_pattern_size(window, literal)
    W = 1 << window
    capacity = W if capacity is None else capacity
    cen = _Census(corpus, window)
    if len(cen.buf) == 0:
        return []
    lengths = [L for L in _LENGTH_LADDER if L >= max(minp, trim_threshold) and L <= W // 4]
    pool = {}
    for L in lengths:
        gain = bits_saved(L, minp, window, literal, extended)
        if gain <= 0:
            continue
        for e in cen.candidates(L, 192):
            pool[e] = (gain, cen.occurrences(e))  # (every occurrence, found once, while this length's hashes are at hand)
    covered = np.zeros(len(cen.buf), dtype=np.int64)
    state = {"cs": np.zeros(len(cen.buf) + 1, dtype=np.int64)}  # prefix sums of `covered`, rebuilt after every acceptance

    def score(entry):
        gain, idx = pool[entry]
        if len(idx) == 0:
            return 0.0, idx
        L = len(entry)
        cs = state["cs"]
        free = (cs[idx + L] - cs[idx]) == 0  # none of its bytes explained yet
        idx = idx[free]
        if len(idx) > 1:  # overlapping occurrences of one entry (periodic text) count once
            keep = [0]
            for k in range(1, len(idx)):
                if idx[k] - idx[keep[-1]] >= L:
                    keep.append(k)
            idx = idx[keep]
        return gain * len(idx), idx

    heap = []
    for e in pool:
        s, _ = score(e)
        if s > 0:
            heapq.heappush(heap, (-s / len(e), e))
    chosen, used = [], 0
    while heap and used < capacity:
        _, e = heapq.heappop(heap)
        if used + len(e) > capacity or any(e in c for c, _, _ in chosen):
            continue
        s, idx = score(e)
        if s <= 0:
            continue
        if heap and -heap[0][0] > s / len(e) * 1.0001:  # stale: somebody else leads now
            heapq.heappush(heap, (-s / len(e), e))
            continue
        for i in idx:
            covered[i : i + len(e)] = 1
        state["cs"] = np.concatenate([[0], np.cumsum(covered)])
        rel = np.sort((idx - cen.start[idx] + len(e)) / float(W))
        q3 = float(rel[min(len(rel) - 1, (3 * len(rel)) // 4)])
        chosen.append((e, float(s), q3))
        used += len(e)
    return chosen


def pack(entries: Sequence[tuple], window: int, literal: int, extended: bool):
    """Entries into a seeded window, right to left: late in the messages / most bits per byte furthest right.
    -> (window bytearray, effective bytes)."""
    import tamp_amd

    W = 1 << window
    out = tamp_amd.initialize_dictionary(W, literal=literal if extended else 8)
    at = W
    for e, s, _pos in sorted(entries, key=lambda t: (t[2], t[1] / len(t[0]), t[0]), reverse=True):
        if s <= 0 or at - len(e) < 0:
            continue
        at -= len(e)
        out[at : at + len(e)] = e
    return out, W - at


def gpu_total(corpus: Sequence[bytes], window: int, literal: int, extended: bool) -> Callable:
    """-> f(dictionary or None) = compressed payload bytes of the whole corpus (headers not counted): one batch launch."""
    import tamp_amd

    flat = np.frombuffer(b"".join(corpus), dtype=np.uint8)
    ln = np.array([len(s) for s in corpus], dtype=np.uint32)
    off = np.zeros(len(ln), dtype=np.uint64)
    off[1:] = np.cumsum(ln[:-1].astype(np.uint64))

    def total(dictionary) -> int:
        kw = {} if dictionary is None else {"dictionary": bytes(dictionary)}
        r = tamp_amd.compress_batch(flat, off, ln, window=window, literal=literal, extended=extended, **kw)
        st = np.asarray(r.status)
        if (st != 0).any():
            raise ValueError("a corpus sample does not fit literal=%d (TAMP_EXCESS_BITS)" % literal)
        return int(np.asarray(r.out_len, dtype=np.int64).sum()) - len(ln)

    return total


def build(corpus, window=10, literal=8, extended=True, trim_threshold=4, target_fill=1.0):
    """-> (window bytearray, effective bytes) at ``target_fill`` of the window."""
    cap = max(0, min(1 << window, int((1 << window) * target_fill)))
    return pack(mine_entries(corpus, window, literal, extended, trim_threshold, cap), window, literal, extended)


def find_knee(points: Sequence[tuple], fraction: float = 0.5) -> int:
    """``points`` = [(dictionary bytes, compressed bytes)], ascending in size.  The knee is the last size whose step still
    buys at least ``fraction`` of the average saving per dictionary byte over the whole range (steps narrower than half
    the typical one are noise and are skipped)."""
    if len(points) <= 2:
        return points[-1][0]
    (x0, y0), (x1, y1) = points[0], points[-1]
    if x1 <= x0 or y0 <= y1:
        return x1
    need = fraction * (y0 - y1) / (x1 - x0)
    typical = (x1 - x0) / (len(points) - 1)
    knee = points[0][0]
    for (xa, ya), (xb, yb) in zip(points, points[1:]):
        if xb - xa >= 0.5 * typical and (ya - yb) / (xb - xa) >= need:
            knee = xb
    return knee


def tradeoff(full, effective: int, window: int, literal: int, extended: bool, total: Callable, n_points: int = 12) -> list:
    """Compressed size of the corpus with only the rightmost S bytes of ``full`` over a seeded window, for a ladder of S."""
    import tamp_amd

    W = 1 << window
    seed = tamp_amd.initialize_dictionary(W, literal=literal if extended else 8)
    if effective == 0:
        return [(0, total(seed))]
    step = max(1, effective // n_points)
    sizes = list(range(step, effective, step))
    if not sizes or sizes[-1] != effective:
        sizes.append(effective)
    out = []
    for s in sizes:
        d = bytearray(seed)
        d[W - s :] = full[W - s :]
        out.append((s, total(d)))
    return out


def build_dictionary_cli(input, output="dictionary.bin", *, window=10, literal=8, extended=True, delimiter="\n",
                         trim_threshold=None, target_fill=None, quiet=False, total: Optional[Callable] = None,
                         log=None) -> dict:
    """The command: read the corpus, build at full size (sweeping the trim threshold if none is given), tabulate the
    tradeoff, pick the knee (or ``target_fill``), rebuild at that size, write the effective bytes.  ``total`` replaces the
    GPU evaluation (tests).  -> a summary dict."""
    log = log if log is not None else (lambda *a: print(*a, file=sys.stderr))
    corpus = read_corpus(input, delimiter)
    if not corpus:
        raise ValueError(f"No samples in {input}")
    W = 1 << window
    total = total if total is not None else gpu_total(corpus, window, literal, extended)
    swept = trim_threshold is None
    best = None
    for tt in ((3, 4, 6, 8) if swept else (trim_threshold,)):
        d, eff = build(corpus, window, literal, extended, tt, 1.0)
        t = total(d)
        if best is None or t < best[0]:
            best = (t, d, eff, tt)
    _, full, eff, trim_threshold = best
    points = tradeoff(full, eff, window, literal, extended, total)
    baseline = total(None)
    knee = find_knee([(0, baseline)] + points) if points[0][0] else points[0][0]
    pick = min((p[0] for p in points), key=lambda s: abs(s - int(W * target_fill))) if target_fill is not None else knee
    if pick <= 0:  # (no step of the table qualifies as the knee: the smallest tabulated size is the selection)
        pick = points[0][0]
    raw = sum(len(s) for s in corpus)
    if not quiet:
        log(f"\nDictionary analysis (window={window}, {W} bytes):")
        log(f"Corpus: {len(corpus)} samples, {raw} bytes total")
        if swept:
            log(f"Auto-tuned trim_threshold: {trim_threshold}")
        log("\nFill  Bytes  Compressed  Ratio  Benefit")
        span = max(1, baseline - points[-1][1])
        for s, c in points:
            ben = 100.0 * (baseline - c) / span
            log(f"{s / W:4.2f}  {s:5d}  {c:10d}  {raw / max(c, 1):4.2f}x  {ben:3.0f}% {'#' * int(ben * 0.4 + 0.5)}{'  <-- selected' if s == pick else ''}")
    fill = target_fill if target_fill is not None else pick / W
    if fill < 1.0:  # a fresh build at the smaller size chooses better than cutting the full one
        d2, eff2 = build(corpus, window, literal, extended, trim_threshold, fill)
        if eff2:
            full, eff = d2, eff2
        else:
            # nothing fits the requested capacity (a target fill below the shortest entry): write the smallest tabulated
            # size of the full build rather than, silently, ALL of it
            eff = min(eff, points[0][0])
    Path(output).write_bytes(bytes(full[W - eff :]))
    final = total(full)
    if not quiet:
        log(f"\nDict size:  {eff} bytes")
        log(f"No dict:    {baseline} bytes compressed ({raw / max(baseline, 1):.2f}x)")
        log(f"With dict:  {final} bytes compressed ({raw / max(final, 1):.2f}x, -{100.0 * (baseline - final) / max(baseline, 1):.1f}% vs no dict)")
    return {"samples": len(corpus), "raw_bytes": raw, "dictionary_bytes": eff, "baseline": baseline, "with_dictionary": final,
            "trim_threshold": trim_threshold, "tradeoff": points, "knee": knee}
