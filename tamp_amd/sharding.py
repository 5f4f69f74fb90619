"""Multi-GPU sharding of a batch: independent streams, no exchange step.

Streams are cut into ``world_size`` contiguous index ranges balanced by total input bytes; rank ``r``
compresses its range on its own device with its own HIP stream.  No collective is on the data path --
the only cross-rank traffic is the caller's optional reduction of per-rank totals.
"""
from __future__ import annotations

import numpy as np


def partition_streams(in_len, world_size: int):
    """-> list of (begin, end) stream index ranges, one per rank, contiguous and covering [0, n).

    Balanced by the prefix sum of ``in_len`` (equal counts for equal-sized streams); a rank may get an
    empty range when there are fewer streams than ranks.
    """
    lens = np.asarray(in_len, dtype=np.uint64)
    n = int(lens.size)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if n == 0:
        return [(0, 0)] * world_size
    prefix = np.concatenate([[0], np.cumsum(lens)])
    total = int(prefix[-1])
    if total == 0:
        cuts = [n * r // world_size for r in range(world_size + 1)]
    else:
        targets = [total * r // world_size for r in range(world_size + 1)]
        cuts = [int(np.searchsorted(prefix, t, side="left")) for t in targets]
        cuts[0], cuts[-1] = 0, n
        for r in range(1, world_size + 1):
            cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def shard_for_rank(in_off, in_len, rank: int, world_size: int):
    """-> (begin, end, byte_begin, byte_end) of rank's contiguous shard (streams must be packed in order)."""
    begin, end = partition_streams(in_len, world_size)[rank]
    off = np.asarray(in_off, dtype=np.uint64)
    lens = np.asarray(in_len, dtype=np.uint64)
    if begin == end:
        return begin, end, 0, 0
    return begin, end, int(off[begin]), int(off[end - 1] + lens[end - 1])
