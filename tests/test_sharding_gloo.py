"""CPU tier: the multi-GPU path with world_size 2 over gloo.

Streams shard embarrassingly: each rank takes a contiguous stream range from
``tamp_amd.partition_streams`` and there is no collective on the data path.  On CPU there is no codec (the
product has no CPU path), so each rank stands in for its device with the oracle; what this test pins is the
sharding logic bench.py uses: the ranges tile [0, n), per-rank outputs concatenate to the single-process
result, and the only cross-rank traffic (a barrier + MAX of elapsed time + SUM of totals) works.
"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, slen, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle.checker import Oracle
    from tamp_amd import partition_streams
    from tamp_amd import workloads as wl

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    in_len = np.full(n, slen, dtype=np.uint32)
    begin, end = partition_streams(in_len, world)[rank]
    rows = wl.synth_text(end - begin, slen, first_index=begin)  # rank-local generation of exactly its shard
    off, ln = wl.csr_for_fixed(end - begin, slen)
    res = Oracle().compress_batch(rows.reshape(-1), off, ln, window=10, literal=8, extended=True, threads=1)
    dist.barrier()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([int(res.out_len.sum()), end - begin], dtype=torch.int64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    q.put((rank, begin, end, [res.stream(i) for i in range(end - begin)], float(t.item()), tot.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    from oracle.checker import Oracle
    from tamp_amd import workloads as wl

    n, slen, world = 12, 1024, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, slen, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = wl.synth_text(n, slen)
    off, ln = wl.csr_for_fixed(n, slen)
    want = Oracle().compress_batch(rows.reshape(-1), off, ln, threads=1)
    streams = []
    assert got[0][1] == 0 and got[-1][2] == n and got[0][2] == got[1][1]
    for _, _, _, s, tmax, tot in got:
        streams += s
        assert abs(tmax - 0.2) < 1e-9 and tot == [int(want.out_len.sum()), n]
    assert streams == [want.stream(i) for i in range(n)]
