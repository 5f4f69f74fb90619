"""Latency of ONE 4 KiB stream alone on the device, for every distinct chunk of the configs[2] stand-in (2,304 chunks of the
three frozen corpora): what bounds a small batch from below is its slowest stream (one workgroup per stream).  Dev tool (GPU box).
   usage: python tools/solo_latency.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0'); L = 4096
for name in ('prose', 'markup', 'python'):
    flat = np.frombuffer(wl.real_text(name, frozen_only=True), dtype=np.uint8)
    k = flat.size // L
    rows = flat[:k * L].reshape(k, L)
    data = torch.from_numpy(rows.reshape(-1).copy()).to(dev)
    off1 = torch.zeros(1, dtype=torch.int64, device=dev); len1 = torch.full((1,), L, dtype=torch.int32, device=dev)
    lat = []
    for i in range(k):
        d = data[i * L:(i + 1) * L]
        ms = [float(tamp_amd.compress_batch(d, off1, len1, max_in_len=L, timing=True).kernel_ms) for _ in range(2)]
        lat.append(min(ms))
    lat = np.array(lat) * 1e3
    print(f"{name:7s} {k} chunks alone: mean {lat.mean():.0f} us  p50 {np.percentile(lat,50):.0f}  p90 {np.percentile(lat,90):.0f}  p99 {np.percentile(lat,99):.0f}  max {lat.max():.0f} us", flush=True)
    # one full round of distinct chunks, all at once (k <= 1792 workgroup slots): contention without queueing
    off, ln = wl.csr_for_fixed(k, L)
    r = [float(tamp_amd.compress_batch(data, torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev), max_in_len=L, timing=True).kernel_ms) for _ in range(3)]
    print(f"{name:7s} all {k} chunks in one launch: {min(r)*1e3:.0f} us", flush=True)
