import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
blob = wl.real_text('prose') + wl.real_text('python') + wl.real_text('markup')
data = (blob * (100_000_000 // len(blob) + 1))[:100_000_000]
c = tamp_amd.compress(data, extended=False)
for w in (10, 12):
    cc = c if w == 10 else tamp_amd.compress(data[:40_000_000], extended=False, window=12)  # (2^15 windows stay out of block mode: a minute)
    want = data if w == 10 else data[:40_000_000]
    for rep in range(3):
        t0 = time.time(); d = tamp_amd.decompress(cc); t1 = time.time()
        r = tamp_amd.decompress_batch([cc], out_cap=len(want) + 64, timing=True)
        print(f"w{w}: {len(want)} B, decompress() {t1-t0:.3f} s = {len(want)/(t1-t0)/1e6:.0f} MB/s; kernels {r.kernel_ms:.1f} ms = {len(want)/r.kernel_ms/1e6:.2f} GB/s; equal {bytes(d) == want and bytes(r.stream(0)) == want}", flush=True)
