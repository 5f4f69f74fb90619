"""One long v1 stream through tamp_amd.decompress (tamp_decompress_long_kernel.hpp) against the input it came from and, for status /
sizes / consumed counts, against the oracle's decoder.  Dev tool (GPU box).   usage: python tools/long_dec_check.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle
oracle = Oracle()
def one(name, data, **kw):
    c = tamp_amd.compress(data, extended=False, **kw)
    t0 = time.time(); d = tamp_amd.decompress(c, dictionary=kw.get('dictionary')); t1 = time.time()
    ok = d == data
    print(f"{name:34s} {len(data):>10d} B -> {len(c):>9d} B  decode {t1-t0:6.3f} s = {len(data)/(t1-t0)/1e6:8.1f} MB/s  equal {ok}", flush=True)
    return ok
prose = wl.real_text('prose'); py = wl.real_text('python')
big = (prose * 12)[:30_000_000]
good = True
good &= one('prose 30 MB w10', big)
rr = tamp_amd.decompress_batch([tamp_amd.compress(big, extended=False)], out_cap=31_000_000, timing=True); print('   kernels', rr.kernel_ms, 'ms')
good &= one('prose 30 MB w10 again', big)
good &= one('python 3 MB w12', py[:3_000_000], window=12)
good &= one('python 3 MB w8', py[:3_000_000], window=8)
good &= one('prose 2 MB w15', prose[:2_000_000], window=15)
good &= one('7-bit text 2 MB literal=7', bytes(b & 127 for b in prose[:2_000_000]), literal=7)
good &= one('zeros 4 MB', bytes(4_000_000))
rng = np.random.default_rng(3)
good &= one('random 1 MB', rng.integers(0, 256, 1_000_000, dtype=np.uint8).tobytes())
dic = (prose[5000:6024] * 1)[:1024]
good &= one('prose 2 MB custom dictionary', prose[:2_000_000], dictionary=dic)
# statuses and counts against the oracle's decoder, through the batch surface
c = tamp_amd.compress(big[:4_000_000], extended=False)
for cut in (0, 1, 2, 3, 7):
    blob = c[:len(c) - cut]
    r = tamp_amd.decompress_batch([blob], out_cap=4_100_000)
    st, out, used = oracle.decompress(blob, cap=4_100_000)
    same = int(r.status[0]) == st and bytes(r.stream(0)) == out and (r.in_consumed is None or int(r.in_consumed[0]) == used)
    print(f"truncated by {cut}: status {int(r.status[0])} / oracle {st}, {len(r.stream(0))} / {len(out)} B, same {same}", flush=True)
    good &= same
print("ALL GOOD" if good else "FAILURES")
