"""One long stream decoded a few times: the target of `rocprofv3 --kernel-trace --stats`.  Dev tool.
   usage: EXT=0|1 python tools/long_dec_prof.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import tamp_amd
from tamp_amd import workloads as wl
ext = os.environ.get('EXT', '0') == '1'
blob = wl.real_text('prose') + wl.real_text('python') + wl.real_text('markup')
data = (blob * (100_000_000 // len(blob) + 1))[:100_000_000]
t0 = time.time(); c = tamp_amd.compress(data, extended=ext); t1 = time.time()
print(f"extended={ext}: compress() {t1-t0:.2f} s, {len(c)} bytes", flush=True)
for rep in range(3):
    r = tamp_amd.decompress_batch([c], out_cap=len(data) + 64, timing=True)
    t0 = time.time(); d = tamp_amd.decompress(c); t1 = time.time()
    print(f"kernels {r.kernel_ms:.2f} ms = {len(data)/r.kernel_ms/1e6:.2f} GB/s out; decompress() {t1-t0:.3f} s = {len(data)/(t1-t0)/1e6:.0f} MB/s; equal {bytes(d) == data and bytes(r.stream(0)) == data}", flush=True)
