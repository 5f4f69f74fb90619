"""Kernel time of the compressor on the headline batch and on the frozen corpora, for whatever library TAMP_AMD_LIB names.  Dev tool."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
N = 65536
def run(rows, reps=8, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms = []
    for it in range(reps):
        r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, timing=True, **kw); ms.append(r.kernel_ms)
    return min(ms[1:]), float(np.median(ms[1:]))
tag = os.path.basename(os.environ.get('TAMP_AMD_LIB', 'libtamp_amd.so'))
print(tag, 'synthetic ext  min %.3f median %.3f ms' % run(wl.synth_text(N, 4096)), flush=True)
print(tag, 'synthetic v1   min %.3f median %.3f ms' % run(wl.synth_text(N, 4096), extended=False), flush=True)
rng = np.random.default_rng(5)
for name in ('prose', 'markup', 'python'):
    blob = wl.real_text(name); n = len(blob) // 4096
    base = np.frombuffer(blob[:n * 4096], dtype=np.uint8).reshape(n, 4096)
    rows = np.ascontiguousarray(base[rng.permutation(np.arange(N) % n)])
    print(tag, name, 'shuffled ext  min %.3f median %.3f ms' % run(rows, reps=4), flush=True)
