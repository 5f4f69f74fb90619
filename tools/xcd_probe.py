"""Why a static one-workgroup-per-stream grid loses on the tiled corpora: workgroup i runs on XCD i % 8, and the frozen corpora are
768 chunks long -- a multiple of 8 -- so each XCD compresses the same 96 chunks over and over.  Period 768 vs 767 vs a shuffle,
persistent grid (default) vs TAMP_AMD_STATIC_GRID=1.  Dev tool (needs an MI355X)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def run(rows, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms = []
    for it in range(4):
        r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, timing=True, **kw); ms.append(r.kernel_ms)
    return min(ms[1:])
rng = np.random.default_rng(5)
for name in ('prose', 'python'):
    blob = wl.real_text(name)
    n = len(blob) // 4096
    base = np.frombuffer(blob[:n * 4096], dtype=np.uint8).reshape(n, 4096)
    for label, idx in (('period %d' % n, np.arange(N) % n), ('period %d' % (n - 1), np.arange(N) % (n - 1)), ('shuffled', rng.permutation(np.arange(N) % n))):
        rows = np.ascontiguousarray(base[idx])
        for grid in ('persistent', 'static'):
            if grid == 'static': os.environ['TAMP_AMD_STATIC_GRID'] = '1'
            else: os.environ.pop('TAMP_AMD_STATIC_GRID', None)
            ms = run(rows)
            print(f"{name:7s} n={N} {label:12s} {grid:10s} {ms:7.3f} ms {N * 4096 / ms / 1e6:6.2f} GB/s", flush=True)
os.environ.pop('TAMP_AMD_STATIC_GRID', None)
rows = wl.synth_text(N, 4096)
for grid in ('persistent', 'static'):
    if grid == 'static': os.environ['TAMP_AMD_STATIC_GRID'] = '1'
    else: os.environ.pop('TAMP_AMD_STATIC_GRID', None)
    ms = run(rows)
    print(f"synth   n={N} {'':12s} {grid:10s} {ms:7.3f} ms {N * 4096 / ms / 1e6:6.2f} GB/s", flush=True)
