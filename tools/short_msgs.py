"""Short-message compress bench (config 5 shape): 256-byte messages, window=8 literal=7 extended, shared custom
   dictionary.  Two data sets: `padded` (workloads.telemetry: one JSON record, space padded) and `dense` (JSON records back
   to back, cut at 256 bytes).  A sample of each is compared with the oracle.  usage: python tools/short_msgs.py [n]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
if os.environ.get('TAMP_VAR'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle
dev = torch.device('cuda:0')
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, int(os.environ.get('SLEN', '256'))
d = wl.telemetry_dictionary(bytes(tamp_amd.initialize_dictionary(256, literal=7)))
def dense_rows(n):
    src = wl.telemetry(4 * n + 8, 256)
    out = np.full((n, L), 0x20, dtype=np.uint8)
    lens = (src != 0x20).sum(axis=1)  # JSON part (no spaces inside)
    k = 0
    for i in range(n):
        p = 0
        while p < L:
            m = min(int(lens[k]), L - p)
            out[i, p:p + m] = src[k, :m]
            p += m; k += 1
            if k >= len(src): k = 0
    return out
sets = {'padded': wl.telemetry(n, L) if L == 256 else None, 'dense': None}
small = dense_rows(8192)
sets['dense'] = np.tile(small, ((n + 8191) // 8192, 1))[:n].copy()
for name, rows in sets.items():
    if rows is None: continue
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t = torch.arange(n, dtype=torch.int64, device=dev) * L
    len_t = torch.full((n,), L, dtype=torch.int32, device=dev)
    ms = []
    for it in range(4):
        r = tamp_amd.compress_batch(data, off_t, len_t, window=8, literal=7, dictionary=d, max_in_len=L, timing=True); ms.append(r.kernel_ms)
    k = 2048
    off, ln = wl.csr_for_fixed(k, L)
    want = Oracle().compress_batch(rows[:k].reshape(-1), off, ln, window=8, literal=7, dictionary=d, threads=8)
    ok = all(r.stream(i) == want.stream(i) for i in range(k))
    print(f"{name:7s} {n} x {L} B: {min(ms):7.2f} ms {n*L/min(ms)/1e6:6.1f} GB/s ratio {float(r.out_len.sum().item())/(n*L):.3f} sample == oracle: {ok}", flush=True)
