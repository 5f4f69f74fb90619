"""Throughput on run-heavy / degenerate inputs (LCG fuzz corpus, stress shapes, zeros, telemetry).  Dev tool."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
if os.environ.get('TAMP_VAR'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])
import tamp_amd, hashlib
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
def run(name, rows, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms=[]
    for it in range(3):
        r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, timing=True, **kw); ms.append(r.kernel_ms)
    h = hashlib.sha256(r.out.cpu().numpy().tobytes()).hexdigest()[:8]
    print(f"{os.environ.get('TAMP_VAR')} {name:22s} {min(ms):8.2f} ms {n*L/min(ms)/1e6:7.2f} GB/s sha {h}", flush=True)
run("lcg_runs ext", wl.lcg_runs(16384, 4096))
run("lcg_runs v1", wl.lcg_runs(16384, 4096), extended=False)
run("stress ext", wl.stress(16383, 4096))
run("stress v1", wl.stress(16383, 4096), extended=False)
run("zeros v1", np.zeros((4096, 4096), np.uint8), extended=False)
run("zeros ext", np.zeros((16384, 4096), np.uint8))
run("telemetry w8 l7", wl.telemetry(1<<18, 256), window=8, literal=7)
