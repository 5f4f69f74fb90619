"""Same-box A/B timing of library variants: TAMP_VAR=X loads tamp_amd/libtamp_varX.so instead of libtamp_amd.so.
   usage: VARS="A B" bash tools/ab.sh"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
if os.environ.get('TAMP_VAR'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])
import tamp_amd
from tamp_amd import workloads as wl
n = 65536
rows = wl.synth_text(n, 4096); off, ln = wl.csr_for_fixed(n, 4096)
dev = torch.device('cuda:0')
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
cap = torch.full((n,), 4609, dtype=torch.int32, device=dev)
out = []
for ext in ((1,) if os.environ.get('AB_ONLY_EXT') == '1' else (0,) if os.environ.get('AB_ONLY_EXT') == '0' else (1, 0)):
    ms = []
    for it in range(6):
        r = tamp_amd.compress_batch(data, off_t, len_t, extended=bool(ext), max_in_len=4096, out_cap=cap, timing=True)
        ms.append(r.kernel_ms)
    out.append(f"ext={ext} {min(ms):.3f} ms ({n*4096/min(ms)/1e6:.1f} GB/s)")
import hashlib
print(os.environ.get('TAMP_VAR'), ' | '.join(out), 'sha', hashlib.sha256(r.out.cpu().numpy().tobytes()[:1<<20]).hexdigest()[:8], flush=True)
