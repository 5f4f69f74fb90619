"""One corpus, one configuration, a few launches: the target of rocprofv3 --pmc runs.  Dev tool.
   usage: CORPUS=prose|python|synth EXT=1 RUNS=1 python tools/one_corpus.py [n_streams]"""
import sys, os, glob
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
which = os.environ.get('CORPUS', 'prose')
if which == 'synth':
    rows = wl.synth_text(N, 4096)
else:
    rows = wl.tile_rows(wl.real_text(which), N, 4096)  # frozen fixtures (tests/golden/make_corpus.py)
off, ln = wl.csr_for_fixed(N, 4096)
data = torch.from_numpy(rows.reshape(-1).copy()).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
ra = {'0': False, '1': True}.get(os.environ.get('RUNS', ''), None)
ms = []
for it in range(4):
    r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=4096, timing=True, extended=os.environ.get('EXT', '1') == '1', run_aware=ra)
    ms.append(r.kernel_ms)
print(f"{which} N={N} ext={os.environ.get('EXT','1')} run_aware={ra}: {min(ms):.2f} ms {N*4096/min(ms)/1e6:.2f} GB/s")
