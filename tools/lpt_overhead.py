import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0'); N = 65536
rows = wl.synth_text(N, 4096); off, ln = wl.csr_for_fixed(N, 4096)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
for it in range(6):
    r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=4096, timing=True)
torch.cuda.synchronize(); print(r.kernel_ms)
