"""The bench's decode leg alone (65,536 x 4 KiB synthetic text compressed at w=10, decoded by what the launcher picks): the
target of the FETCH_SIZE / WRITE_SIZE passes behind also.decompress.roofline.traffic.  Dev tool (GPU box)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
n, L = 65536, 4096
dev = torch.device('cuda:0')
rows = wl.synth_text(n, L)
off, ln = wl.csr_for_fixed(n, L)
r = tamp_amd.compress_batch(torch.from_numpy(rows.reshape(-1)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev),
                            torch.from_numpy(ln.astype(np.int32)).to(dev), window=10, max_in_len=L)
for _ in range(3):
    d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L, timing=True)
print(f"decode {d.kernel_ms:.3f} ms, compressed {int(r.out_len.to(torch.int64).sum())} B, out {n * L} B")
