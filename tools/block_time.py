import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
blob = wl.real_text('prose'); n = 100_000_000
flat = np.frombuffer((blob * (n // len(blob) + 1))[:n], dtype=np.uint8).copy()
d = torch.from_numpy(flat).to(dev); off = torch.zeros(1, dtype=torch.int64, device=dev); ln = torch.tensor([n], dtype=torch.int32, device=dev)
for it in range(4):
    r = tamp_amd.compress_batch(d, off, ln, window=10, literal=8, extended=False, max_in_len=n, timing=True)
    torch.cuda.synchronize(); print(f"kernel {r.kernel_ms:.2f} ms out {int(r.out_len[0])}", flush=True)
