"""configs[3]-shaped decode (262,144 streams, windows 2^8..2^12) under different split-scratch budgets.  Dev tool."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
print("mem_get_info (free, total) GB:", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info()])
n, L = 262144, 4096
base = wl.synth_text(n // 5 + 1, L, first_index=1 << 20)
streams = []
outs = {}
for w in range(8, 13):
    rows = base[: n // 5 + 1]
    off, ln = wl.csr_for_fixed(len(rows), L)
    r = tamp_amd.compress_batch(torch.from_numpy(rows.reshape(-1)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev), window=w, max_in_len=L)
    outs[w] = r
# interleave: stream i uses window 8 + i % 5
lens = torch.stack([outs[8 + k].out_len.to(torch.int64)[: n // 5] for k in range(5)], 1).reshape(-1)
offs_src = torch.stack([outs[8 + k].out_off.to(torch.int64)[: n // 5] for k in range(5)], 1).reshape(-1)
which = (torch.arange(lens.numel(), device=dev) % 5)
in_off = torch.cumsum(lens, 0) - lens
slab = torch.empty(int(lens.sum().item()) + 64, dtype=torch.uint8, device=dev)
for k in range(5):
    ids = torch.nonzero(which == k).flatten()
    ll = lens[ids]
    rep = torch.repeat_interleave(torch.arange(len(ids), device=dev), ll)
    within = torch.arange(int(ll.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(ll, 0) - ll, ll)
    slab[in_off[ids][rep] + within] = outs[8 + k].out[offs_src[ids][rep] + within]
m = lens.numel()
for env in (None, "8192", "2048", "512"):
    if env is None: os.environ.pop("TAMP_AMD_SPLIT_SCRATCH_MB", None)
    else: os.environ["TAMP_AMD_SPLIT_SCRATCH_MB"] = env
    tamp_amd.trim(0)
    ms = []
    for _ in range(4):
        d = tamp_amd.decompress_batch(slab, in_off, lens.to(torch.int32), out_cap=L + 8, timing=True)
        ms.append(float(d.kernel_ms))
    print(f"SCRATCH_MB={env}: {['%.2f' % x for x in ms]} ms  streams={m} ok={bool((d.status == 2).all().item())}")
