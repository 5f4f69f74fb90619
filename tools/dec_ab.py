"""Decode time of the bench batch (65,536 x 4 KiB) and of configs[3]-like mixes for whatever library TAMP_AMD_LIB names: min / median of 10.  Dev tool."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
tag = os.path.basename(os.environ.get('TAMP_AMD_LIB', 'libtamp_amd.so'))
for name, n, L, w, src in (('synthetic 65536x4K', 65536, 4096, 10, None), ('prose 65536x4K', 65536, 4096, 10, 'prose'), ('synthetic 262144x4K w12', 262144, 4096, 12, None), ('telemetry 1Mx256', 1 << 20, 256, 8, 'tel')):
    if src == 'tel': rows = wl.telemetry(n, L)
    elif src: rows = wl.tile_rows(wl.real_text(src), n, L)
    else: rows = wl.synth_text(n, L)
    off, ln = wl.csr_for_fixed(n, L)
    kw = dict(window=w, literal=7) if src == 'tel' else dict(window=w)
    r = tamp_amd.compress_batch(torch.from_numpy(rows.reshape(-1)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev), max_in_len=L, **kw)
    ms = []
    for _ in range(11):
        d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L, timing=True); ms.append(float(d.kernel_ms))
    ok = bool((d.out_len == L).all().item()) and bool(torch.equal(d.out.view(-1)[: n * L].view(n, L)[:256].cpu(), torch.from_numpy(rows[:256])))
    print(f"{tag} decode {name}: min {min(ms[1:]):.3f} median {float(np.median(ms[1:])):.3f} ms ok={ok}", flush=True)
