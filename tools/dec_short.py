"""Decoder timing on batches of short messages (BASELINE configs[4] shape) and on the 4 KiB text batch: lane decoder against the
split decoder (wave-per-stream resolve for out_cap <= 1 KiB).  Dev tool (GPU box)."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
def run(name, rows, modes=('lane', 'split', 'auto'), **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, **kw)
    cap = torch.full((n,), L, dtype=torch.int32, device=dev)
    for mode in modes:
        os.environ['TAMP_AMD_DECODER'] = mode
        if mode == 'auto': del os.environ['TAMP_AMD_DECODER']
        ms = []
        for it in range(4):
            d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=cap, dictionary=kw.get('dictionary'), timing=True)
            ms.append(d.kernel_ms)
        st = d.status.cpu().numpy() if hasattr(d.status, 'cpu') else np.asarray(d.status)
        ok = bool((d.out[:n*L].cpu().numpy() == rows.reshape(-1)).all()) and set(np.unique(st).tolist()) <= {1, 2}
        print(f"{name:28s} {mode:6s}: {min(ms):7.3f} ms  {n*L/min(ms)/1e6:7.1f} GB/s out  ok={ok}", flush=True)
tel_dict = wl.telemetry_dictionary(bytes(tamp_amd.initialize_dictionary(256, literal=7)))
run("telemetry 1Mx256 w8 l7", wl.telemetry(1 << 20, 256), window=8, literal=7)
run("telemetry 1Mx256 w8 l7 dict", wl.telemetry(1 << 20, 256), window=8, literal=7, dictionary=tel_dict)
run("telemetry 64Kx256 w8 l7", wl.telemetry(1 << 16, 256), window=8, literal=7)
run("telemetry 1Mx256 w10 l8", wl.telemetry(1 << 20, 256))
run("telemetry 16Kx256 w10 l8", wl.telemetry(1 << 14, 256))
run("text 65536x1K w8", wl.synth_text(65536, 1024), window=8)
run("text 262144x1K w10", wl.synth_text(262144, 1024))
run("text 1Mx512 w9", wl.synth_text(1 << 20, 512), window=9)
run("text 65536x4K w10 ext", wl.synth_text(65536, 4096), modes=('split', 'auto'))
