import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
def run(name, rows, variants, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, **kw)
    cap = torch.full((n,), L, dtype=torch.int32, device=dev)
    for bpt, wmax in variants:
        os.environ['TAMP_AMD_SPLIT_BPT'] = bpt; os.environ['TAMP_AMD_DECODER'] = 'split'; os.environ['TAMP_AMD_SPLIT_WAVE_MAX'] = wmax
        ms = []
        for it in range(4):
            d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=cap, dictionary=kw.get('dictionary'), timing=True); ms.append(d.kernel_ms)
        ok = bool((d.out[:n*L].cpu().numpy() == rows.reshape(-1)).all())
        print(f"{name} bpt={bpt} wave_max={wmax}: {min(ms):.3f} ms {n*L/min(ms)/1e6:.1f} GB/s ok={ok}", flush=True)
V = (('4', '1024'), ('8', '1024'), ('4', '2048'), ('8', '2048'))
run("text 65536x4K w10", wl.synth_text(65536, 4096), V)
run("text 131072x2K w10", wl.synth_text(131072, 2048), V)

tel_dict = wl.telemetry_dictionary(bytes(tamp_amd.initialize_dictionary(256, literal=7)))
run("telemetry 1Mx256 dict", wl.telemetry(1 << 20, 256), (('4', '1024'), ('8', '1024')), window=8, literal=7, dictionary=tel_dict)
