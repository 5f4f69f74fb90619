"""Experiment: the LDS-row lane decoder and the global-window lane decoder on two halves of one batch, concurrently on two
   HIP streams (they are bound by different resources: LDS capacity / cache-resident windows).  usage: python tools/dec_split.py [n]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 4096
rows = wl.synth_text(n, L); off, ln = wl.csr_for_fixed(n, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L)
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(frac_lds, reps=3):
    k = int(n * frac_lds) // 64 * 64
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        if k:
            os.environ['TAMP_AMD_DECODER'] = 'lane'
            with torch.cuda.stream(sA):
                outs.append(tamp_amd.decompress_batch(r.out, r.out_off[:k], r.out_len[:k], out_cap=L + 8, max_window_bits=10, scan_headers=False, stream=sA.cuda_stream))
        if k < n:
            os.environ['TAMP_AMD_DECODER'] = 'global'
            with torch.cuda.stream(sB):
                outs.append(tamp_amd.decompress_batch(r.out, r.out_off[k:], r.out_len[k:], out_cap=L + 8, max_window_bits=10, scan_headers=False, stream=sB.cuda_stream))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = all(bool((o.status == 2).all().item()) for o in outs)
    print(f"n={n} LDS-lane share {frac_lds:.2f}: {best*1e3:7.2f} ms  {n*L/best/1e9:6.1f} GB/s out  ok={ok}", flush=True)
for f in (0.0, 1.0, 0.3, 0.4, 0.5, 0.6):
    run(f)
