"""Experiment: two split-decoder calls (half the batch each) in flight at once on two HIP streams, from two host threads,
   against the same batch in one call -- do PARSE (serial chains, VALU issue) and RESOLVE (LDS traffic) overlap?
   usage: python tools/dec_overlap.py [n]"""
import os, sys, time, threading
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 524288, 4096
rows = wl.synth_text(65536, L); off, ln = wl.csr_for_fixed(65536, L)
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L)
rep = n // 65536
in_off = r.out_off.repeat(rep); in_len = r.out_len.repeat(rep)   # the same compressed streams, decoded n / 65536 times over
os.environ['TAMP_AMD_DECODER'] = 'split'
streams = [torch.cuda.Stream() for _ in range(4)]
def one(lo, hi, st, res, i):
    with torch.cuda.stream(st):
        res[i] = tamp_amd.decompress_batch(r.out, in_off[lo:hi], in_len[lo:hi], out_cap=L + 8, stream=st.cuda_stream)
for parts in (1, 2, 4):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        res = [None] * parts
        t0 = time.perf_counter()
        th = [threading.Thread(target=one, args=(n * i // parts, n * (i + 1) // parts, streams[i], res, i)) for i in range(parts)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = all(bool((o.status == 2).all().item()) for o in res)
    print(f"n={n} in {parts} concurrent call(s): {best*1e3:7.2f} ms wall {n*L/best/1e9:6.1f} GB/s out ok={ok}", flush=True)
