"""Throughput on real text found on the box (Python stdlib sources, licence prose) next to config 2.  Dev tool."""
import sys, os, glob
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
if os.environ.get('TAMP_VAR'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle, Ref
dev = torch.device('cuda:0')
def corpus(patterns, nbytes):
    buf = bytearray()
    for pat in patterns:
        for f in sorted(glob.glob(pat)):
            try: buf += open(f, 'rb').read()
            except Exception: pass
            if len(buf) >= nbytes: return bytes(buf[:nbytes])
    return bytes(buf)
def run(name, rows, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms=[]
    for it in range(3):
        r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, timing=True, **kw); ms.append(r.kernel_ms)
    olen = r.out_len.cpu().numpy(); offs = r.out_off.cpu().numpy(); outh = r.out.cpu().numpy()
    k = min(n, 512)
    want = Oracle().compress_batch(rows[:k].reshape(-1), *wl.csr_for_fixed(k, L), threads=16, **kw)
    ok = all(outh[offs[i]:offs[i]+olen[i]].tobytes()==want.stream(i) for i in range(k))
    cpu = ""
    if Ref.available():
        kk = min(n, 4096)
        rr = Ref().compress_batch(rows[:kk].reshape(-1), *wl.csr_for_fixed(kk, L), threads=32, **kw)
        cpu = f" | reference C 32 thr: {kk*L/rr.seconds/1e9:5.2f} GB/s"
    print(f"{name:34s} n={n} {min(ms):7.2f} ms {n*L/min(ms)/1e6:6.2f} GB/s ratio={olen.sum()/rows.size:.3f} parity={ok}{cpu}", flush=True)
N = 16384
py = wl.real_text('python')   # frozen fixtures (tests/golden/make_corpus.py)
md = wl.real_text('prose')
print(len(py), len(md))
for name, blob in (("python sources (frozen)", py), ("prose (frozen)", md), ("markup (frozen)", wl.real_text('markup'))):
    n = len(blob)//4096
    if n == 0: continue
    rows = np.frombuffer(blob[:n*4096], dtype=np.uint8).reshape(n, 4096).copy()
    if n < N: rows = np.tile(rows, ((N+n-1)//n, 1))[:N]
    run(name + " ext", rows)
    run(name + " v1", rows, extended=False)
run("synthetic text ext", wl.synth_text(N, 4096))
