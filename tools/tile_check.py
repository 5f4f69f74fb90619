"""Quick parity check of the tile-ring compress kernel against the reference C / oracle.  Dev tool (GPU box)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle, Ref
chk = Ref() if Ref.available() else Oracle()
dev = torch.device('cuda:0')
def run(name, flat, off, ln, **kw):
    t0 = time.time()
    got = tamp_amd.compress_batch(flat, off, ln, max_in_len=int(ln.max()) if len(ln) else 0, **kw)
    torch.cuda.synchronize()
    kw2 = dict(kw); kw2.pop('run_aware', None)
    want = chk.compress_batch(flat, off, ln, threads=16, **kw2)
    st = np.asarray(got.status.cpu() if hasattr(got.status, 'cpu') else got.status)
    bad = 0; first = None
    for i in range(len(ln)):
        g = got.stream(i)
        if g != want.stream(i) or int(st[i]) != int(want.status[i]):
            bad += 1
            if first is None:
                w = want.stream(i)
                k = next((j for j in range(min(len(g), len(w))) if g[j] != w[j]), min(len(g), len(w)))
                first = (i, len(g), len(w), k, int(st[i]), int(want.status[i]))
    print(f"{name:40s} n={len(ln):6d} bad={bad} first={first} {time.time()-t0:.1f}s", flush=True)
    return bad
tot = 0
for ext in (True, False):
    rows = wl.synth_text(512, 4096); off, ln = wl.csr_for_fixed(512, 4096)
    tot += run(f"synth ext={ext}", rows.reshape(-1), off, ln, window=10, literal=8, extended=ext)
    for name in ('prose', 'python'):
        blob = wl.real_text(name)
        flat, off, ln = wl.split_fixed(blob[:(1 << 20) + 777], 4096)
        tot += run(f"{name} ext={ext}", flat, off, ln, window=10, literal=8, extended=ext)
    # ragged lengths incl. tiny
    rng = np.random.default_rng(5)
    lens = np.concatenate([np.arange(0, 40), rng.integers(1, 9000, 200)]).astype(np.uint32)
    blob = wl.real_text('prose')
    off = np.zeros(len(lens), np.uint64); off[1:] = np.cumsum(lens[:-1])
    flat = np.frombuffer(blob[:int(lens.sum())], dtype=np.uint8)
    for w in (8, 9, 10):
        tot += run(f"ragged w={w} ext={ext}", flat, off, lens, window=w, literal=8, extended=ext)
    # runs / zeros / periodic
    specials = [bytes(5000), b'ab' * 3000, b'abc' * 2000, bytes([7]) * 300 + b'xyz' + bytes([7]) * 3000, (b'x' * 20 + b'hello world ') * 300,
                bytes(rng.integers(0, 4, 6000, dtype=np.uint8)), bytes(rng.integers(0, 256, 6000, dtype=np.uint8))]
    lens = np.array([len(x) for x in specials], np.uint32); off = np.zeros(len(lens), np.uint64); off[1:] = np.cumsum(lens[:-1])
    flat = np.frombuffer(b''.join(specials), dtype=np.uint8)
    for w in (8, 10):
        tot += run(f"specials w={w} ext={ext}", flat, off, lens, window=w, literal=8, extended=ext)
    one = np.frombuffer(wl.real_text('prose')[:300000], dtype=np.uint8)
    tot += run(f"one 300k stream ext={ext}", one, np.zeros(1, np.uint64), np.array([300000], np.uint32), window=10, literal=8, extended=ext)
print("TOTAL BAD", tot)
