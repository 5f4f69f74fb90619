"""Expensive-first ordering on DISTINCT streams (the frozen corpora hold 2,304 distinct chunks, so the 24,414-stream stand-in
repeats each ~10 times, and any sort groups the copies): 20,000 distinct 4 KiB chunks of Python source from the image's
site-packages + 4,000 distinct chunks of prose-like text (licences, docs), in file order; the whole batch and an eighth of it,
ordering on / off.  Dev tool (GPU box)."""
import glob, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0'); L = 4096
def gather(patterns, nbytes):
    buf = bytearray()
    for pat in patterns:
        for f in sorted(glob.glob(pat, recursive=True)):
            if not os.path.isfile(f) or os.path.islink(f): continue
            try: buf += open(f, 'rb').read()
            except OSError: pass
            if len(buf) >= nbytes: return bytes(buf[:nbytes])
    return bytes(buf)
py = gather(['/usr/local/lib/python3.10/dist-packages/torch/**/*.py', '/usr/local/lib/python3.10/dist-packages/transformers/**/*.py', '/usr/lib/python3/dist-packages/**/*.py'], 20000 * L)
tx = gather(['/opt/rocm/share/html/**/*.html', '/usr/share/common-licenses/*'], 4000 * L)  # (regular files only: a recursive glob over /usr/share/doc met something that never returned)
blob = np.frombuffer(py + tx, dtype=np.uint8)
n = blob.size // L
rows = blob[: n * L].reshape(n, L)
rows = rows[np.random.default_rng(3).permutation(n)]   # (a batch mixes its sources)
print(f"{n} distinct chunks ({len(py) // L} Python, {len(tx) // L} text)")
def t(r, mode):
    os.environ['TAMP_AMD_LPT'] = mode
    k = len(r); off, ln = wl.csr_for_fixed(k, L)
    d = torch.from_numpy(np.ascontiguousarray(r).reshape(-1)).to(dev); o = torch.from_numpy(off.astype(np.int64)).to(dev); l = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms = [float(tamp_amd.compress_batch(d, o, l, max_in_len=L, timing=True).kernel_ms) for _ in range(6)]
    return float(np.median(ms[1:]))
for k in (n, n // 2, n // 4, n // 8):
    a, b = t(rows[:k], '0'), t(rows[:k], '1')
    print(f"  {k:6d} streams: caller's order {a:.3f} ms, expensive first {b:.3f} ms ({100 * (b / a - 1):+.1f} %)")
