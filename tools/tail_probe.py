"""Is a real-text batch bound by its slowest streams?  Times the prose corpus at several batch sizes and finds the
   slowest 4 KiB chunks by timing 256-stream slices.  Dev tool."""
import sys, os, glob
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
def corpus(patterns):
    buf = bytearray()
    for pat in patterns:
        for f in sorted(glob.glob(pat)):
            try: buf += open(f, 'rb').read()
            except Exception: pass
    return bytes(buf)
def t(rows, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1).copy()).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms = []
    for it in range(3):
        r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, timing=True, **kw); ms.append(r.kernel_ms)
    return min(ms)
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
md = corpus(['/opt/skills/guides/*.md', os.path.join(root, '*.md'), '/usr/share/common-licenses/*', '/usr/share/doc/*/copyright'])
n = len(md) // 4096
rows = np.frombuffer(md[:n * 4096], dtype=np.uint8).reshape(n, 4096)
print("prose chunks:", n)
for N in (1536, 4096, 16384, 65536):
    big = np.tile(rows, ((N + n - 1) // n, 1))[:N]
    ms = t(big)
    print(f"N={N:6d} ext {ms:7.2f} ms {N*4096/ms/1e6:6.2f} GB/s   per 1536-slot round: {ms/(N/1536):.3f} ms")
# per-chunk cost: each chunk replicated 1536x (one full round of identical streams)
cost = []
for i in range(0, n, max(1, n // 48)):
    ms = t(np.tile(rows[i:i+1], (1536, 1)))
    cost.append((ms, i))
cost.sort(reverse=True)
print("slowest sampled chunks (ms per full round of copies):", [(round(m, 2), i) for m, i in cost[:8]])
print("median:", round(cost[len(cost)//2][0], 2), "fastest:", round(cost[-1][0], 2))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
open(os.path.join(root, "gpurun_out", "slow_chunk.bin"), "wb").write(rows[cost[0][1]].tobytes())
open(os.path.join(root, "gpurun_out", "slow_chunk2.bin"), "wb").write(rows[cost[1][1]].tobytes())
for m, i in cost[:3]:
    print("---- chunk", i, "ms", round(m, 2)); print(rows[i].tobytes()[:600].decode('latin1'))
