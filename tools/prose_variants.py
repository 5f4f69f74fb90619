"""What makes real text slow in the extended format?  Prose corpus with its runs clipped to 2 / 8 / unlimited bytes.  Dev tool."""
import sys, os, glob, re
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
buf = bytearray()
for pat in ['/opt/skills/guides/*.md', os.path.join(root, '*.md'), '/usr/share/common-licenses/*', '/usr/share/doc/*/copyright']:
    for f in sorted(glob.glob(pat)):
        try: buf += open(f, 'rb').read()
        except Exception: pass
md = bytes(buf)
def clip(b, k):
    return re.sub(rb'(.)\1{%d,}' % k, lambda m: m.group(1) * k, b, flags=re.S)
def t(blob, N=32768, **kw):
    n = len(blob) // 4096
    rows = np.frombuffer(blob[:n * 4096], dtype=np.uint8).reshape(n, 4096)
    rows = np.tile(rows, ((N + n - 1) // n, 1))[:N]
    off, ln = wl.csr_for_fixed(N, 4096)
    data = torch.from_numpy(rows.reshape(-1).copy()).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    ms = []
    for it in range(3):
        r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=4096, timing=True, **kw); ms.append(r.kernel_ms)
    ol = r.out_len.cpu().numpy()
    return min(ms), N * 4096 / min(ms) / 1e6, ol.sum() / (N * 4096)
for name, blob in (("original", md), ("runs clipped to 8", clip(md, 8)), ("runs clipped to 2", clip(md, 2))):
    for ext in (True, False):
        ms, gbs, ratio = t(blob, extended=ext)
        print(f"{name:20s} ext={int(ext)} {ms:7.2f} ms {gbs:6.2f} GB/s ratio {ratio:.3f}", flush=True)
ms, gbs, ratio = t(bytes(wl.synth_text(4096, 4096).reshape(-1)))
print(f"{'synthetic':20s} ext=1 {ms:7.2f} ms {gbs:6.2f} GB/s ratio {ratio:.3f}")
