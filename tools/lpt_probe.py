"""Does the ORDER of the streams in a batch matter?  Real-text batches have a tail: a workgroup that picks an expensive stream
last keeps the device waiting.  The kernel maps workgroup i to table row i, so permuting the tables permutes the schedule.
usage: python tools/lpt_probe.py [n_streams]   (GPU box)"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24414
L = 4096
def distinct_python(n):
    """n DISTINCT 4 KiB chunks of Python source from the image's site-packages (torch, transformers: ~80 MiB), file order"""
    import glob
    buf = bytearray()
    for pat in ('/usr/local/lib/python3.10/dist-packages/torch/**/*.py', '/usr/local/lib/python3.10/dist-packages/transformers/**/*.py'):
        for f in sorted(glob.glob(pat, recursive=True)):
            try: buf += open(f, 'rb').read()
            except OSError: pass
            if len(buf) >= n * L: break
    assert len(buf) >= n * L, len(buf)
    return np.frombuffer(bytes(buf[: n * L]), dtype=np.uint8).reshape(n, L).copy()
for name in (os.environ.get('CORPORA', 'prose,python').split(',')):
    if name == 'distinct':
        rows = distinct_python(N)   # no shuffle: a file's chunks stay neighbours, as in a real batch
    else:
        rows = wl.tile_rows(wl.real_text(name), N, L)
        # different copies of a chunk should not sit next to each other only: shuffle once, deterministically
        rows = rows[np.random.default_rng(1).permutation(N)]
    flat = torch.from_numpy(rows.reshape(-1)).to(dev)
    eq = (rows[:, 1:] == rows[:, :-1])
    adj = eq.sum(1)                                     # bytes equal to their predecessor
    base_off = np.arange(N, dtype=np.int64) * L
    cap = tamp_amd.compress_bound(L, 8)
    def run(order, label):
        off_t = torch.from_numpy(base_off[order]).to(dev)
        len_t = torch.full((N,), L, dtype=torch.int32, device=dev)
        ms = []
        for _ in range(5):
            r = tamp_amd.compress_batch(flat, off_t, len_t, max_in_len=L, out_cap=cap, timing=True)
            ms.append(r.kernel_ms)
        print(f"{name:7s} n={N} {label:28s} {min(ms):7.3f} ms  {N * L / min(ms) / 1e6:6.2f} GB/s", flush=True)
        return r
    r = run(np.arange(N), "as given")
    size = r.out_len.cpu().numpy()
    run(np.argsort(-adj, kind='stable'), "most equal neighbours first")
    run(np.argsort(adj, kind='stable'), "fewest equal neighbours first")
    run(np.argsort(-size, kind='stable'), "largest output first")
    run(np.argsort(size, kind='stable'), "smallest output first")
