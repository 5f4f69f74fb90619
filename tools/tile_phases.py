"""Per-phase cycles of the tile kernel (-DTAMP_TILE_DBG build, libtamp_amd_tdbg.so).  Dev tool (GPU box).
   usage: WL=synth|prose|python EXT=1 python tools/tile_phases.py [n]"""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['TAMP_AMD_LIB'] = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tamp_amd', 'libtamp_amd_tdbg.so')
os.environ.setdefault('TAMP_AMD_ENCODER', 'tile')
import numpy as np, torch
import tamp_amd
from tamp_amd import _lib, workloads as wl
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.tamp_amd_prof_read(buf)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device('cuda:0')
for WL in os.environ.get('WL', 'synth,prose,python').split(','):
    rows = wl.synth_text(n, 4096) if WL == 'synth' else wl.tile_rows(wl.real_text(WL), n, 4096)
    off, ln = wl.csr_for_fixed(n, 4096)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    for ext in (1, 0):
        for it in range(2):
            r = tamp_amd.compress_batch(data, off_t, len_t, extended=bool(ext), max_in_len=4096, timing=True)
            torch.cuda.synchronize()
            lib.tamp_amd_prof_read(buf)
        v = np.array(list(buf), dtype=np.float64) / n
        tot = v[:7].sum()
        print(f"{WL:7s} ext={ext} {r.kernel_ms:6.2f} ms  builds/stream {v[8]:.1f} queries {v[9]:.0f} rounds {v[10]:.1f} | cycles/stream: fill {v[0]:.0f} index {v[1]:.0f} sort {v[2]:.0f} match {v[3]:.0f} jump {v[4]:.0f} walk {v[5]:.0f} emit {v[6]:.0f} total {tot:.0f}", flush=True)
