"""Per-phase cycle / counter read-out of the -DTAMP_PROF build (make -C tamp_amd/csrc prof).  Dev tool.
   usage (GPU box, repo root): WL=synth_text python tools/prof_phases.py 8192   (WL=glob:<patterns> for files)"""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get('PROF_LIB', 'libtamp_amd_prof.so'))
import tamp_amd
from tamp_amd import workloads as wl
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.tamp_amd_prof_read(buf)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
WL = os.environ.get('WL', 'synth_text')
if WL.startswith('glob:'):
    import glob
    blob = bytearray()
    for pat in WL[5:].split(','):
        for f in sorted(glob.glob(pat)):
            blob += open(f, 'rb').read()
    k = len(blob) // 4096
    rows = np.frombuffer(bytes(blob[:k*4096]), dtype=np.uint8).reshape(k, 4096)
    rows = np.tile(rows, ((n + k - 1) // k, 1))[:n].copy()
elif WL.startswith('corpus:'):
    rows = wl.tile_rows(wl.real_text(WL[7:]), n, 4096)
else:
    rows = getattr(wl, WL)(n, int(os.environ.get('SLEN', '4096')))
SLEN = rows.shape[1]
off, ln = wl.csr_for_fixed(n, SLEN)
CONF = dict(window=int(os.environ.get('WINDOW', '10')), literal=int(os.environ.get('LITERAL', '8')))
if os.environ.get('TELDICT'):
    CONF['dictionary'] = wl.telemetry_dictionary(bytes(tamp_amd.initialize_dictionary(256, literal=7)))
dev = torch.device('cuda:0')
data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
for ext in (1, 0):
    for it in range(2):
        r = tamp_amd.compress_batch(data, off_t, len_t, extended=bool(ext), max_in_len=SLEN, timing=True, **CONF)
        torch.cuda.synchronize()
        lib.tamp_amd_prof_read(buf)
    v = np.array(list(buf), dtype=np.float64) / n
    print(f"   epochs/stream={v[12]:.1f} positions matched/stream={v[13]:.0f}")
    print(f"   extended matches/stream: searched={v[14]:.1f} settled without a search={v[15]:.1f}")
    print(f"   walk detail: slow steps/stream={v[11]:.1f} cycles in ext-continuation steps={v[5]:.0f} other slow steps={v[9]:.0f}")
    print(f"   fine: setup={v[6]:.0f} loop={v[7]:.0f} special+epilog={v[8]:.0f} barrierwait={v[9]:.0f} iters(thread0)={v[10]:.0f}")
    print(f"ext={ext} kernel_ms={r.kernel_ms:.2f} cycles/stream: load+zero={v[0]:.0f} index={v[1]:.0f} match={v[2]:.0f} walk={v[3]:.0f} emit={v[4]:.0f}  (s_memtime ticks @100MHz?)")
