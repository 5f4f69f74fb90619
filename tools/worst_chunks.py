"""What makes the slowest 4 KiB chunks slow?  Finds the chunks of a frozen corpus with the largest solo latency and reads the
-DTAMP_PROF build's per-stream counters for each (epochs, positions matched, slow steps, searches, cycles per phase).
Dev tool (GPU box): python tools/worst_chunks.py python 5"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else 'python'
top = int(sys.argv[2]) if len(sys.argv) > 2 else 5
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0'); L = 4096
flat = np.frombuffer(wl.real_text(name, frozen_only=True), dtype=np.uint8)
k = flat.size // L
rows = flat[:k * L].reshape(k, L)
data = torch.from_numpy(rows.reshape(-1).copy()).to(dev)
off1 = torch.zeros(1, dtype=torch.int64, device=dev); len1 = torch.full((1,), L, dtype=torch.int32, device=dev)
lat = np.array([min(float(tamp_amd.compress_batch(data[i * L:(i + 1) * L], off1, len1, max_in_len=L, timing=True).kernel_ms) for _ in range(2)) for i in range(k)])
order = np.argsort(-lat)
print(f"{name}: solo latency us: mean {lat.mean()*1e3:.0f}, worst {[(int(i), round(float(lat[i])*1e3)) for i in order[:top]]}, median chunk {int(order[k//2])} {lat[order[k//2]]*1e3:.0f}")
sel = [int(i) for i in order[:top]] + [int(order[k // 2])]
import subprocess, json
if os.environ.get('CHILD') != '1':
    out = subprocess.run([sys.executable, __file__, name, str(top)], env=dict(os.environ, CHILD='1', SEL=json.dumps(sel), TAMP_AMD_LIB=os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_amd_prof.so')), capture_output=True, text=True)
    print(out.stdout[out.stdout.index('PROF'):] if 'PROF' in out.stdout else out.stdout + out.stderr[-2000:])
else:
    lib = _lib.load(); buf = (C.c_ulonglong * 16)()
    print('PROF build, 64 copies of each chunk, per-stream figures:')
    for i in json.loads(os.environ['SEL']):
        r64 = np.tile(rows[i:i + 1], (64, 1)); o, l = wl.csr_for_fixed(64, L)
        d = torch.from_numpy(r64.reshape(-1)).to(dev)
        lib.tamp_amd_prof_read(buf)
        tamp_amd.compress_batch(d, torch.from_numpy(o.astype(np.int64)).to(dev), torch.from_numpy(l.astype(np.int32)).to(dev), max_in_len=L)
        torch.cuda.synchronize(); lib.tamp_amd_prof_read(buf)
        v = np.array(list(buf), dtype=np.float64) / 64
        txt = bytes(rows[i][:4096])
        runs12 = sum(1 for j in range(0, 4084) if txt[j:j+12] == txt[j:j+1] * 12 and (j == 0 or txt[j-1] != txt[j]))
        print(f" chunk {i}: epochs {v[12]:.0f} positions matched {v[13]:.0f} slow steps {v[11]:.0f} searches {v[14]:.0f} | cycles load {v[0]:.0f} index {v[1]:.0f} match {v[2]:.0f} walk {v[3]:.0f} emit {v[4]:.0f} | runs of 12+: {runs12}, spaces {txt.count(b' ')}, newlines {txt.count(10)}")
