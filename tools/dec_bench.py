"""Decoder timing, both variants (TAMP_AMD_DECODER=wave|lane), default max_window_bits.  Dev tool."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
if os.environ.get('TAMP_VAR'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_var%s.so' % os.environ['TAMP_VAR'])
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0')
def run(name, rows, **kw):
    n, L = rows.shape
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, **kw)
    cap = torch.full((n,), L, dtype=torch.int32, device=dev)
    for mode in ('wave', 'lane', 'global', 'split', 'auto'):
        os.environ['TAMP_AMD_DECODER'] = mode
        if mode == 'auto': del os.environ['TAMP_AMD_DECODER']
        ms = []
        for it in range(4):
            d = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=cap, dictionary=kw.get('dictionary'), timing=True)
            ms.append(d.kernel_ms)
        ok = bool((d.out[:n*L].cpu().numpy() == rows.reshape(-1)).all())
        print(f"{name:26s} {mode}: {min(ms):7.3f} ms  {n*L/min(ms)/1e6:7.1f} GB/s out  ok={ok}", flush=True)
for nn in (16384,):
    run(f"text {nn}x4K w10 ext", wl.synth_text(nn, 4096))
run("text 65536x4K w10 ext", wl.synth_text(65536, 4096))
run("text 262144x4K w10 ext", wl.synth_text(262144, 4096))
run("text 65536x4K w10 v1", wl.synth_text(65536, 4096), extended=False)
run("telemetry 1Mx256 w8 l7", wl.telemetry(1<<20, 256), window=8, literal=7)
run("telemetry 64Kx256 w8 l7", wl.telemetry(1<<16, 256), window=8, literal=7)
run("text 16384x16K w10", wl.synth_text(16384, 16384))
run("text 65536x4K w8", wl.synth_text(65536, 4096), window=8)
run("text 16384x4K w12", wl.synth_text(16384, 4096), window=12)

# BASELINE config 4 shape: windows 8..12 mixed in one batch
def run_mixed(n, L):
    import numpy as np
    from oracle.checker import Oracle
    o = Oracle()
    rows = wl.synth_text(n, L)
    wsel = np.arange(n) % 5 + 8
    comp = [None] * n
    for w in range(8, 13):
        ids = np.nonzero(wsel == w)[0]
        res = o.compress_batch(np.ascontiguousarray(rows[ids]).reshape(-1), *wl.csr_for_fixed(len(ids), L), window=w, threads=32)
        for k, i in enumerate(ids): comp[i] = res.stream(k)
    flat, off, ln = tamp_amd.pack_streams(comp) if hasattr(tamp_amd, 'pack_streams') else (None, None, None)
    from tamp_amd.batch import pack_streams
    flat, off, ln = pack_streams(comp)
    d = torch.from_numpy(flat).to(dev); o_t = torch.from_numpy(off.astype(np.int64)).to(dev); l_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    cap = torch.full((n,), L, dtype=torch.int32, device=dev)
    for mode in ('wave', 'global', 'auto'):
        os.environ['TAMP_AMD_DECODER'] = mode
        if mode == 'auto': del os.environ['TAMP_AMD_DECODER']
        ms = []
        for it in range(3):
            r = tamp_amd.decompress_batch(d, o_t, l_t, out_cap=cap, timing=True); ms.append(r.kernel_ms)
        ok = bool((r.out[:n*L].cpu().numpy() == rows.reshape(-1)).all())
        print(f"config 4: {n}x{L} windows 8..12 {mode}: {min(ms):7.3f} ms {n*L/min(ms)/1e6:7.1f} GB/s out ok={ok}", flush=True)
run_mixed(262144, 4096)
