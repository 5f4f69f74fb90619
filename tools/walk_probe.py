"""How much of the kernel's time hangs on the serial walk: the -DTAMP_PROF build can skip the token listing (TAMP_AMD_DBG
0x200000), the extended-match search (0x400000) or the on-demand match of deferred positions (0x800000) -- the bytes are
wrong, the time is what an infinitely fast version of that piece would leave.  Dev tool (GPU box)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
    import numpy as np, torch
    from tamp_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtamp_amd_prof.so')
    import tamp_amd
    from tamp_amd import workloads as wl
    dev = torch.device('cuda:0'); N = 65536
    rng = np.random.default_rng(5)
    out = []
    for name in ('synth', 'prose', 'markup', 'python'):
        if name == 'synth':
            rows = wl.synth_text(N, 4096)
        else:
            blob = wl.real_text(name); n = len(blob) // 4096
            base = np.frombuffer(blob[:n * 4096], dtype=np.uint8).reshape(n, 4096)
            rows = np.ascontiguousarray(base[rng.permutation(np.arange(N) % n)])
        off, ln = wl.csr_for_fixed(N, 4096)
        data = torch.from_numpy(rows.reshape(-1)).to(dev); off_t = torch.from_numpy(off.astype(np.int64)).to(dev); len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
        ms = []
        for it in range(4):
            r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=4096, timing=True); ms.append(r.kernel_ms)
        out.append(f"{name} {min(ms[1:]):.3f}")
    print(f"dbg={int(os.environ.get('TAMP_AMD_DBG', '0')):#x}: " + "  ".join(out) + "  (ms, 65,536 x 4 KiB, extended, PROF build)", flush=True)
else:
    for d in (0, 0x200000, 0x400000, 0x800000, 0xE00000):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TAMP_AMD_DBG=str(d)))
