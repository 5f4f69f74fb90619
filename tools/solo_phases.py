"""Where the time of ONE stream alone on the device goes: the slowest 4 KiB chunk of each frozen corpus, timed alone with every
repeatable section of the -DTAMP_PROF build run twice (TAMP_AMD_DBG bits of tools/phase_valu5.sh): the difference is that
section's share of the stream's LATENCY (not of its issue slots).  Dev tool (GPU box).   usage: python tools/solo_phases.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tamp_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get('PROF_LIB', 'libtamp_amd_prof.so'))
import tamp_amd
from tamp_amd import workloads as wl
dev = torch.device('cuda:0'); L = 4096
SECT = [(0x10000, 'load'), (0x20000, 'index'), (0x100, 'bucket loop'), (0x200, 'wrap zone'), (0x2000000, 'second pass'),
        (0x4000000, 'settled tokens'), (0x40000, 'jump tables'), (0x8000000, 'walk'), (0x18000000, 'walk w/o listing'), (0x80000, 'emit')]
off1 = torch.zeros(1, dtype=torch.int64, device=dev); len1 = torch.full((1,), L, dtype=torch.int32, device=dev)
def solo(d, reps=3):
    return min(float(tamp_amd.compress_batch(d, off1, len1, max_in_len=L, timing=True).kernel_ms) for _ in range(reps)) * 1e3
for name in ('prose', 'markup', 'python'):
    flat = np.frombuffer(wl.real_text(name, frozen_only=True), dtype=np.uint8)
    k = flat.size // L
    data = torch.from_numpy(flat[:k * L].copy()).to(dev)
    os.environ['TAMP_AMD_DBG'] = '0'
    lat = np.array([solo(data[i * L:(i + 1) * L], 1) for i in range(k)])
    order = np.argsort(-lat)
    for which, i in (('slowest', int(order[0])), ('median', int(order[k // 2]))):
        d = data[i * L:(i + 1) * L]
        base = solo(d)
        r = tamp_amd.compress_batch(d, off1, len1, max_in_len=L)
        parts = []
        for bit, label in SECT:
            os.environ['TAMP_AMD_DBG'] = str(bit)
            parts.append((label, solo(d) - base))
        os.environ['TAMP_AMD_DBG'] = '0'
        print(f"{name:7s} {which} chunk {i}: {base:.0f} us alone, {int(r.out_len[0])} B out | " + "  ".join(f"{l} {v:+.0f}" for l, v in parts), flush=True)
