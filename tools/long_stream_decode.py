"""One long stream there and back through the Python surface: block-mode compress (v1), then the decoder a single stream gets (wave per
stream).  Dev tool (GPU box).   usage: python tools/long_stream_decode.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
blob = wl.real_text('prose')
data = (blob * (32_000_000 // len(blob) + 1))[:32_000_000]
t0 = time.time(); c = tamp_amd.compress(data, extended=False); t1 = time.time()
print(f"compress 32 MB v1: {t1-t0:.3f} s, {len(c)} B")
t0 = time.time(); d = tamp_amd.decompress(c); t1 = time.time()
print(f"decompress: {t1-t0:.3f} s = {len(d)/(t1-t0)/1e6:.1f} MB/s, equal {d == data}")
