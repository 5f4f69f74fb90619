"""Randomised streams of both formats (EXT_P = share of extended ones, default 0.5) through the long-stream decoder (tamp_decompress_long_kernel.hpp; TAMP_AMD_LONGDEC_MIN lowered so that
streams of a few hundred bytes take it too) against the oracle's decoder: bytes, status, consumed count -- windows 8..15,
literal bits 5..8, custom dictionaries, FLUSH tokens, truncated and corrupted streams, output room from too small to ample.
usage: python tools/fuzz_long_decode_gpu.py [seconds]   (GPU box)"""
import os, random, sys, time
os.environ.setdefault('TAMP_AMD_LONGDEC_MIN', '64')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle
oracle = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(os.environ.get('FUZZ_SEED', os.environ.get('SEED', '7'))))
def plain(n):
    k = rng.randrange(7)
    if k == 0: return bytes(wl.synth_text(1, max(n, 1), first_index=rng.randrange(1 << 20))[0][:n])
    if k == 1: return bytes(wl.lcg_runs(1, max(n, 1), first_index=rng.randrange(1 << 20))[0][:n])
    if k == 2: return bytes(rng.choice(b"ab") for _ in range(n))
    if k == 3: return bytes(n)
    if k == 4:
        unit = bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 2000)))
        return (unit * (n // len(unit) + 1))[:n]
    if k == 5: return wl.real_text(rng.choice(['prose', 'python', 'markup']))[rng.randrange(2_000_000):][:n]
    return bytes(rng.randrange(256) for _ in range(n))
t0 = time.time(); streams = taken = 0
while time.time() - t0 < budget:
    w = rng.randrange(8, 16); lit = rng.choice([5, 6, 7, 8, 8, 8])
    dic = None
    if rng.random() < 0.25:
        dic = plain(1 << w)
        if lit < 8: dic = bytes(b & ((1 << lit) - 1) for b in dic)
    x = plain(rng.choice([200, 3000, 20_000, 200_000, rng.randrange(1, 600_000)]))
    if lit < 8: x = bytes(b & ((1 << lit) - 1) for b in x)
    ops, pos = [], 0
    while pos < len(x):
        k = rng.randrange(1, 150_000)
        ops.append(("write", x[pos:pos + k])); pos += k
        if rng.random() < 0.3: ops.append(("flush", rng.random() < 0.7))
    ops.append(("close",))
    st, blob = oracle.stream_script(ops, window=w, literal=lit, extended=rng.random() < float(os.environ.get('EXT_P', '0.5')), dictionary=dic)
    assert st == 0
    u = rng.random()
    if u < 0.2 and len(blob) > 3: blob = blob[:rng.randrange(1, len(blob))]
    elif u < 0.35 and len(blob) > 3:
        b = bytearray(blob); b[rng.randrange(1, len(b))] ^= 1 << rng.randrange(8); blob = bytes(b)
    cap = rng.choice([len(x) + 64, len(x) + 64, len(x), max(len(x) - 1, 0), rng.randrange(0, len(x) + 2), 2 * len(x) + 1000])
    wb = rng.choice([15, w, w, min(15, w + 1), max(8, w - 1)])
    want = oracle.decompress(blob, cap=cap, dictionary=dic, max_window_bits=wb)
    r = tamp_amd.decompress_batch([blob], out_cap=cap, dictionary=dic, max_window_bits=wb)
    got = (int(r.status[0]), bytes(r.stream(0)), int(r.in_consumed[0]) if r.in_consumed is not None else want[2])
    if got != want:
        print("MISMATCH", streams, "w", w, "lit", lit, "dict", dic is not None, "len", len(x), "blob", len(blob), "cap", cap, "wb", wb,
              (got[0], len(got[1]), got[2]), (want[0], len(want[1]), want[2]))
        np.save('gpurun_out/long_decode_fail.npy', np.frombuffer(blob, dtype=np.uint8)); sys.exit(1)
    streams += 1
print(f"long-stream decode fuzz ok: {streams} streams, {time.time()-t0:.0f} s")
