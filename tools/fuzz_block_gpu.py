"""Randomised differential run of BLOCK MODE (one long v1 stream over all workgroups) against the oracle: the
threshold is lowered to 1 KiB so that streams of a few blocks take it; random lengths around block boundaries, windows 2^8..2^14,
custom dictionaries, output room that ends inside the stream, both with pass 1's tables kept and with the re-match.
   usage: python tools/fuzz_block_gpu.py [seconds]   -- prints a summary line; exits 1 on the first mismatch."""
import os, random, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['TAMP_AMD_BLOCK_MIN'] = '1024'
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle

o = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(os.environ.get('FUZZ_SEED', '1')))
t0 = time.time()
rounds = 0
def gen(n, k):
    r = np.random.default_rng(k)
    kind = k % 6
    if kind == 0: return wl.synth_text(1, n, first_index=k)[0]
    if kind == 1: return wl.lcg_runs(1, n, first_index=k)[0]
    if kind == 2: return r.integers(0, 256, n, dtype=np.uint8)
    if kind == 3: return (r.integers(0, 4, n, dtype=np.uint8) * 37 + 65).astype(np.uint8)
    if kind == 4: return np.repeat(r.integers(0, 256, (n + 6) // 7, dtype=np.uint8), 7)[:n].copy()
    buf = bytearray(r.integers(97, 110, 64, dtype=np.uint8).tobytes())   # long back-references
    while len(buf) < n:
        if r.random() < 0.6:
            d = int(r.integers(1, min(len(buf), 1200) + 1)); m = int(r.integers(2, 200))
            for _ in range(m): buf.append(buf[-d])
        else:
            buf += r.integers(97, 123, int(r.integers(1, 12)), dtype=np.uint8).tobytes()
    return np.frombuffer(bytes(buf[:n]), dtype=np.uint8).copy()
while time.time() - t0 < budget:
    window = rng.choice([8, 9, 10, 10, 10, 11, 12, 13, 14])
    blk = 1024
    n = rng.choice([1024, 1025, 2047, 2048, 2049, 3000, 4096 + rng.randrange(-20, 20), rng.randrange(1024, 20000), rng.randrange(20000, 200000),
                    blk * rng.randrange(1, 40) + rng.choice([-1, 0, 1, 13, 14, 15, 16])])
    n = max(n, 1024)
    flat = np.ascontiguousarray(gen(n, rng.randrange(1 << 20)))
    d = None
    if rng.random() < 0.3:
        d = np.random.default_rng(rng.randrange(1 << 20)).integers(0, 256, 1 << window, dtype=np.uint8).tobytes()
    if rng.random() < 0.5: os.environ['TAMP_AMD_BLOCK_REMATCH'] = '1'
    else: os.environ.pop('TAMP_AMD_BLOCK_REMATCH', None)
    st, want = o.compress(flat.tobytes(), window=window, literal=8, extended=False, dictionary=d)
    cap = None
    if rng.random() < 0.25: cap = rng.randrange(1, len(want) + 3)
    got = tamp_amd.compress_batch([flat.tobytes()], window=window, literal=8, extended=False, dictionary=d, **({} if cap is None else {'out_cap': cap}))
    exp = want if cap is None else want[:cap]
    exp_status = 0 if (cap is None or cap >= len(want)) else 1
    if got.stream(0) != exp or int(got.status[0]) != exp_status:
        g = got.stream(0)
        k = next((i for i in range(min(len(g), len(exp))) if g[i] != exp[i]), -1)
        print('BLOCK MODE MISMATCH', dict(window=window, n=n, cap=cap, custom=d is not None, rematch=os.environ.get('TAMP_AMD_BLOCK_REMATCH')),
              'status', int(got.status[0]), exp_status, 'len', len(g), len(exp), 'first diff', k)
        np.save(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'fuzz_block_fail.npy'), flat)
        sys.exit(1)
    rounds += 1
print(f'block-mode fuzz ok: {rounds} streams, {int(time.time() - t0)} s')
