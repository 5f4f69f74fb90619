"""Randomised differential run on the GPU box: compress (all modes) and decompress (both decoders) against the oracle.
   usage: python tools/fuzz_gpu.py [seconds]   -- prints a summary line; exits 1 on the first mismatch."""
import os, random, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle

o = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(os.environ.get('FUZZ_SEED', '1')))
t0 = time.time()
rounds = streams = 0
gens = [lambda n, L, k: wl.synth_text(n, L, first_index=k), lambda n, L, k: wl.lcg_runs(n, L, first_index=k),
        lambda n, L, k: wl.stress(n, L, first_index=k), lambda n, L, k: wl.telemetry(n, L, first_index=k),
        lambda n, L, k: np.random.default_rng(k).integers(0, 256, (n, L), dtype=np.uint8),
        lambda n, L, k: (np.random.default_rng(k).integers(0, 4, (n, L), dtype=np.uint8) * 37 + 65).astype(np.uint8),
        lambda n, L, k: np.repeat(np.random.default_rng(k).integers(0, 256, (n, (L + 6) // 7), dtype=np.uint8), 7, axis=1)[:, :L].copy()]
def long_repeats(n, L, k):
    # text-like bytes with long back-references (20-300 bytes copied from up to 1200 bytes back): long extended matches
    r = np.random.default_rng(k)
    rows = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        buf = bytearray(r.integers(97, 97 + int(r.integers(2, 20)), min(L, 64), dtype=np.uint8).tobytes())
        while len(buf) < L:
            if r.random() < 0.6 and len(buf) > 30:
                d = int(r.integers(1, min(len(buf), 1200) + 1)); m = int(r.integers(14, 300))
                for _ in range(m): buf.append(buf[-d])
            else:
                buf += r.integers(97, 123, int(r.integers(1, 12)), dtype=np.uint8).tobytes()
        rows[i] = np.frombuffer(bytes(buf[:L]), dtype=np.uint8)
    return rows
gens.append(long_repeats)
gens.append(long_repeats)
while time.time() - t0 < budget:
    n = rng.choice([1, 3, 64, 200, 700])
    L = rng.choice([1, 2, 17, 100, 333, 1024, 3000, 4096, 9000])
    window = rng.choice([8, 9, 10, 10, 10, 11, 12, 15])
    literal = rng.choice([8, 8, 8, 7])
    ext = rng.random() < 0.6
    lazy = rng.random() < 0.3
    rows = rng.choice(gens)(n, L, rng.randrange(1 << 20))
    if literal == 7:
        rows = rows & 0x7F
    d = None
    if rng.random() < 0.25:
        d = np.random.default_rng(rng.randrange(1 << 20)).integers(0, 128, 1 << window, dtype=np.uint8).tobytes()
    kw = dict(window=window, literal=literal, extended=ext, lazy_matching=lazy, dictionary=d)
    flat = np.ascontiguousarray(rows).reshape(-1)
    off, ln = wl.csr_for_fixed(n, L)
    got = tamp_amd.compress_batch(flat, off, ln, max_in_len=L, run_aware=rng.choice([None, False, True, True]), **kw)
    want = o.compress_batch(flat, off, ln, threads=16, **{('lazy' if k == 'lazy_matching' else k): v for k, v in kw.items()})
    comp = []
    for i in range(n):
        a, b = got.stream(i), want.stream(i)
        if int(got.status[i]) != int(want.status[i]) or a != b:
            print('COMPRESS MISMATCH', kw if d is None else {**kw, 'dictionary': 'custom'}, n, L, i, int(got.status[i]), int(want.status[i]), len(a), len(b))
            outdir = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out')
            os.makedirs(outdir, exist_ok=True)
            np.save(os.path.join(outdir, 'fuzz_fail_input.npy'), np.asarray(rows[i]))
            open(os.path.join(outdir, 'fuzz_fail_got.bin'), 'wb').write(a)
            open(os.path.join(outdir, 'fuzz_fail_want.bin'), 'wb').write(b)
            if d is not None: open(os.path.join(outdir, 'fuzz_fail_dict.bin'), 'wb').write(d)
            sys.exit(1)
        comp.append(b)
    cap = rng.choice([L + 8, L, max(1, L // 2), L + 300])
    trunc = rng.random() < 0.3
    if trunc:
        comp = [c[: rng.randrange(0, len(c) + 1)] for c in comp]
    for mode in ('wave', 'lane', 'global', 'split'):
        os.environ['TAMP_AMD_DECODER'] = mode
        res = tamp_amd.decompress_batch(comp, out_cap=cap, dictionary=d, max_window_bits=rng.choice([window, 15]))
        for i in range(n):
            st, out, cons = o.decompress(comp[i], dictionary=d, cap=cap, max_window_bits=15)
            if (int(res.status[i]), res.stream(i), int(res.in_consumed[i])) != (st, out, cons):
                print('DECOMPRESS MISMATCH', mode, window, literal, ext, n, L, i, cap, trunc, int(res.status[i]), st, int(res.out_len[i]), len(out), int(res.in_consumed[i]), cons)
                sys.exit(1)
    del os.environ['TAMP_AMD_DECODER']
    rounds += 1
    streams += n
print(f"fuzz ok: {rounds} rounds, {streams} streams, {time.time() - t0:.0f} s")
