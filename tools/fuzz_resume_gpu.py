"""Randomised differential run of the resumable decoder (tamp_batch_decompress_resume through tamp_amd.DecoderBatch)
against the oracle's decoder object: random streams (flush tokens, dictionary resets, custom dictionaries, corrupted
bytes), random chunking of input and output room, a few hundred objects per launch.  usage: fuzz_resume_gpu.py SECONDS"""
import os, random, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import _lib, workloads as wl
from oracle.checker import Oracle

oracle = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(os.environ.get('SEED', '1'))
rng = random.Random(seed)
t0 = time.time()
rounds = objects = calls_total = 0


def rand_plain(n):
    k = rng.randrange(6)
    if k == 0:
        return bytes(wl.synth_text(1, max(n, 1), first_index=rng.randrange(1 << 20))[0][:n])
    if k == 1:
        return bytes(wl.lcg_runs(1, max(n, 1), first_index=rng.randrange(1 << 20))[0][:n])
    if k == 2:
        return bytes(rng.choice(b"ab") for _ in range(n))
    if k == 3:
        return bytes(n)
    if k == 4:
        unit = bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 40)))
        return (unit * (n // len(unit) + 1))[:n]
    return bytes(rng.randrange(256) for _ in range(n))


while time.time() - t0 < budget:
    w = rng.randrange(8, 16)
    wb = rng.choice([15, w, w, min(15, w + 1)])
    lit = rng.choice([5, 6, 7, 8, 8, 8])
    use_dict = rng.random() < 0.25
    dic = rand_plain(1 << w) if use_dict else None
    if dic is not None and lit < 8:
        dic = bytes(b & ((1 << lit) - 1) for b in dic)
    conf_given = rng.random() < 0.3
    ext_all, dr_all = rng.random() < 0.7, rng.random() < 0.3
    jobs = []
    for j in range(rng.choice([1, 7, 64, 200])):
        ext = ext_all if conf_given else rng.random() < 0.7
        dr = dr_all if conf_given else rng.random() < 0.3
        x = rand_plain(rng.choice([0, 1, 40, 700, 3000, rng.randrange(1, 6000)]))
        if lit < 8:
            x = bytes(b & ((1 << lit) - 1) for b in x)
        ops, pos = [], 0
        while pos < len(x):
            k = rng.randrange(1, 1200)
            ops.append(("write", x[pos:pos + k])); pos += k
            u = rng.random()
            if u < 0.15: ops.append(("flush", rng.random() < 0.7))
            elif u < 0.22 and dr and dic is None: ops.append(("reset",))
        ops.append(("close",))
        st, blob = oracle.stream_script(ops, window=w, literal=lit, extended=ext, dictionary=dic, dictionary_reset=dr)
        assert st == 0
        if rng.random() < 0.2 and len(blob) > 3:
            b = bytearray(blob); b[rng.randrange(1, len(b))] ^= 1 << rng.randrange(8); blob = bytes(b)
        if conf_given:
            blob = blob[1 + (blob[0] & 1):]
        script = [(rng.choice([0, 1, 2, 3, 5, 9, 40, 1000, 100000]), rng.choice([0, 1, 2, 3, 7, 20, 64, 300, 5000]))
                  for _ in range(rng.randrange(1, 60))]
        script.append((1 << 20, 1 << 15))
        jobs.append((blob, script))
    conf = (w, lit, dic is not None, ext_all, dr_all) if conf_given else None
    tconf = _lib.TampAmdConf(window=w, literal=lit, use_custom_dictionary=int(dic is not None), extended=int(ext_all),
                             dictionary_reset=int(dr_all)) if conf_given else None
    want_all = [oracle.decode_script(b, s, conf=conf, window_bits=wb, dictionary=dic) for b, s in jobs]
    try:
        batch = tamp_amd.DecoderBatch(len(jobs), window_bits=wb, conf=tconf, dictionary=dic)
    except ValueError:
        assert all(r0 != 0 for r0, _ in want_all), (w, wb, lit)
        continue
    assert all(r0 == 0 for r0, _ in want_all)
    pos = [0] * len(jobs)
    for step in range(max(len(s) for _, s in jobs)):
        chunks, caps = [], []
        for i, (blob, script) in enumerate(jobs):
            take, cap = script[step] if step < len(script) else (0, 0)
            chunks.append(blob[pos[i]:pos[i] + take]); caps.append(cap)
        status, outs, consumed = batch.step(chunks, np.array(caps, dtype=np.uint32))
        for i, (blob, script) in enumerate(jobs):
            if step < len(script):
                got = (int(status[i]), outs[i], int(consumed[i]))
                want = want_all[i][1][step]
                if got != want:
                    print("MISMATCH seed", seed, "round", rounds, "obj", i, "step", step, script[step], "w", w, "wb", wb, "lit", lit,
                          "dict", dic is not None, "conf", conf, (got[0], len(got[1]), got[2]), (want[0], len(want[1]), want[2]))
                    sys.exit(1)
                pos[i] += got[2]
                calls_total += 1
    rounds += 1; objects += len(jobs)
print(f"resume fuzz ok: {rounds} rounds, {objects} objects, {calls_total} calls, {time.time()-t0:.0f} s")
