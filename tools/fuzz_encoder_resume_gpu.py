"""Randomised differential run of the compressor objects below flush granularity (tamp_batch_compress_resume through
tamp_amd.EncoderBatch) against real reference objects (oracle/_ref, built in place from the reference's sources):
random op sequences -- compress / poll / sink / flush / compress_and_flush -- with random piece sizes and output
room, every call's (status, bytes, consumed) compared.  usage: fuzz_encoder_resume_gpu.py SECONDS   (needs an MI355X)"""
import os, random, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Ref

ref = Ref()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(os.environ.get('SEED', '1'))
rng = random.Random(seed)
t0 = time.time()
rounds = objects = calls_total = 0


def rand_plain(n):
    k = rng.randrange(7)
    if k == 0:
        return bytes(wl.synth_text(1, max(n, 1), first_index=rng.randrange(1 << 20))[0][:n])
    if k == 1:
        return bytes(wl.lcg_runs(1, max(n, 1), first_index=rng.randrange(1 << 20))[0][:n])
    if k == 2:
        return bytes(rng.choice(b"ab") for _ in range(n))
    if k == 3:
        return bytes([rng.randrange(256)]) * n
    if k == 4:
        unit = bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 40)))
        return (unit * (n // len(unit) + 1))[:n]
    if k == 5:
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(97, 100)]) * rng.randrange(1, 30)
        return bytes(out[:n])
    return bytes(rng.randrange(256) for _ in range(n))


while time.time() - t0 < budget:
    w = rng.choice([8, 8, 9, 10, 10, 11, 12, 15])
    lit = rng.choice([5, 7, 8, 8, 8])
    ext = rng.random() < 0.75
    lazy = rng.random() < 0.3
    dr = rng.random() < 0.3
    app = dr and rng.random() < 0.2
    dic = rand_plain(1 << w) if (rng.random() < 0.2 and not app) else None
    if dic is not None and lit < 8:
        dic = bytes(b & ((1 << lit) - 1) for b in dic)
    n = rng.choice([1, 5, 64, 150])
    kw = dict(window=w, literal=lit, extended=ext, dictionary=dic, dictionary_reset=dr, append=app, lazy_matching=lazy)
    batch = tamp_amd.EncoderBatch(n, **kw)
    scripts = [[] for _ in range(n)]
    got = [[] for _ in range(n)]
    for step in range(rng.randrange(1, 40)):
        kind = rng.choice(["compress", "compress", "compress", "poll", "sink", "flush", "caf"])
        small = rng.random() < 0.3
        caps = [rng.choice([0, 1, 2, 3, 5, 6, 8, 20]) if small else rng.choice([64, 300, 5000]) for _ in range(n)]
        datas = []
        for i in range(n):
            x = rand_plain(rng.choice([0, 1, 3, 15, 16, 17, 40, 300, rng.randrange(1, 1500)]))
            if lit < 8 and rng.random() < 0.97:
                x = bytes(b & ((1 << lit) - 1) for b in x)
            datas.append(x)
        tok = rng.random() < 0.6
        if kind == "compress":
            st, outs, cons = batch.compress(datas, caps)
            ops = [("compress", datas[i], caps[i]) for i in range(n)]
        elif kind == "poll":
            st, outs, cons = batch.poll(caps)
            ops = [("poll", caps[i]) for i in range(n)]
        elif kind == "sink":
            cons = batch.sink(datas)
            st, outs = [0] * n, [b""] * n
            ops = [("sink", datas[i]) for i in range(n)]
        elif kind == "flush":
            st, outs, cons = batch.flush(caps, tok)
            ops = [("flush", tok, caps[i]) for i in range(n)]
        else:
            st, outs, cons = batch.compress_and_flush(datas, caps, tok)
            ops = [("compress_and_flush", datas[i], tok, caps[i]) for i in range(n)]
        for i in range(n):
            scripts[i].append(ops[i])
            got[i].append((int(st[i]), outs[i], int(cons[i])))
    for i in range(n):
        r0, want = ref.encode_script(scripts[i], **kw)
        assert r0 == 0
        if got[i] != want:
            for k, (g, x) in enumerate(zip(got[i], want)):
                if g != x:
                    print("MISMATCH seed", seed, "round", rounds, "obj", i, "call", k, scripts[i][k][0],
                          [len(a) if isinstance(a, bytes) else a for a in scripts[i][k][1:]], kw | {"dictionary": dic is not None},
                          "got", (g[0], g[1].hex()[:40], g[2]), "want", (x[0], x[1].hex()[:40], x[2]))
                    break
            sys.exit(1)
        calls_total += len(want)
    rounds += 1
    objects += n
print(f"encoder resume fuzz ok: {rounds} rounds, {objects} objects, {calls_total} calls, {time.time()-t0:.0f} s")
