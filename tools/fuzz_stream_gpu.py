"""Randomised op scripts (write / flush / reset_dictionary / close) on tamp_amd.Compressor vs the oracle's segment
   restatement, on the GPU box.  usage: python tools/fuzz_stream_gpu.py [seconds]"""
import io, os, random, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Oracle

o = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(os.environ.get('FUZZ_SEED', '5')))
srcs = [bytes(wl.synth_text(1, 30000, first_index=321)[0]), bytes(wl.lcg_runs(1, 12000, first_index=3)[0]),
        bytes(wl.stress(3, 12000)[1]), bytes(wl.stress(3, 12000)[2]), b"\0" * 9000, bytes(wl.telemetry(40, 256).reshape(-1))]
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    src = rng.choice(srcs)
    dr = rng.random() < 0.5
    window = rng.choice([8, 9, 10, 12, 15])
    conf = dict(window=window, literal=8, extended=rng.random() < 0.7, dictionary_reset=dr,
                append=dr and rng.random() < 0.2, lazy_matching=rng.random() < 0.25)
    if not conf['append'] and rng.random() < 0.2:
        conf['dictionary'] = bytes(rng.randrange(256) for _ in range(1 << window))
    pos, ops = rng.randrange(0, 3000), []
    for _ in range(rng.randrange(1, 8)):
        x = rng.random()
        if x < 0.55:
            n = rng.choice([0, 1, 2, 3, 15, 16, 17, 31, 100, 700, 3000, 6000])
            ops.append(("write", src[pos: pos + n])); pos += n
        elif x < 0.85:
            ops.append(("flush", rng.random() < 0.7))
        elif dr:
            ops.append(("reset",))
    ops.append(("close",))
    st, want = o.stream_script(ops, **conf)
    assert st == 0
    f = io.BytesIO()
    c = tamp_amd.Compressor(f, **conf)
    for op in ops:
        if op[0] == "write": c.write(op[1])
        elif op[0] == "flush": c.flush(write_token=bool(op[1]))
        elif op[0] == "reset": c.reset_dictionary()
        else: c.close()
    if f.getvalue() != want:
        print("MISMATCH", {k: (v if k != 'dictionary' else 'custom') for k, v in conf.items()}, [(x[0], len(x[1]) if x[0] == 'write' else x[1:]) for x in ops])
        sys.exit(1)
    cases += 1
print(f"stream fuzz ok: {cases} scripts, {time.time() - t0:.0f} s")
