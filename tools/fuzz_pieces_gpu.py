"""Randomised write / flush scripts on tamp_amd.Compressor with every write handed over as a piece (PIECE_MIN = 1), against a
live reference object (oracle/_ref): the stream must be the reference's at every flush point and at the end, the running
byte count at most eight bytes ahead of it in between (the reference's object holds back the bits of its last token, <= 4 bytes,
and in the extended format a run / extended match it has not closed yet, <= 35 bits more; 5 was seen once in 5,000 scripts).
usage: python tools/fuzz_pieces_gpu.py [seconds]   (GPU box)"""
import io, os, random, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import tamp_amd
from tamp_amd import workloads as wl
from oracle.checker import Ref

ref = Ref()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(os.environ.get('FUZZ_SEED', '9')))
tamp_amd.Compressor.PIECE_MIN = 1
# PIECES_FOCUS=1: only the class of the one mismatch seen in round 6 (70,000 bytes, window 2^10, v1 format, no dictionary reset)
FOCUS = os.environ.get('PIECES_FOCUS', '') not in ('', '0')
def long_repeats(L, k):
    r = np.random.default_rng(k)
    buf = bytearray(r.integers(97, 97 + int(r.integers(2, 20)), 64, dtype=np.uint8).tobytes())
    while len(buf) < L:
        if r.random() < 0.6 and len(buf) > 30:
            d = int(r.integers(1, min(len(buf), 1200) + 1)); m = int(r.integers(14, 300))
            for _ in range(m): buf.append(buf[-d])
        elif r.random() < 0.3:
            buf += bytes([int(r.integers(97, 110))]) * int(r.integers(2, 400))
        else:
            buf += r.integers(97, 123, int(r.integers(1, 40)), dtype=np.uint8).tobytes()
    return bytes(buf[:L])
def source(k):
    kind = k % 6
    L = rng.choice([200, 3000, 20000, 70000])
    if FOCUS: L = 70000
    if kind == 0: return bytes(wl.synth_text(1, L, first_index=k)[0])
    if kind == 1: return bytes(wl.lcg_runs(1, L, first_index=k)[0])
    if kind == 2: return bytes(wl.stress(1, L, first_index=k)[0])
    if kind == 3: return long_repeats(L, k)
    if kind == 4: return wl.real_text('python')[(k * 7919) % 2_000_000:][:L]
    return wl.real_text('prose')[(k * 104729) % 2_000_000:][:L]
t0 = time.time(); scripts = calls = 0
while time.time() - t0 < budget:
    src = source(scripts)
    window = rng.choice([8, 9, 10, 10, 10, 11, 12])
    literal = 8
    ext = rng.random() < 0.75
    dreset = rng.random() < 0.2
    if FOCUS: window, ext, dreset = 10, False, False
    ops, pos = [], 0
    while pos < len(src):
        k = rng.choice([1, 2, 7, 15, 16, 17, 31, 100, 1000, 5000, 30000])
        ops.append(("write", src[pos:pos + k])); pos += k
        if rng.random() < 0.1: ops.append(("flush", rng.random() < 0.5))
        if dreset and rng.random() < 0.03: ops.append(("reset",))
    ops.append(("close",))
    want_counts = []
    rc, want = ref.stream_script(ops, window=window, literal=literal, extended=ext, dictionary_reset=dreset, counts=want_counts)
    assert rc == 0, rc
    f = io.BytesIO()
    c = tamp_amd.Compressor(f, window=window, literal=literal, extended=ext, dictionary_reset=dreset)
    got_counts = []
    for op in ops:
        if op[0] == "write": got_counts.append(c.write(op[1]))
        elif op[0] == "flush": got_counts.append(c.flush(op[1]))
        elif op[0] == "reset": got_counts.append(c.reset_dictionary())
        else: got_counts.append(c.close())
        calls += 1
    if f.getvalue() != want:
        got = f.getvalue()
        first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        print("STREAM MISMATCH", scripts, window, ext, dreset, "sizes", len(got), len(want), "first differing byte", first,
              "bytes out before each op", list(np.cumsum(got_counts))[:60],
              [(o[0], len(o[1]) if o[0] == 'write' else o[1:]) for o in ops][:60], flush=True)
        again = 0  # the same script three more times: a wrong answer that comes back is an algorithm bug, one that does not is a race
        for _ in range(3):
            f2 = io.BytesIO()
            c2 = tamp_amd.Compressor(f2, window=window, literal=literal, extended=ext, dictionary_reset=dreset)
            for op in ops:
                if op[0] == "write": c2.write(op[1])
                elif op[0] == "flush": c2.flush(op[1])
                elif op[0] == "reset": c2.reset_dictionary()
                else: c2.close()
            again += f2.getvalue() != want
        print("  replayed 3 times:", again, "wrong", flush=True)
        try:
            import pickle
            os.makedirs("gpurun_out", exist_ok=True)
            pickle.dump(dict(ops=ops, window=window, ext=ext, dreset=dreset, got=got, want=want), open("gpurun_out/pieces_fail.pkl", "wb"))
        except Exception as e:  # noqa: BLE001
            print("  (not saved:", e, ")")
        sys.exit(1)
    gc, wc = np.cumsum(got_counts), np.cumsum(want_counts)
    for i, op in enumerate(ops):
        d = int(gc[i] - wc[i])
        if (op[0] == "write" and not 0 <= d <= 8) or (op[0] != "write" and d != 0):
            print("COUNT MISMATCH", scripts, i, op[0], d); sys.exit(1)
    scripts += 1
print(f"fuzz_pieces: {scripts} scripts, {calls} calls, all equal to the reference object ({time.time()-t0:.0f} s)")
