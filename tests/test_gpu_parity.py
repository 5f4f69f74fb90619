"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden vectors.

Bit-exact everywhere (integer/byte work): compressed bytes, decoded bytes, per-stream status codes.
"""
import hashlib
import random

import numpy as np
import pytest
from conftest import load_golden, unb64, workload_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tamp_amd
    from tamp_amd import _lib

    lib = _lib.load()  # raises if the native library is missing: no silent fallback
    assert lib.tamp_amd_device_count() >= 1, "no HIP device visible"
    return tamp_amd


def test_known_answer_compress(ta):
    ka = load_golden("known_answers.json")
    for c in ka["compress"]:
        got = ta.compress(unb64(c["input"]), dictionary=unb64(c["dictionary"]), **c["conf"])
        assert got.hex() == c["expected"], (c["name"], c["cite"])


def test_known_answer_decompress(ta):
    ka = load_golden("known_answers.json")
    for c in ka["decompress"]:
        res = ta.decompress_batch([bytes.fromhex(c["compressed"])], out_cap=4096, dictionary=unb64(c["dictionary"]))
        assert int(res.status[0]) == c["status"], (c["name"], c["cite"])
        assert res.stream(0) == unb64(c["expected"]), c["name"]


def test_python_surface_like_reference_tests(ta):
    # tests/test_compressor.py:66-105 (write + flush(write_token=False), context manager, byte counts)
    import io

    expected = bytes.fromhex("58b3041c8100030000")
    with io.BytesIO() as f:
        c = ta.Compressor(f, extended=False)
        n = c.write(b"foo foo foo")
        n += c.flush(write_token=False)
        assert f.getvalue() == expected and n == len(expected)
    with io.BytesIO() as f, ta.Compressor(f, extended=False) as c:  # tests/test_compressor.py:107-143
        c.write(b"f"), c.write(b"oo"), c.write(b" fo"), c.write(b"o foo")
        c.flush(write_token=False)
        assert f.getvalue() == expected
    assert ta.compress("foo foo foo", extended=False) == expected  # str input, :248-268
    assert ta.decompress(expected) == b"foo foo foo"
    with io.BytesIO(expected) as f:  # tests/test_decompressor.py:69-94
        d = ta.Decompressor(f)
        assert d.read(4) == b"foo " and d.read(2) == b"fo" and d.read(-1) == b"o foo"
    with pytest.raises(ta.ExcessBitsError):  # tests/test_compressor.py:238-246
        ta.compress(b"\xff", literal=7, extended=False)
    with pytest.raises(ValueError):
        ta.Compressor(io.BytesIO(), window=9, literal=7, dictionary=bytearray(256))
    with pytest.raises(ValueError):
        ta.Compressor(io.BytesIO(), literal=4)
    with pytest.raises(ValueError):  # tests/test_decompressor.py:115-122
        ta.Decompressor(io.BytesIO(bytes([0b00010100])))


def test_restricted_output(ta, oracle):
    ka = load_golden("known_answers.json")
    for c in ka["decompress"]:
        if c["status"] != 2:
            continue
        comp = bytes.fromhex(c["compressed"])
        full = unb64(c["expected"])
        caps = list(range(0, len(full) + 2))
        res = ta.decompress_batch([comp] * len(caps), out_cap=np.array(caps, dtype=np.uint32),
                                  dictionary=unb64(c["dictionary"]))
        for j, cap in enumerate(caps):
            want = oracle.decompress(comp, dictionary=unb64(c["dictionary"]), cap=cap)
            assert (int(res.status[j]), res.stream(j), int(res.in_consumed[j])) == want, (c["name"], cap)


def test_device_vectors(ta):
    vs = load_golden("device_vectors.json")
    res = ta.decompress_batch([unb64(v["data"]) for v in vs], out_cap=1 << 16)
    for j, v in enumerate(vs):
        assert (int(res.status[j]), res.stream(j), int(res.in_consumed[j])) == (v["status"], unb64(v["output"]), v["consumed"]), v["name"]


def test_generated_reference_outputs(ta):
    g = load_golden("generated.json")
    tel = unb64(g["telemetry_dictionary"])
    groups = {}
    rows_cache = {}
    for c in g["cases"]:
        wlname = c["workload"]
        if wlname not in rows_cache:
            need = 1 + max(k["index"] for k in g["cases"] if k["workload"] == wlname)
            rows_cache[wlname] = workload_rows(wlname)(need)
        key = (wlname, tuple(sorted(c["conf"].items())), c["dictionary"])
        groups.setdefault(key, []).append(c)
    n_checked = 0
    for (wlname, conf_items, dname), cases in groups.items():
        conf = dict(conf_items)
        datas = [rows_cache[wlname][c["index"]].tobytes() for c in cases]
        d = tel if dname == "telemetry" else None
        res = ta.compress_batch(datas, dictionary=d, **conf)
        for j, c in enumerate(cases):
            assert hashlib.sha256(datas[j]).hexdigest() == c["input_sha256"]
            assert int(res.status[j]) == c["status"], (wlname, c["index"], conf)
            assert res.stream(j) == unb64(c["compressed"]), (wlname, c["index"], conf)
            n_checked += 1
        ok = [j for j, c in enumerate(cases) if c["status"] == 0]
        if ok:
            back = ta.decompress_batch([unb64(cases[j]["compressed"]) for j in ok], out_cap=len(datas[0]) + 8, dictionary=d)
            for k, j in enumerate(ok):
                assert int(back.status[k]) == 2 and back.stream(k) == datas[j]
    assert n_checked > 250


def _rand_inputs(rng, wl, n):
    kind = rng.randrange(5)
    if kind == 0:
        return wl.synth_text(1, n, first_index=rng.randrange(1 << 20))[0].tobytes()
    if kind == 1:
        return bytes(rng.randrange(256) for _ in range(n))
    if kind == 2:
        return wl.lcg_runs(1, n, first_index=rng.randrange(1 << 20))[0].tobytes()
    if kind == 3:
        return wl.stress(1, n, first_index=rng.randrange(1 << 20))[0].tobytes()
    return bytes([rng.randrange(256)]) * n


def test_differential_vs_oracle(ta, oracle):
    """Random configurations, ragged batches (fuzz/fuzz_round_trip.c:14-90 pattern): GPU == oracle, byte for byte."""
    from tamp_amd import workloads as wl

    rng = random.Random(7)
    for it in range(90):
        w, lit = rng.randrange(8, 16), rng.randrange(5, 9)
        ext = rng.random() < 0.6
        lazy = rng.random() < 0.35
        d = None
        if rng.random() < 0.3:
            d = (_rand_inputs(rng, wl, 1 << w) + bytes(1 << w))[: 1 << w]
        datas = []
        for _ in range(rng.randrange(1, 24)):
            n = rng.choice([0, 1, 2, 3, 15, 16, 17, 33, 100, 256, 1000, 4096, rng.randrange(1, 9000)])
            x = _rand_inputs(rng, wl, n)
            if lit < 8 and rng.random() < 0.9:
                x = bytes(b & ((1 << lit) - 1) for b in x)
            datas.append(x)
        res = ta.compress_batch(datas, window=w, literal=lit, extended=ext, dictionary=d, lazy_matching=lazy)
        comps = []
        for j, x in enumerate(datas):
            st, want = oracle.compress(x, window=w, literal=lit, extended=ext, dictionary=d, lazy_matching=lazy)
            assert int(res.status[j]) == st, (it, j, w, lit, ext, lazy, len(x))
            assert res.stream(j) == want, (it, j, w, lit, ext, lazy, len(x))
            comps.append(want)
        # decode: exact, short and corrupted streams
        dec_in, caps = [], []
        for x, cmp_ in zip(datas, comps):
            dec_in.append(cmp_), caps.append(len(x) + 8)
            dec_in.append(cmp_), caps.append(max(0, len(x) - rng.randrange(0, 3)))
            if len(cmp_) > 2:
                bad = bytearray(cmp_)
                bad[rng.randrange(1, len(bad))] ^= 1 << rng.randrange(8)
                dec_in.append(bytes(bad[: rng.randrange(2, len(bad) + 1)])), caps.append(len(x) + 300)
        back = ta.decompress_batch(dec_in, out_cap=np.array(caps, dtype=np.uint32), dictionary=d)
        for j, (cmp_, cap) in enumerate(zip(dec_in, caps)):
            want = oracle.decompress(cmp_, dictionary=d, cap=cap)
            got = (int(back.status[j]), back.stream(j), int(back.in_consumed[j]))
            assert got == want, (it, j, w, lit, ext, cap, got[0], want[0], len(got[1]), len(want[1]))


def _runny_text(rng, n):
    """Indented / ruled / padded text: runs of 8-80 equal bytes between short words (what the run-aware build lists)."""
    out = bytearray()
    while len(out) < n:
        k = rng.randrange(6)
        if k == 0:
            out += b"\n" + b" " * rng.choice([4, 8, 8, 12, 16, 20, 40])
        elif k == 1:
            out += bytes([rng.choice(b"-=* #")]) * rng.randrange(7, 80)
        elif k == 2:
            out += b" " * rng.randrange(1, 12)
        else:
            out += bytes(rng.choice(b"abcdefgh(): ") for _ in range(rng.randrange(1, 9)))
    return bytes(out[:n])


def test_run_aware_build_matches_oracle(ta, oracle):
    """TAMP_AMD_HINT_RUNS (long runs listed once instead of indexed byte by byte) produces the same bytes as the
    reference on run-heavy and ordinary inputs, every window size, both formats; AUTO on host batches likewise."""
    from tamp_amd import workloads as wl

    rng = random.Random(11)
    for it in range(60):
        w = rng.choice([8, 9, 10, 10, 10, 11, 12, 13, 15])
        lit = rng.choice([8, 8, 8, 7])
        ext = rng.random() < 0.7
        d = None
        if rng.random() < 0.3:
            d = (_runny_text(rng, 1 << w) if rng.random() < 0.5 else _rand_inputs(rng, wl, 1 << w) + bytes(1 << w))[: 1 << w]
        datas = []
        for _ in range(rng.randrange(1, 16)):
            n = rng.choice([0, 1, 7, 8, 9, 17, 100, 256, 1000, 4096, 4096, rng.randrange(1, 12000)])
            x = _runny_text(rng, n) if rng.random() < 0.6 else _rand_inputs(rng, wl, n)
            if lit < 8:
                x = bytes(b & 0x7F for b in x)
            datas.append(x)
        for mode in (True, None):
            res = ta.compress_batch(datas, window=w, literal=lit, extended=ext, dictionary=d, run_aware=mode)
            for j, x in enumerate(datas):
                st, want = oracle.compress(x, window=w, literal=lit, extended=ext, dictionary=d)
                assert int(res.status[j]) == st, (it, j, w, lit, ext, mode, len(x))
                assert res.stream(j) == want, (it, j, w, lit, ext, mode, len(x))
    # a batch the host sampler routes to the run-aware build (source-code-like), at config-2 stream size
    rows = np.frombuffer(b"".join(_runny_text(rng, 4096) for _ in range(512)), dtype=np.uint8).reshape(512, 4096)
    off, ln = wl.csr_for_fixed(512, 4096)
    res = ta.compress_batch(rows.reshape(-1), off, ln, max_in_len=4096)
    want = oracle.compress_batch(rows.reshape(-1), off, ln, threads=8)
    for j in range(512):
        assert res.stream(j) == want.stream(j), j


def test_more_streams_than_one_launch_holds(ta, oracle):
    """One stream per workgroup: a batch above 2^20 streams takes several launches (first_stream); every stream must
    still come out, in place."""
    import torch

    n, L = (1 << 20) + 4099, 24
    rng = np.random.default_rng(3)
    rows = rng.integers(97, 101, (n, L), dtype=np.uint8)          # four letters: short matches, a few runs
    rows[:, :4] = np.frombuffer(np.arange(n, dtype=np.uint32).tobytes(), dtype=np.uint8).reshape(n, 4) | 0x40
    dev = torch.device("cuda:0")
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off = torch.arange(n, dtype=torch.int64, device=dev) * L
    ln = torch.full((n,), L, dtype=torch.int32, device=dev)
    res = ta.compress_batch(data, off, ln, max_in_len=L)
    assert bool((res.status == 0).all().item())
    sel = [0, 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, n - 1] + [int(x) for x in rng.integers(0, n, 200)]
    for i in sel:
        st, want = oracle.compress(rows[i].tobytes())
        assert st == 0 and res.stream(i) == want, i
    back = ta.decompress_batch(res.out, res.out_off, res.out_len, out_cap=L + 8)
    assert bool((back.status == 2).all().item()) and bool((back.out_len == L).all().item())
    got = back.out.view(-1)[: n * (L + 8)].view(n, L + 8)[:, :L]
    assert bool((got == data.view(n, L)).all().item())


def test_excess_bits_and_invalid_conf(ta, oracle):
    from tamp_amd import workloads as wl

    tel = wl.telemetry(8, 256).copy()
    tel[1, 40] = 0xC3
    tel[5, 0] = 0x80
    d = wl.telemetry_dictionary(bytes(ta.initialize_dictionary(256, literal=7)))
    res = ta.compress_batch([r.tobytes() for r in tel], window=8, literal=7, dictionary=d)
    for j in range(8):
        st, want = oracle.compress(tel[j].tobytes(), window=8, literal=7, dictionary=d)
        assert (int(res.status[j]), res.stream(j)) == (st, want)
    assert int(res.status[1]) == -2 and int(res.status[5]) == -2 and int(res.status[0]) == 0
    res = ta.compress_batch([b"abc"], window=16)
    assert int(res.status[0]) == -3
    res = ta.compress_batch([b"abc" * 100], out_cap=np.array([10], dtype=np.uint32))
    assert int(res.status[0]) == 1 and int(res.out_len[0]) <= 10


def test_full_size_config2_properties(ta, oracle):
    """BASELINE config 2 at full size (65,536 x 4 KiB, w=10): device-resident, size-independent checks --
    every stream OK, GPU decode(GPU encode(x)) == x for all streams, and a sample of streams bit-exact vs oracle."""
    import torch

    from tamp_amd import workloads as wl

    n = 65536
    rows = wl.synth_text(n, 4096)
    in_off, in_len = wl.csr_for_fixed(n, 4096)
    dev = torch.device("cuda:0")
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t = torch.from_numpy(in_off.astype(np.int64)).to(dev)
    len_t = torch.from_numpy(in_len.astype(np.int32)).to(dev)
    for ext in (True, False):
        res = ta.compress_batch(data, off_t, len_t, window=10, literal=8, extended=ext, max_in_len=4096)
        torch.cuda.synchronize()
        assert bool((res.status == 0).all())
        olen = res.out_len.cpu().numpy()
        ratio = olen.sum() / rows.size
        assert 0.4 < ratio < 0.6, ratio
        back = ta.decompress_batch(res.out, res.out_off, res.out_len, out_cap=4096)
        torch.cuda.synchronize()
        # cap == exact size with pad bits left -> the reference reports OUTPUT_FULL or INPUT_EXHAUSTED; both mean done
        assert bool(((back.status == 1) | (back.status == 2)).all())
        assert bool((back.out_len == 4096).all())
        assert torch.equal(back.out[: n * 4096], data)
        sample = list(range(0, n, 997))
        want = oracle.compress_batch(rows[sample].reshape(-1), *wl.csr_for_fixed(len(sample), 4096), extended=ext, threads=8)
        out_host = res.out.cpu().numpy()
        offs = res.out_off.cpu().numpy()
        for k, i in enumerate(sample):
            assert out_host[offs[i] : offs[i] + olen[i]].tobytes() == want.stream(k), (ext, i)


def test_long_single_stream(ta, oracle):
    """BASELINE config 1 shape: one 64 KiB stream (many epochs), plus w=15."""
    from tamp_amd import workloads as wl

    x = wl.synth_text(1, 65536)[0].tobytes()
    for conf in (dict(window=10, extended=True), dict(window=10, extended=False), dict(window=15, extended=True),
                 dict(window=12, literal=7, extended=True)):
        data = x if conf.get("literal", 8) == 8 else bytes(b & 0x7F for b in x)
        st, want = oracle.compress(data, **conf)
        res = ta.compress_batch([data], **conf)
        assert int(res.status[0]) == st and res.stream(0) == want, conf
        assert ta.decompress(want) == data


def _replay(ta, ops, conf):
    import io

    f = io.BytesIO()
    c = ta.Compressor(f, **conf)
    for op in ops:
        if op[0] == "write":
            c.write(op[1])
        elif op[0] == "flush":
            c.flush(write_token=bool(op[1]))
        elif op[0] == "reset":
            c.reset_dictionary()
        elif op[0] == "close":
            c.close()
    return f.getvalue()


def test_streaming_scripts_golden(ta):
    """tamp_amd.Compressor driven like tests/test_compressor_decompressor.py:312-556 drives tamp.Compressor:
    every byte that reaches the file equals what the reference object wrote (tests/golden/streaming.json)."""
    recs = load_golden("streaming.json")
    sessions = {}
    for rec in recs:
        conf = dict(rec["conf"])
        if "dictionary" in conf:
            conf["dictionary"] = unb64(conf["dictionary"])
        ops = [("write", unb64(op[1])) if op[0] == "write" else tuple(op) for op in rec["ops"]]
        assert rec["status"] == 0
        got = _replay(ta, ops, conf)
        assert got == unb64(rec["expected"]), (rec["name"], rec["cite"])
        sessions[rec["name"]] = (got, b"".join(op[1] for op in ops if op[0] == "write"))
        if rec["decodes"]:
            back = ta.decompress(got, dictionary=conf.get("dictionary"))
            assert hashlib.sha256(bytes(back)).hexdigest() == rec["plain_sha256"], rec["name"]
    # append mode: two sessions concatenated decode as one stream (test_append_mode_roundtrip, :510-530)
    (a, pa), (b, pb) = sessions["append_session_1"], sessions["append_session_2"]
    assert bytes(ta.decompress(a + b)) == pa + pb


def test_streaming_differential_vs_oracle(ta, oracle):
    from tamp_amd import workloads as wl

    rng = random.Random(99)
    srcs = [bytes(wl.synth_text(1, 20000, first_index=123)[0]), bytes(wl.lcg_runs(1, 8000, first_index=9)[0]),
            bytes(wl.stress(3, 8192)[1]), bytes(wl.stress(3, 8192)[2]), b"\0" * 6000]
    for k in range(60):
        src = rng.choice(srcs)
        dr = rng.random() < 0.5
        conf = dict(window=rng.choice([8, 9, 10, 12, 15]), literal=8, extended=rng.random() < 0.7,
                    dictionary_reset=dr, append=dr and rng.random() < 0.2, lazy_matching=rng.random() < 0.2)
        pos, ops = rng.randrange(0, 2000), []
        for _ in range(rng.randrange(1, 7)):
            x = rng.random()
            if x < 0.55:
                n = rng.choice([0, 1, 2, 3, 15, 16, 17, 31, 100, 700, 3000])
                ops.append(("write", src[pos : pos + n]))
                pos += n
            elif x < 0.85:
                ops.append(("flush", rng.random() < 0.7))
            elif dr:
                ops.append(("reset",))
        ops.append(("close",))
        st, want = oracle.stream_script(ops, **conf)
        assert st == 0
        assert _replay(ta, ops, conf) == want, (k, conf, [(o[0], len(o[1]) if o[0] == "write" else o[1:]) for o in ops])


def test_streaming_errors(ta):
    import io

    with pytest.raises(ValueError):
        ta.Compressor(io.BytesIO(), append=True)  # append needs dictionary_reset (compressor.c:209)
    with pytest.raises(ValueError):
        ta.Compressor(io.BytesIO()).reset_dictionary()  # TAMP_INVALID_CONF (compressor.c:846)
    c = ta.Compressor(io.BytesIO(), literal=7, extended=False)
    c.write(b"\xff")
    with pytest.raises(ta.ExcessBitsError):  # tests/test_compressor.py:239-246
        c.flush()


def test_reference_named_c_api(ta, oracle):
    """include/tamp_compat.h: the reference's own symbols (ctests/test_compressor.c round-trip shape) via ctypes."""
    import ctypes as C

    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    lib = _lib.load()

    class TampConf(C.Structure):
        _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                    ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1),
                    ("lazy_matching", C.c_uint16, 1)]

    comp_t, decomp_t = C.c_ubyte * 48, C.c_ubyte * 24
    lib.tamp_compressor_init.restype = C.c_int8
    lib.tamp_compressor_compress_and_flush.restype = C.c_int8
    lib.tamp_compressor_compress_and_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                                       C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_bool]
    lib.tamp_decompressor_init.restype = C.c_int8
    lib.tamp_decompressor_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint8]
    lib.tamp_decompressor_decompress.restype = C.c_int8
    lib.tamp_decompressor_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p,
                                                 C.c_size_t, C.POINTER(C.c_size_t)]
    data = wl.synth_text(1, 3000)[0].tobytes()
    for kw in (dict(window=10, literal=8, extended=1), dict(window=10, literal=8, extended=0),
               dict(window=9, literal=8, extended=1, use_custom_dictionary=1)):
        conf = TampConf(**kw)
        W = 1 << conf.window
        window = (C.c_ubyte * W)()
        dic = None
        if conf.use_custom_dictionary:
            dic = wl.synth_text(1, W, first_index=77)[0].tobytes()
            C.memmove(window, dic, W)
        c = comp_t()
        assert lib.tamp_compressor_init(c, C.byref(conf), window) == 0
        out = (C.c_ubyte * 4096)()
        written, consumed = C.c_size_t(0), C.c_size_t(0)
        src = (C.c_ubyte * len(data)).from_buffer_copy(data)
        r = lib.tamp_compressor_compress_and_flush(c, out, 4096, C.byref(written), src, len(data), C.byref(consumed), False)
        st, want = oracle.compress(data, window=conf.window, literal=8, extended=bool(conf.extended), dictionary=dic)
        assert (r, bytes(out[: written.value]), consumed.value) == (st, want, len(data))
        # a second call on the same object is the next segment of the same stream (window carried in `window`)
        r = lib.tamp_compressor_compress_and_flush(c, out, 4096, C.byref(written), src, 10, C.byref(consumed), False)
        st2, both = oracle.stream_script([("write", data), ("flush", False), ("write", data[:10]), ("flush", False)],
                                         window=conf.window, literal=8, extended=bool(conf.extended), dictionary=dic)
        assert (r, st2, bytes(out[: written.value]), consumed.value) == (0, 0, both[len(want):], 10)
        # decode: conf from the header ...
        if dic is not None:
            C.memmove(window, dic, W)
        d = decomp_t()
        assert lib.tamp_decompressor_init(d, None, window, conf.window) == 0
        back = (C.c_ubyte * 4096)()
        cbuf = (C.c_ubyte * len(want)).from_buffer_copy(want)
        r = lib.tamp_decompressor_decompress(d, back, 4096, C.byref(written), cbuf, len(want), C.byref(consumed))
        assert (r, bytes(back[: written.value]), consumed.value) == (2, data, len(want))
        # ... or handed to init, with the input starting after the header (the Cython binding's way); the window buffer
        # was decoded into, as with the reference, so a custom dictionary has to be put back first
        if dic is not None:
            C.memmove(window, dic, W)
        d = decomp_t()
        assert lib.tamp_decompressor_init(d, C.byref(conf), window, conf.window) == 0
        cbuf = (C.c_ubyte * (len(want) - 1)).from_buffer_copy(want[1:])
        r = lib.tamp_decompressor_decompress(d, back, 4096, C.byref(written), cbuf, len(want) - 1, C.byref(consumed))
        assert (r, bytes(back[: written.value]), consumed.value) == (2, data, len(want) - 1)
    # decompressor.c:311: header asks for a bigger window than the caller's buffer
    d = decomp_t()
    window = (C.c_ubyte * 256)()
    assert lib.tamp_decompressor_init(d, None, window, 8) == 0
    want = oracle.compress(b"abcabcabc")[1]
    cbuf = (C.c_ubyte * len(want)).from_buffer_copy(want)
    assert lib.tamp_decompressor_decompress(d, (C.c_ubyte * 64)(), 64, None, cbuf, len(want), None) == -3


def test_reference_named_streaming_c_api(ta, oracle):
    """tamp_compressor_flush / _reset_dictionary / tamp_compress_stream / tamp_decompress_stream under the reference's
    names (compressor.h:193,217,338; decompressor.h:190), with the library's own memory adaptors (common.h:255-290)."""
    import ctypes as C

    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    lib = _lib.load()

    class TampConf(C.Structure):
        _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                    ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1),
                    ("lazy_matching", C.c_uint16, 1)]

    class MemReader(C.Structure):
        _fields_ = [("data", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    class MemWriter(C.Structure):
        _fields_ = [("data", C.c_void_p), ("capacity", C.c_size_t), ("pos", C.c_size_t)]

    sz = C.POINTER(C.c_size_t)
    for fn, args in (("tamp_compressor_init", [C.c_void_p, C.c_void_p, C.c_void_p]),
                     ("tamp_compressor_compress_and_flush", [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_void_p,
                                                             C.c_size_t, sz, C.c_bool]),
                     ("tamp_compressor_flush", [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_bool]),
                     ("tamp_compressor_reset_dictionary", [C.c_void_p, C.c_void_p, C.c_size_t, sz]),
                     ("tamp_compress_stream", [C.c_void_p] * 5 + [sz, sz, C.c_void_p, C.c_void_p]),
                     ("tamp_decompress_stream", [C.c_void_p] * 5 + [sz, sz, C.c_void_p, C.c_void_p]),
                     ("tamp_decompressor_init", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint8])):
        getattr(lib, fn).restype = C.c_int8
        getattr(lib, fn).argtypes = args
    mem_read = C.cast(lib.tamp_stream_mem_read, C.c_void_p)
    mem_write = C.cast(lib.tamp_stream_mem_write, C.c_void_p)
    text = wl.synth_text(1, 9000, first_index=31)[0].tobytes()
    d1, d2 = text[:2500], text[2500:9000]

    # 1. flush tokens + reset_dictionary, the ctests/test_compressor.c:403-613 shape
    for lazy in (0, 1):
        conf = TampConf(window=10, literal=8, extended=1, dictionary_reset=1, lazy_matching=lazy)
        window, comp = (C.c_ubyte * 1024)(), (C.c_ubyte * 48)()
        assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
        got = bytearray()
        out, w, k = (C.c_ubyte * 16384)(), C.c_size_t(0), C.c_size_t(0)
        s1 = (C.c_ubyte * len(d1)).from_buffer_copy(d1)
        s2 = (C.c_ubyte * len(d2)).from_buffer_copy(d2)
        assert lib.tamp_compressor_compress_and_flush(comp, out, 16384, C.byref(w), s1, len(d1), C.byref(k), True) == 0
        got += bytes(out[: w.value])
        assert lib.tamp_compressor_flush(comp, out, 16384, C.byref(w), True) == 0  # redundant: suppressed
        assert w.value == 0
        assert lib.tamp_compressor_reset_dictionary(comp, out, 16384, C.byref(w)) == 0
        got += bytes(out[: w.value])
        assert lib.tamp_compressor_compress_and_flush(comp, out, 16384, C.byref(w), s2, len(d2), C.byref(k), False) == 0
        got += bytes(out[: w.value])
        st, want = oracle.stream_script([("write", d1), ("flush", True), ("flush", True), ("reset",), ("write", d2),
                                         ("flush", False)], dictionary_reset=True, lazy_matching=bool(lazy))
        assert st == 0 and bytes(got) == want
        assert bytes(ta.decompress(bytes(got))) == d1 + d2
    conf = TampConf(window=10, literal=8, extended=1)
    comp = (C.c_ubyte * 48)()
    assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
    assert lib.tamp_compressor_reset_dictionary(comp, out, 16384, C.byref(w)) == -3  # compressor.c:846

    # 2. callback stream API, ctests/test_stream.c round-trip shape
    for ext in (1, 0):
        conf = TampConf(window=10, literal=8, extended=ext)
        assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
        src = (C.c_ubyte * len(text)).from_buffer_copy(text)
        dst = (C.c_ubyte * 16384)()
        rd, wr = MemReader(C.addressof(src), len(text), 0), MemWriter(C.addressof(dst), 16384, 0)
        cin, cout = C.c_size_t(0), C.c_size_t(0)
        assert lib.tamp_compress_stream(comp, mem_read, C.byref(rd), mem_write, C.byref(wr), C.byref(cin),
                                        C.byref(cout), None, None) == 0
        st, want = oracle.compress(text, extended=bool(ext))
        assert (cin.value, cout.value, bytes(dst[: wr.pos])) == (len(text), len(want), want)
        dec = (C.c_ubyte * 24)()
        assert lib.tamp_decompressor_init(dec, None, window, 10) == 0
        back = (C.c_ubyte * 16384)()
        rd, wr = MemReader(C.addressof(dst), len(want), 0), MemWriter(C.addressof(back), 16384, 0)
        assert lib.tamp_decompress_stream(dec, mem_read, C.byref(rd), mem_write, C.byref(wr), C.byref(cin),
                                          C.byref(cout), None, None) == 0
        assert (cin.value, cout.value, bytes(back[: wr.pos])) == (len(want), len(text), text)
        # a writer that is too small: TAMP_WRITE_ERROR (common.h:163)
        assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
        rd, wr = MemReader(C.addressof(src), len(text), 0), MemWriter(C.addressof(dst), 100, 0)
        assert lib.tamp_compress_stream(comp, mem_read, C.byref(rd), mem_write, C.byref(wr), None, None, None,
                                        None) == -12


def test_mixed_window_batch_and_header_prepass(ta, oracle):
    """BASELINE config 4 shape: one decode launch over streams whose windows differ (w in 8..12).  The window limit
    keeps the reference's meaning (decompressor.c:311) whether the library sizes its on-chip windows from a header
    pre-pass (default) or from the limit itself (scan_headers=False)."""
    from tamp_amd import workloads as wl

    rng = random.Random(4)
    rows = wl.synth_text(640, 1500, first_index=900)
    streams, plain, wbits = [], [], []
    for i in range(rows.shape[0]):
        w = rng.choice([8, 9, 10, 11, 12])
        data = rows[i].tobytes()[: rng.randrange(1, 1500)]
        st, comp = oracle.compress(data, window=w, extended=bool(i & 1))
        assert st == 0
        streams.append(comp), plain.append(data), wbits.append(w)
    for limit in (15, 12, 10, 8):
        for scan in (True, False):
            res = ta.decompress_batch(streams, out_cap=1500, max_window_bits=limit, scan_headers=scan)
            for i, (p, w) in enumerate(zip(plain, wbits)):
                if w > limit:
                    assert int(res.status[i]) == -3, (limit, scan, i)
                else:
                    assert int(res.status[i]) == 2 and res.stream(i) == p, (limit, scan, i, w)
    # every header above the limit: nothing decodes, nothing crashes
    big = [s for s, w in zip(streams, wbits) if w == 12] * 3
    res = ta.decompress_batch(big[:300], out_cap=1500, max_window_bits=9)
    assert (np.asarray(res.status[:300]) == -3).all()


def test_decoder_variants_long_streams(ta, oracle, monkeypatch):
    """All decoders (wave per stream; lane per stream with LDS windows, lean / bulk build; lane per stream with the
    windows in the global scratch slab) on streams long enough for the bulk path:
    plain, extended (RLE + extended-match tokens), custom dictionary, FLUSH / dictionary_reset streams, truncated
    input and restricted output -- status, length, bytes and consumed counts against the oracle."""
    from tamp_amd import workloads as wl

    rng = random.Random(12)
    text = [wl.synth_text(1, 6000, first_index=300 + i)[0].tobytes() for i in range(6)]
    runs = [wl.lcg_runs(1, 5000, first_index=i)[0].tobytes() for i in range(3)]
    stress = [wl.stress(3, 6000)[i].tobytes() for i in range(3)]
    d10 = wl.synth_text(1, 1024, first_index=999)[0].tobytes()
    cases = []  # (compressed, dictionary)
    for data in text + runs + stress:
        for kw in (dict(), dict(extended=False), dict(window=8), dict(window=9, literal=8, extended=True)):
            st, comp = oracle.compress(data, **kw)
            assert st == 0
            cases.append((comp, None))
    for data in text[:3]:
        st, comp = oracle.compress(data, dictionary=d10)
        cases.append((comp, d10))
    for rec in load_golden("streaming.json"):  # FLUSH tokens, double-FLUSH resets
        if rec["decodes"] and "dictionary" not in rec["conf"] and rec["conf"].get("window", 10) <= 10:
            cases.append((unb64(rec["expected"]), None))
    # truncated inputs
    for comp, d in list(cases[:12]):
        cases.append((comp[: rng.randrange(40, len(comp))], d))
    for mode in ("wave", "lane", "global", "split"):
        monkeypatch.setenv("TAMP_AMD_DECODER", mode)
        for d in (None, d10):
            group = [c for c, dd in cases if dd is d]
            for cap in (8192, 1000, 4097, 3):
                res = ta.decompress_batch(group, out_cap=cap, dictionary=d, max_window_bits=10)
                for i, comp in enumerate(group):
                    st, want, consumed = oracle.decompress(comp, dictionary=d, cap=cap, max_window_bits=10)
                    assert (int(res.status[i]), res.stream(i)) == (st, want), (mode, cap, i, len(comp))
                    assert int(res.in_consumed[i]) == consumed, (mode, cap, i)


def test_mixed_window_binning_large_batch(ta, oracle):
    """BASELINE config 4 at a size where the library picks the lane-per-stream decoder with global-memory windows
    (>= 49,152 streams, some windows above 2^10): every window size in one launch."""
    from tamp_amd import workloads as wl

    n, L = 66560, 320
    rows = wl.synth_text(n, L, first_index=5000)
    wsel = np.arange(n) % 5 + 8  # windows 8..12, interleaved
    comp = [None] * n
    for w in range(8, 13):
        ids = np.nonzero(wsel == w)[0]
        sub = np.ascontiguousarray(rows[ids])
        res = oracle.compress_batch(sub.reshape(-1), *wl.csr_for_fixed(len(ids), L), window=w, threads=16)
        assert (res.status == 0).all()
        for k, i in enumerate(ids):
            comp[i] = res.stream(k)
    comp[7] = b""           # an empty stream
    comp[8] = comp[8][:1]   # header only
    back = ta.decompress_batch(comp, out_cap=L + 8)  # (an exact cap ends most streams with OUTPUT_FULL, as in the reference)
    st = np.asarray(back.status)
    olen = np.asarray(back.out_len)
    bad = [i for i in range(n) if i not in (7, 8) and (st[i] != 2 or back.stream(i) != rows[i].tobytes())]
    assert not bad, bad[:10]
    assert st[7] == 2 and olen[7] == 0 and st[8] == 2 and olen[8] == 0
    # the limit still means what it means in the reference when the batch is binned
    back = ta.decompress_batch(comp, out_cap=L + 8, max_window_bits=11)
    st = np.asarray(back.status)
    assert (st[wsel == 12] == -3).all() and (st[(wsel < 12) & (np.arange(n) > 8)] == 2).all()


def test_concurrent_streams_do_not_share_decoder_scratch(ta, monkeypatch):
    """Two decode calls in flight on two HIP streams (global-window decoder: per-lane window slots in a scratch slab):
    each stream owns its slab, so neither corrupts the other's windows."""
    import torch
    from tamp_amd import workloads as wl

    dev = torch.device("cuda:0")
    monkeypatch.setenv("TAMP_AMD_DECODER", "global")
    jobs = []
    for k, w in enumerate((10, 12)):
        n, L = 20000, 2048
        rows = wl.synth_text(n, L, first_index=70000 * (k + 1))
        off, ln = wl.csr_for_fixed(n, L)
        data = torch.from_numpy(rows.reshape(-1)).to(dev)
        r = ta.compress_batch(data, torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev),
                              window=w, max_in_len=L)
        jobs.append((torch.cuda.Stream(dev), r, data, n, L))
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):  # launches alternate between the two streams and overlap on the device
        for s, r, data, n, L in jobs:
            with torch.cuda.stream(s):
                outs.append((ta.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L + 8, scan_headers=False), data, n, L))
    torch.cuda.synchronize()
    for back, data, n, L in outs:
        assert bool((back.out_len == L).all().item())
        got = back.out[: n * (L + 8)].view(n, L + 8)[:, :L].reshape(-1)
        assert torch.equal(got, data)


def test_host_memory_batches_run_as_overlapping_chunks(ta, oracle, monkeypatch):
    """Host-memory calls (TAMP_AMD_MEM_HOST) are cut into chunks of consecutive streams that overlap on the library's
    own HIP streams: results must not depend on where the cuts fall -- ragged and empty streams, per-stream error
    codes, consumed counts, pinned and pageable buffers, an unordered offset table (one chunk), and two host threads
    calling at once."""
    import ctypes as C
    import threading

    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    lib = _lib.load()
    rng = random.Random(99)
    datas = []
    for i in range(1500):
        n = rng.choice([0, 1, 17, 300, 2048, 4096, rng.randrange(1, 6000)])
        x = _rand_inputs(rng, wl, n)
        datas.append(x)
    datas[7] = b"\xff" * 40 + datas[7]  # fine with literal=8; gives EXCESS_BITS below with literal=7
    want8 = [oracle.compress(x, window=10, literal=8) for x in datas]
    want7 = [oracle.compress(x, window=9, literal=7) for x in datas[:200]]

    monkeypatch.setenv("TAMP_AMD_HOST_CHUNK_MB", "1")
    for streams in ("16", "200", "100000"):
        monkeypatch.setenv("TAMP_AMD_HOST_CHUNK_STREAMS", streams)
        res = ta.compress_batch(datas, window=10, literal=8)
        for j, (st, blob) in enumerate(want8):
            assert int(res.status[j]) == st and res.stream(j) == blob, (streams, j)
        res7 = ta.compress_batch(datas[:200], window=9, literal=7)
        for j, (st, blob) in enumerate(want7):
            assert int(res7.status[j]) == st and res7.stream(j) == blob, (streams, j)
        assert any(st == _lib.EXCESS_BITS for st, _ in want7)
        # decode with exact, short and generous capacities, and a truncated stream
        dec_in, caps = [], []
        for x, (st, blob) in zip(datas, want8):
            dec_in.append(blob), caps.append(len(x) + rng.randrange(0, 9))
            if len(x) > 4:
                dec_in.append(blob), caps.append(len(x) - rng.randrange(1, 4))
                dec_in.append(blob[: len(blob) // 2]), caps.append(len(x))
        back = ta.decompress_batch(dec_in, out_cap=np.array(caps, dtype=np.uint32))
        for j, (blob, cap) in enumerate(zip(dec_in, caps)):
            want = oracle.decompress(blob, cap=cap)
            assert (int(back.status[j]), back.stream(j), int(back.in_consumed[j])) == want, (streams, j)

    # pinned buffers (tamp_amd_host_alloc) and an offset table that is not ascending: one chunk, same bytes
    monkeypatch.setenv("TAMP_AMD_HOST_CHUNK_STREAMS", "64")
    flat, in_off, in_len = ta.pack_streams(datas)
    n = len(datas)
    cap = np.array([ta.compress_bound(len(x), 8, False) for x in datas], dtype=np.uint32)
    out_off = np.cumsum(cap, dtype=np.uint64) - cap

    def pinned(arr):
        p = lib.tamp_amd_host_alloc(max(arr.nbytes, 1))
        assert p
        view = np.frombuffer((C.c_ubyte * max(arr.nbytes, 1)).from_address(p), dtype=arr.dtype, count=arr.size)
        view[:] = arr
        return p, view

    held = [pinned(a) for a in (flat, in_off.astype(np.uint64), in_len.astype(np.uint32), out_off, cap)]
    p_out, v_out = pinned(np.zeros(int(cap.sum()) + 1, np.uint8))
    p_len, v_len = pinned(np.zeros(n, np.uint32))
    p_st, v_st = pinned(np.zeros(n, np.int8))
    conf = _lib.TampAmdConf(window=10, literal=8, extended=1)
    rc = lib.tamp_batch_compress(C.byref(conf), None, held[0][0], held[1][0], held[2][0], p_out, held[3][0], held[4][0],
                                 p_len, p_st, n, 0, _lib.MEM_HOST, 0, None)
    assert rc == 0
    for j, (st, blob) in enumerate(want8):
        assert int(v_st[j]) == st and bytes(v_out[int(out_off[j]) : int(out_off[j]) + int(v_len[j])]) == blob, j
    for p, _ in held + [(p_out, 0), (p_len, 0), (p_st, 0)]:
        lib.tamp_amd_host_free(p)

    perm = np.arange(n)[::-1].copy()  # same slab, streams listed last-first
    res = ta.compress_batch(flat, in_off[perm], in_len[perm], window=10)
    for k, j in enumerate(perm):
        assert int(res.status[k]) == want8[j][0] and res.stream(k) == want8[j][1], (k, j)

    # one contiguous shard of streams per device, a host thread each (TAMP_AMD_ALL_DEVICES; here three shards that
    # share the one visible device)
    monkeypatch.setenv("TAMP_AMD_FANOUT", "3")
    res = ta.compress_batch(datas, window=10, device=_lib.ALL_DEVICES)
    for j, (st, blob) in enumerate(want8):
        assert int(res.status[j]) == st and res.stream(j) == blob, j
    back = ta.decompress_batch([b for _, b in want8], out_cap=np.array([len(x) + 8 for x in datas], dtype=np.uint32),
                               device=_lib.ALL_DEVICES)
    for j, x in enumerate(datas):
        assert int(back.status[j]) == 2 and back.stream(j) == x, j
    monkeypatch.delenv("TAMP_AMD_FANOUT")

    errs = []

    def worker(lo, hi):
        try:
            for _ in range(3):
                r = ta.compress_batch(datas[lo:hi], window=10)
                for j in range(lo, hi):
                    assert r.stream(j - lo) == want8[j][1]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(0, 700)), threading.Thread(target=worker, args=(700, 1500))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs


def _resume_golden(rec):
    conf = tuple(rec["conf"]) if rec["conf"] is not None else None
    dic = unb64(rec["dictionary"]) if rec["dictionary"] else None
    return (unb64(rec["data"]), [tuple(s) for s in rec["script"]], conf, dic,
            [(r, unb64(out), k) for r, out, k in rec["calls"]])


def _drive_decoder_batch(ta, jobs, window_bits):
    """jobs: [(data, script, conf, dictionary)] sharing conf-kind and dictionary per DecoderBatch is not required by the
    kernel, but DecoderBatch gives every object the same initial state: group jobs accordingly before calling."""
    from tamp_amd import _lib

    data0, _, conf, dic = jobs[0]
    tconf = None
    if conf is not None:
        tconf = _lib.TampAmdConf(window=conf[0], literal=conf[1], use_custom_dictionary=int(conf[2]), extended=int(conf[3]),
                                 dictionary_reset=int(conf[4]))
    try:
        batch = ta.DecoderBatch(len(jobs), window_bits=window_bits, conf=tconf, dictionary=dic)
    except ValueError as e:
        return int(str(e).rsplit(" ", 1)[1]), None
    pos = [0] * len(jobs)
    calls = [[] for _ in jobs]
    for step in range(max(len(j[1]) for j in jobs)):
        chunks, caps = [], []
        for i, (data, script, _, _) in enumerate(jobs):
            take, cap = script[step] if step < len(script) else (0, 0)
            chunks.append(data[pos[i] : pos[i] + take]), caps.append(cap)
        status, outs, consumed = batch.step(chunks, np.array(caps, dtype=np.uint32))
        for i, (data, script, _, _) in enumerate(jobs):
            if step < len(script):
                calls[i].append((int(status[i]), outs[i], int(consumed[i])))
                pos[i] += int(consumed[i])
    return 0, calls


def test_decoder_resume_golden_scripts(ta):
    """Decoder objects advanced call by call on the device (tamp_batch_decompress_resume) against what one reference
    TampDecompressor returned for the same calls (tests/golden/decoder_resume.json): status, bytes and consumed count
    of every call -- tokens cut short by a full output buffer, by the end of the input, headers split over calls."""
    recs = load_golden("decoder_resume.json") + load_golden("decoder_resume_fuzz.json")
    groups = {}
    for rec in recs:
        data, script, conf, dic, want = _resume_golden(rec)
        groups.setdefault((conf, dic, rec["window_bits"]), []).append((rec, data, script, want))
    for (conf, dic, wb), items in groups.items():
        r0, calls = _drive_decoder_batch(ta, [(d, s, conf, dic) for _, d, s, _ in items], wb)
        for k, (rec, _, _, want) in enumerate(items):
            assert r0 == rec["init"], rec["name"]
            if calls is not None:
                assert calls[k] == want, rec["name"]


def test_decoder_resume_differential_vs_oracle(ta, oracle):
    """Random streams (flush tokens, dictionary resets, corrupted bytes), random chunking of input and output room,
    hundreds of objects per launch: every call equals the oracle's resumable decoder."""
    from tamp_amd import workloads as wl

    rng = random.Random(4242)
    for rnd in range(6):
        w = rng.randrange(8, 16)
        wb = rng.choice([15, w, w])
        lit = rng.choice([7, 8, 8])
        jobs = []
        for j in range(120):
            ext = rng.random() < 0.7
            dr = rng.random() < 0.3
            x = _rand_inputs(rng, wl, rng.choice([0, 1, 40, 700, 3000, rng.randrange(1, 5000)]))
            if lit < 8:
                x = bytes(b & 127 for b in x)
            ops, pos = [], 0
            while pos < len(x):
                k = rng.randrange(1, 900)
                ops.append(("write", x[pos : pos + k]))
                pos += k
                u = rng.random()
                if u < 0.15:
                    ops.append(("flush", rng.random() < 0.7))
                elif u < 0.22 and dr:
                    ops.append(("reset",))
            ops.append(("close",))
            st, blob = oracle.stream_script(ops, window=w, literal=lit, extended=ext, dictionary_reset=dr)
            assert st == 0
            if rng.random() < 0.2 and len(blob) > 3:
                b = bytearray(blob)
                b[rng.randrange(1, len(b))] ^= 1 << rng.randrange(8)
                blob = bytes(b)
            script = [(rng.choice([0, 1, 2, 3, 5, 9, 40, 1000, 100000]), rng.choice([0, 1, 2, 3, 7, 20, 64, 300, 5000]))
                      for _ in range(rng.randrange(1, 50))]
            script.append((1 << 20, 1 << 15))
            jobs.append((blob, script, None, None))
        r0, calls = _drive_decoder_batch(ta, jobs, wb)
        assert r0 == 0
        for j, (blob, script, _, _) in enumerate(jobs):
            want0, want = oracle.decode_script(blob, script, window_bits=wb)
            assert want0 == 0 and calls[j] == want, (rnd, j, w, wb, lit)


def test_reference_named_decompressor_object_resumes(ta, oracle):
    """tamp_decompressor_decompress under the reference's name, called the way its own stream loop and the Cython
    wrapper call it (decompressor.c:585-640, tamp/_c_decompressor.pyx:77-129): small inputs, small output buffers,
    state in the caller's 24-byte object and window buffer."""
    import ctypes as C

    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    lib = _lib.load()
    sz = C.POINTER(C.c_size_t)
    lib.tamp_decompressor_init.restype = C.c_int8
    lib.tamp_decompressor_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint8]
    lib.tamp_decompressor_decompress.restype = C.c_int8
    lib.tamp_decompressor_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_void_p, C.c_size_t, sz]
    rng = random.Random(5)
    text = wl.synth_text(1, 5000, first_index=3)[0].tobytes()
    runs = wl.lcg_runs(1, 3000, first_index=2)[0].tobytes()
    for plain, w in ((text, 10), (runs, 9), (bytes(900) + text[:600], 12)):
        st, blob = oracle.compress(plain, window=w)
        assert st == 0
        script = [(rng.choice([1, 3, 17, 200]), rng.choice([1, 5, 33, 400])) for _ in range(4000)]
        want0, want = oracle.decode_script(blob, script, window_bits=w)
        obj, window = (C.c_ubyte * 24)(), (C.c_ubyte * (1 << w))()
        assert lib.tamp_decompressor_init(obj, None, window, w) == want0 == 0
        pos, back = 0, bytearray()
        for (take, cap), (wst, wout, wcons) in zip(script, want):
            chunk = blob[pos : pos + take]
            out = (C.c_ubyte * max(cap, 1))()
            nw, nc = C.c_size_t(0), C.c_size_t(0)
            got = lib.tamp_decompressor_decompress(obj, out, cap, C.byref(nw), chunk, len(chunk), C.byref(nc))
            assert (got, bytes(out[: nw.value]), nc.value) == (wst, wout, wcons), (w, pos)
            pos += nc.value
            back += bytes(out[: nw.value])
            if pos == len(blob) and got == 2 and nw.value == 0:
                break
        assert bytes(back) == plain


def _encoder_golden(rec):
    conf = dict(rec["conf"])
    if conf.get("dictionary"):
        conf["dictionary"] = unb64(conf["dictionary"])
    ops = [[op[0]] + [unb64(a) if isinstance(a, str) else a for a in op[1:]] for op in rec["ops"]]
    want = [(r, unb64(out), k) for r, out, k in rec["calls"]]
    return conf, ops, want


def test_encoder_objects_golden_scripts(ta):
    """Compressor objects below flush granularity on the device (tamp_batch_compress_resume): sink / poll / compress /
    flush / compress_and_flush with small pieces and small output buffers, each call's status, bytes and consumed
    count as one reference TampCompressor returned them (tests/golden/encoder_resume.json)."""
    recs = load_golden("encoder_resume.json")
    for rec in recs:
        conf, ops, want = _encoder_golden(rec)
        enc = ta.EncoderBatch(1, **conf)
        got = []
        for op in ops:
            if op[0] == "sink":
                got.append((0, b"", int(enc.sink([op[1]])[0])))
                continue
            if op[0] == "poll":
                st, outs, cons = enc.poll([op[1]])
            elif op[0] == "flush":
                st, outs, cons = enc.flush([op[2]], bool(op[1]))
            elif op[0] == "compress":
                st, outs, cons = enc.compress([op[1]], [op[2]])
            else:
                st, outs, cons = enc.compress_and_flush([op[1]], [op[3]], bool(op[2]))
            got.append((int(st[0]), outs[0], int(cons[0])))
        assert got == want, rec["name"]


def test_reference_named_compressor_below_flush_granularity(ta):
    """tamp_compressor_sink / _full / _poll / _compress / _flush under the reference's names on a caller-allocated
    48-byte object (compressor.h:101-227), replaying the recorded scripts through the C symbols."""
    import ctypes as C

    from tamp_amd import _lib

    lib = _lib.load()

    class TampConf(C.Structure):
        _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                    ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1),
                    ("lazy_matching", C.c_uint16, 1)]

    sz = C.POINTER(C.c_size_t)
    protos = {
        "tamp_compressor_init": (C.c_int8, [C.c_void_p, C.c_void_p, C.c_void_p]),
        "tamp_compressor_sink": (None, [C.c_void_p, C.c_char_p, C.c_size_t, sz]),
        "tamp_compressor_full": (C.c_bool, [C.c_void_p]),
        "tamp_compressor_poll": (C.c_int8, [C.c_void_p, C.c_void_p, C.c_size_t, sz]),
        "tamp_compressor_compress": (C.c_int8, [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_char_p, C.c_size_t, sz]),
        "tamp_compressor_flush": (C.c_int8, [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_bool]),
        "tamp_compressor_compress_and_flush": (C.c_int8, [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_char_p, C.c_size_t,
                                                          sz, C.c_bool]),
    }
    for fn, (res, args) in protos.items():
        getattr(lib, fn).restype = res
        getattr(lib, fn).argtypes = args
    recs = {r["name"]: r for r in load_golden("encoder_resume.json")}
    for name in ("sink_poll_text", "sink_poll_runs_small_output", "compress_tiny_output", "flush_needs_two_bytes",
                 "lazy_pieces", "append_and_reset_conf", "custom_dictionary", "random_0", "random_1", "random_2"):
        conf, ops, want = _encoder_golden(recs[name])
        w = conf.get("window", 10)
        tc = TampConf(window=w, literal=conf.get("literal", 8), use_custom_dictionary=int("dictionary" in conf),
                      extended=int(conf.get("extended", True)), dictionary_reset=int(conf.get("dictionary_reset", False)),
                      append=int(conf.get("append", False)), lazy_matching=int(conf.get("lazy_matching", False)))
        obj, window = (C.c_ubyte * 48)(), (C.c_ubyte * (1 << w))()
        if "dictionary" in conf:
            C.memmove(window, conf["dictionary"], 1 << w)
        assert lib.tamp_compressor_init(obj, C.byref(tc), window) == 0
        for k, (op, (wst, wout, wcons)) in enumerate(zip(ops, want)):
            nw, nc = C.c_size_t(0), C.c_size_t(0)
            if op[0] == "sink":
                lib.tamp_compressor_sink(obj, op[1], len(op[1]), C.byref(nc))
                got = (0, b"", nc.value)
                assert lib.tamp_compressor_full(obj) == (bytes(obj)[8 + 7] == 16)  # input_size, compressor.h:27
            else:
                cap = op[-1]
                out = (C.c_ubyte * max(cap, 1))()
                if op[0] == "poll":
                    st = lib.tamp_compressor_poll(obj, out, cap, C.byref(nw))
                elif op[0] == "flush":
                    st = lib.tamp_compressor_flush(obj, out, cap, C.byref(nw), bool(op[1]))
                elif op[0] == "compress":
                    st = lib.tamp_compressor_compress(obj, out, cap, C.byref(nw), op[1], len(op[1]), C.byref(nc))
                else:
                    st = lib.tamp_compressor_compress_and_flush(obj, out, cap, C.byref(nw), op[1], len(op[1]), C.byref(nc),
                                                                bool(op[2]))
                got = (st, bytes(out[: nw.value]), nc.value)
            assert got == (wst, wout, wcons), (name, k, op[0])


def test_reference_named_object_mixes_token_level_and_segment_level_calls(ta, oracle):
    """One TampCompressor object driven through both device paths: pieces below flush granularity (token-by-token
    kernel, state in the object), then whole segments of several KiB (the batch kernel's segment mode, taken when the
    object is between segments) -- the emitted bytes are one stream, identical to the oracle's for the same writes and
    flush points, and the object's state stays coherent across the switch."""
    import ctypes as C

    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    lib = _lib.load()

    class TampConf(C.Structure):
        _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                    ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1),
                    ("lazy_matching", C.c_uint16, 1)]

    sz = C.POINTER(C.c_size_t)
    lib.tamp_compressor_init.restype = C.c_int8
    lib.tamp_compressor_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tamp_compressor_compress.restype = C.c_int8
    lib.tamp_compressor_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_char_p, C.c_size_t, sz]
    lib.tamp_compressor_compress_and_flush.restype = C.c_int8
    lib.tamp_compressor_compress_and_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_char_p, C.c_size_t, sz,
                                                       C.c_bool]
    text = wl.synth_text(1, 30000, first_index=5)[0].tobytes()
    runs = wl.lcg_runs(1, 9000, first_index=8)[0].tobytes()
    for lazy, dr in ((0, 0), (1, 1)):
        tc = TampConf(window=10, literal=8, extended=1, dictionary_reset=dr, lazy_matching=lazy)
        obj, window = (C.c_ubyte * 48)(), (C.c_ubyte * 1024)()
        assert lib.tamp_compressor_init(obj, C.byref(tc), window) == 0
        out = (C.c_ubyte * 70000)()
        emitted = bytearray()

        def call(fn, data, *extra):
            nw, nc = C.c_size_t(0), C.c_size_t(0)
            assert fn(obj, out, len(out), C.byref(nw), data, len(data), C.byref(nc), *extra) == 0
            assert nc.value == len(data)
            emitted.extend(bytes(out[: nw.value]))

        pieces = [text[:37], text[37:700], runs[:5000], text[700:9000], runs[5000:9000] + text[9000:20000], text[20000:20011]]
        call(lib.tamp_compressor_compress, pieces[0])                       # token level, ring left half full
        call(lib.tamp_compressor_compress_and_flush, pieces[1], False)      # token level (ring not empty, < 2 KiB)
        call(lib.tamp_compressor_compress_and_flush, pieces[2], True)       # segment level
        call(lib.tamp_compressor_compress_and_flush, pieces[3], True)       # segment level
        call(lib.tamp_compressor_compress, pieces[5])                       # token level again, 11 bytes stay in the ring
        call(lib.tamp_compressor_compress_and_flush, pieces[4], False)      # token level: the ring is not empty
        ops = [("write", pieces[0]), ("write", pieces[1]), ("flush", False), ("write", pieces[2]), ("flush", True),
               ("write", pieces[3]), ("flush", True), ("write", pieces[5]), ("write", pieces[4]), ("flush", False)]
        st, want = oracle.stream_script(ops, window=10, literal=8, extended=True, dictionary_reset=bool(dr),
                                        lazy_matching=bool(lazy))
        assert st == 0 and bytes(emitted) == want, (lazy, dr, len(emitted), len(want))
