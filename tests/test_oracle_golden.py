"""CPU tier: the oracle (oracle/tamp_oracle.c) against every golden vector.

This is what pins the oracle: the reference's own known-answer vectors, its dictionary bytes,
its decoder robustness vectors, and outputs of the reference C library generated in the build
container (tests/golden/make_golden.py).
"""
import hashlib

import numpy as np
import pytest
from conftest import load_golden, unb64, workload_rows


def test_known_answer_compress(oracle):
    ka = load_golden("known_answers.json")
    assert len(ka["compress"]) >= 13
    for c in ka["compress"]:
        st, got = oracle.compress(unb64(c["input"]), dictionary=unb64(c["dictionary"]), **c["conf"])
        assert st == 0, c["name"]
        assert got.hex() == c["expected"], (c["name"], c["cite"])


def test_known_answer_decompress(oracle):
    ka = load_golden("known_answers.json")
    for c in ka["decompress"]:
        st, got, _ = oracle.decompress(bytes.fromhex(c["compressed"]), dictionary=unb64(c["dictionary"]), cap=4096)
        assert st == c["status"], (c["name"], c["cite"])
        assert got == unb64(c["expected"]), c["name"]


def test_restricted_output_resume_points(oracle):
    """tests/test_decompressor.py:69-94,256-275: a short output buffer stops with OUTPUT_FULL at the
    requested size; the bytes produced are the prefix of the full output."""
    ka = load_golden("known_answers.json")
    for c in ka["decompress"]:
        if c["status"] != 2:
            continue
        full = unb64(c["expected"])
        for cap in range(0, len(full)):
            st, got, _ = oracle.decompress(bytes.fromhex(c["compressed"]), dictionary=unb64(c["dictionary"]), cap=cap)
            assert st == 1 and got == full[:cap], (c["name"], cap)


def test_dictionaries(oracle):
    d = load_golden("dictionaries.json")
    assert oracle.initialize_dictionary(256, 8).hex() == d["first256_literal8"]
    for rec in d["table"]:
        got = oracle.initialize_dictionary(rec["size"], rec["literal"])
        assert hashlib.sha256(got).hexdigest() == rec["sha256"], rec
        assert got[:32].hex() == rec["head"]


def test_min_pattern_size(oracle):
    # common.c:54-56
    for w in range(8, 16):
        for lit in range(5, 9):
            assert oracle.min_pattern_size(w, lit) == 2 + (w > 10 + 2 * (lit - 5))


def test_device_vectors(oracle):
    for v in load_golden("device_vectors.json"):
        st, out, consumed = oracle.decompress(unb64(v["data"]), cap=1 << 16)
        assert (st, out, consumed) == (v["status"], unb64(v["output"]), v["consumed"]), v["name"]


def test_generated_reference_outputs(oracle):
    g = load_golden("generated.json")
    tel = unb64(g["telemetry_dictionary"])
    rows_cache = {}
    n_checked = 0
    for c in g["cases"]:
        wlname = c["workload"]
        if wlname not in rows_cache:
            need = 1 + max(k["index"] for k in g["cases"] if k["workload"] == wlname)
            rows_cache[wlname] = workload_rows(wlname)(need)
        data = rows_cache[wlname][c["index"]].tobytes()
        assert hashlib.sha256(data).hexdigest() == c["input_sha256"], ("generator drifted", wlname, c["index"])
        d = tel if c["dictionary"] == "telemetry" else None
        st, comp = oracle.compress(data, dictionary=d, **c["conf"])
        assert st == c["status"], (wlname, c["index"], c["conf"])
        assert comp == unb64(c["compressed"]), (wlname, c["index"], c["conf"])
        if st == 0:
            dst, out, _ = oracle.decompress(comp, dictionary=d, cap=len(data) + 8)
            assert dst == 2 and out == data
        n_checked += 1
    assert n_checked > 250


def _script(rec):
    conf = dict(rec["conf"])
    if "dictionary" in conf:
        conf["dictionary"] = unb64(conf["dictionary"])
    ops = [("write", unb64(op[1])) if op[0] == "write" else tuple(op) for op in rec["ops"]]
    return conf, ops


def test_streaming_scripts(oracle):
    """write / flush(write_token) / reset_dictionary / close sequences, bytes emitted by ONE reference object
    (tests/golden/streaming.json; shapes of tests/test_compressor_decompressor.py:312-556)."""
    recs = load_golden("streaming.json")
    assert len(recs) >= 60
    for rec in recs:
        conf, ops = _script(rec)
        st, got = oracle.stream_script(ops, **conf)
        assert st == rec["status"], rec["name"]
        assert got == unb64(rec["expected"]), (rec["name"], rec["cite"])
        if rec["decodes"]:
            dst, back, _ = oracle.decompress(got, dictionary=conf.get("dictionary"), cap=1 << 16)
            assert dst == 2 and hashlib.sha256(back).hexdigest() == rec["plain_sha256"], rec["name"]


def _resume_record(rec):
    conf = tuple(rec["conf"]) if rec["conf"] is not None else None
    dic = unb64(rec["dictionary"]) if rec["dictionary"] else None
    script = [tuple(s) for s in rec["script"]]
    want = [(r, unb64(out), k) for r, out, k in rec["calls"]]
    return unb64(rec["data"]), script, conf, dic, want


def test_decoder_resume_scripts(oracle):
    """One decompressor object driven call by call with small inputs and small output room: status, bytes and
    consumed count of every call as recorded from the reference (tests/golden/decoder_resume.json; shapes of
    tests/test_decompressor.py:99-144, ctests/test_decompressor.c:105-144)."""
    recs = load_golden("decoder_resume.json")
    assert len(recs) >= 50
    recs = recs + load_golden("decoder_resume_fuzz.json")  # objects a randomised GPU run once got wrong (calls after an error)
    for rec in recs:
        data, script, conf, dic, want = _resume_record(rec)
        r0, calls = oracle.decode_script(data, script, conf=conf, window_bits=rec["window_bits"], dictionary=dic)
        assert r0 == rec["init"], rec["name"]
        assert calls == want, rec["name"]


def test_encoder_resume_golden_is_consistent_with_the_oracle(oracle):
    """tests/golden/encoder_resume.json (calls on one reference compressor object below flush granularity): whatever
    the chunking and the output room, the bytes a script emitted are one valid stream -- the oracle's decoder turns
    them back into exactly the input the calls consumed."""
    recs = load_golden("encoder_resume.json")
    assert len(recs) >= 45
    checked = 0
    for rec in recs:
        assert rec["init"] == 0
        conf = dict(rec["conf"])
        dic = unb64(conf["dictionary"]) if conf.get("dictionary") else None
        if conf.get("append"):
            continue  # starts with the append marker instead of a header: not a stream of its own
        emitted, plain, ok, full_pending = bytearray(), bytearray(), True, False
        for op, (st, out, consumed) in zip(rec["ops"], rec["calls"]):
            emitted += unb64(out)
            if op[0] in ("compress", "compress_and_flush", "sink"):
                plain += unb64(op[1])[:consumed]
            ok = ok and st >= 0
            full_pending = st == 1
        if not ok or full_pending or rec["ops"][-1][0] != "flush":
            continue
        # a flush without the FLUSH token pads the last byte with zero bits: fine at the end, not mid-stream
        mid = rec["ops"][:-1]
        if any((op[0] == "flush" and not op[1]) or (op[0] == "compress_and_flush" and not op[2]) for op in mid):
            continue
        dst, back, _ = oracle.decompress(bytes(emitted), dictionary=dic, cap=1 << 16)
        assert dst == 2 and back == bytes(plain), rec["name"]
        checked += 1
    assert checked >= 10


def test_invalid_conf(oracle):
    # compressor.c:208-209, tests/test_compressor.py:420-433
    assert oracle.compress(b"x", window=7)[0] == -3
    assert oracle.compress(b"x", window=16)[0] == -3
    assert oracle.compress(b"x", literal=4)[0] == -3
    assert oracle.compress(b"x", literal=9)[0] == -3
    # decompressor.c:284 second header byte must be zero; :311 window above the buffer's maximum
    assert oracle.decompress(bytes([0x59, 0x01, 0x00]))[0] == -3
    assert oracle.decompress(bytes([0x58 | 0xE0, 0x00]), max_window_bits=10)[0] == -3


def test_excess_bits(oracle):
    # compressor.c:629-631, tests/test_compressor.py:238-246
    st, out = oracle.compress(b"\xff", literal=7, extended=False)
    assert st == -2 and out == bytes([0x50])
    st, _ = oracle.compress(b"abc\x80", literal=7, extended=True)
    assert st == -2


def test_empty_and_tiny(oracle):
    assert oracle.compress(b"", extended=False) == (0, bytes([0x58]))
    assert oracle.compress(b"")[1] == bytes([0x5A])
    assert oracle.decompress(b"") == (2, b"", 0)
    assert oracle.decompress(bytes([0x5A])) == (2, b"", 1)
    for n in range(1, 40):
        data = bytes((7 * i + n) & 0xFF for i in range(n))
        for ext in (False, True):
            st, comp = oracle.compress(data, extended=ext)
            assert st == 0
            assert oracle.decompress(comp, cap=n + 4)[:2] == (2, data)


def test_worst_case_size_bound(oracle):
    from oracle.checker import worst_case_compressed_size

    rng = np.random.default_rng(1)
    for lit in (5, 6, 7, 8):
        data = rng.integers(0, 1 << lit, 4096, dtype=np.uint8).tobytes()
        for ext in (False, True):
            st, comp = oracle.compress(data, literal=lit, extended=ext)
            assert st == 0 and len(comp) <= worst_case_compressed_size(4096, lit)
