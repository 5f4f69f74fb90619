#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the REFERENCE, in the build container only.

Two kinds of fixture are written:

* ``known_answers.json`` -- the known-answer vectors the reference's own tests hold for this
  path, restated as data (input / configuration / expected bytes) with the file:line they come
  from.  Each one is re-checked here against the reference C library before it is written.
* ``generated.json`` -- outputs of the reference C library itself (oracle/_ref/libtamp_ref.so,
  built in place from /root/reference by oracle/Makefile) on this repo's deterministic synthetic
  inputs (tamp_amd/workloads.py), one record per (workload, stream, configuration): SHA-256 of
  the input, the compressed bytes (base64) and the decode status/size.
* ``dictionaries.json`` -- tamp_initialize_dictionary for every (size, literal) pair.
* ``streaming.json`` -- op scripts (write / flush / reset_dictionary / close) replayed on one reference
  compressor object, with the bytes it emitted.
* ``decoder_resume.json`` -- call scripts (input chunk, output room) replayed on one reference decompressor
  object, with each call's status / bytes / consumed count.
* ``encoder_resume.json`` -- call scripts below flush granularity (sink / poll / compress / flush with the caller's
  output room) replayed on one reference compressor object, with each call's status / bytes / consumed count.
* ``device_vectors.json`` -- the reference's malformed/valid decoder vectors
  (devices/vectors/*.bin, data files its own tests replay) with the status/output the reference
  decoder produces for them.

Nothing under /root/reference is copied except those small data vectors; `/root/reference`
does not exist on the GPU box and no test reads it at run time.
"""
import base64
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.checker import Ref  # noqa: E402
from tamp_amd import workloads as wl  # noqa: E402

REFROOT = "/root/reference"


def b64(b: bytes) -> str:
    return base64.b64encode(b).decode()


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def dict_with(size: int, fill: int, patches) -> bytes:
    d = bytearray([fill]) * size
    for off, data in patches:
        d[off : off + len(data)] = data
    return bytes(d)


def known_answers(ref: Ref):
    foo = b"foo foo foo"
    foo_v1 = bytes.fromhex("58b3041c8100030000")
    comp = [
        # name, cite, conf, dictionary, input, expected
        ("foo_v1", "tests/test_compressor.py:66-105", dict(window=10, literal=8, extended=False), None, foo, foo_v1),
        ("foo_7bit", "tests/test_compressor.py:145-170", dict(window=10, literal=7, extended=False), None, foo,
         bytes([0b01010000, 0b11100110, 0b00001000, 0b00111010, 0b00000100, 0b00000000, 0b00001100, 0b00000000])),
        ("foo_predefined_dictionary", "tests/test_compressor.py:172-198", dict(window=8, literal=7, extended=False),
         dict_with(256, 0, [(0, foo)]), foo, bytes([0b00010100, 0b01010100, 0b00000000])),
        ("oob_2_byte_pattern", "tests/test_compressor.py:207-236", dict(window=10, literal=8, extended=False), None,
         b"Q\x00Q", bytes([0b01011000, 0b10101000, 0b11000000, 0b00101010, 0b00100000])),
        ("extended_rle_20", "tests/test_compressor.py:307-332", dict(window=10, literal=8, extended=True), None,
         b"A" * 20, bytes([0x5A, 0xA0, 0xAA, 0xB1])),
        ("extended_rle_5", "tests/test_compressor.py:334-356", dict(window=10, literal=8, extended=True), None,
         b"B" * 5, bytes([0x5A, 0xA1, 0x2A, 0x84])),
        ("extended_match_14", "tests/test_compressor.py:358-391", dict(window=8, literal=8, extended=True),
         dict_with(256, 0, [(0, b"abcdefghijklmn")]), b"abcdefghijklmn", bytes([0x1E, 0x4E, 0x00, 0x00])),
        ("extended_match_16", "tests/test_compressor.py:393-418", dict(window=8, literal=8, extended=True),
         dict_with(256, 0, [(0, b"abcdefghijklmnop")]), b"abcdefghijklmnop", bytes([0x1E, 0x4E, 0x40, 0x00])),
        ("finder_window_edge", "ctests/test_compressor.c:802-810", dict(window=8, literal=8, extended=False),
         dict_with(256, ord("a"), [(250, b"UVWXYZ")]), b"WXYZ!!", bytes([0x1C, 0x47, 0xE4, 0x86, 0x42])),
        ("finder_alignment_phases", "ctests/test_compressor.c:812-824", dict(window=8, literal=8, extended=False),
         dict_with(256, 0xFF, [(3, b"Qa"), (13, b"Qab"), (26, b"Qabc"), (40, b"Qabcd")]), b"Qabcd",
         bytes([0x1C, 0x59, 0x40])),
        ("finder_swar_bytes", "ctests/test_compressor.c:826-838", dict(window=8, literal=8, extended=False),
         dict_with(256, 0x01, [(10, b"\x00\x7f"), (50, b"\x80\x00\x80"), (100, b"\x00\x00\x00"),
                               (200, b"\x00\x80\x7f\x01\xff")]), b"\x00\x80\x7f\x01\xff", bytes([0x1C, 0x5E, 0x40])),
        ("finder_max_pattern_early_exit", "ctests/test_compressor.c:840-848",
         dict(window=8, literal=8, extended=False), dict_with(256, ord("z"), [(30, b"ABCDEFGHIJKLMNOP")]),
         b"ABCDEFGHIJKLMNOP", bytes([0x1C, 0x4E, 0x3D, 0x50])),
        ("default_is_extended", "BASELINE.md section 1 (measured from the reference C)",
         dict(window=10, literal=8, extended=True), None, foo, bytes.fromhex("5ab3041c8100030000")),
    ]
    payload = b"payload " * 20
    big = ref.initialize_dictionary(4096)
    decomp = [
        # name, cite, compressed, dictionary, expected output, expected status
        ("foo_v1", "tests/test_decompressor.py:40-67", foo_v1, None, foo, 2),
        ("flush_tokens", "tests/test_decompressor.py:96-113",
         bytes([0b01011000, 0b10101000, 0b10101010, 0b11000000, 0b10101011, 0b10101010, 0b11000000]), None, b"QW", 2),
        ("dst_immediately_after_src", "tests/test_decompressor.py:124-158",
         bytes([0b01011100, 0b10110000, 0b10110000, 0b00000000]), dict_with(1024, 0, [(0, b"abcd")]), b"aabc", 2),
        ("extended_rle_20", "tests/test_decompressor.py:193-211", bytes([0x5A, 0xA0, 0xAA, 0xB1]), None, b"A" * 20, 2),
        ("extended_rle_5", "tests/test_decompressor.py:213-230", bytes([0x5A, 0xA1, 0x2A, 0x84]), None, b"B" * 5, 2),
        ("extended_match_14", "tests/test_decompressor.py:232-254", bytes([0x1E, 0x4E, 0x00, 0x00]),
         dict_with(256, 0, [(0, b"abcdefghijklmn")]), b"abcdefghijklmn", 2),
        ("malicious_oob", "ctests/test_decompressor.c:74-98", bytes([0b01011000, 0b00111111, 0b11110000]), None, b"",
         -4),
        ("oversized_custom_dictionary", "tests/test_bug_regressions.py:177-190",
         bytes.fromhex("5eb8586f36c06cb248130009c8004f08004f320013c20000"), big, payload, 2),
        ("default_stream", "tests/test_bug_regressions.py:179,210-220",
         bytes.fromhex("5ab8586f36c06cb248130009c8004f08004f320013c20000"), None, payload, 2),
    ]
    out = {"compress": [], "decompress": []}
    for name, cite, conf, d, data, expected in comp:
        st, got = ref.compress(data, dictionary=d, **conf)
        assert st == 0 and got == expected, (name, st, got.hex(), expected.hex())
        out["compress"].append(dict(name=name, cite=cite, conf=conf, dictionary=b64(d) if d else None,
                                    input=b64(data), expected=expected.hex()))
    for name, cite, comp_bytes, d, expected, status in decomp:
        st, got, _ = ref.decompress(comp_bytes, dictionary=d, cap=4096)
        assert st == status and got == expected, (name, st, got, expected)
        out["decompress"].append(dict(name=name, cite=cite, compressed=comp_bytes.hex(),
                                      dictionary=b64(d) if d else None, expected=b64(expected), status=status))
    return out


def dictionaries(ref: Ref):
    expected_256 = (
        b"\x00.//r.0. t>\n/>snas.trnr i\x00r/a\x00snat./.r\x00i o.s tneo>.as>\na.ta\x00 aa\x00\x00\x000oe ri\x00a>eatsi\n.\ni.str\n"
        b"//snesr.ost<  \x00\ni\neoa\x00se0.o\n\n>aori>n0.>./.oonen0<\x00<r o\n\naas0< ai\n0\x00na\x00e><.\noas to \n></se>>ts/"
        b"oreatinter.n0 >s\n/.e.><. r si<>/<san\x00ae t 0.r.o/0./a r/ttn nn.<re.t0 \x00r\x00ro"
    )  # tests/test_pseudorandom.py:22-24
    assert ref.initialize_dictionary(256, 8) == expected_256
    recs = []
    for wbits in range(8, 16):
        for lit in (5, 6, 7, 8):
            d = ref.initialize_dictionary(1 << wbits, lit)
            recs.append(dict(size=1 << wbits, literal=lit, sha256=sha(d), head=d[:32].hex()))
    return dict(cite="tests/test_pseudorandom.py:22-24; common.c:18-52", first256_literal8=expected_256.hex(),
                table=recs)


def generated(ref: Ref):
    cases = []

    def add(workload, rows, confs, dictionary=None, dict_name=None):
        for i, row in enumerate(rows):
            data = row.tobytes()
            for conf in confs:
                st, comp = ref.compress(data, dictionary=dictionary, **conf)
                dst, dout, _ = ref.decompress(comp, dictionary=dictionary, cap=len(data) + 64) if st == 0 else (None, b"", 0)
                if st == 0:
                    assert dst == 2 and dout == data
                cases.append(dict(workload=workload, index=i, length=len(data), input_sha256=sha(data), conf=conf,
                                  dictionary=dict_name, status=st, compressed=b64(comp), compressed_sha256=sha(comp)))

    base = [dict(window=10, literal=8, extended=False), dict(window=10, literal=8, extended=True)]
    add("synth_text:4096", wl.synth_text(24, 4096), base)
    add("synth_text:4096", wl.synth_text(4, 4096), [dict(window=w, literal=8, extended=True) for w in (8, 9, 11, 12)] +
        [dict(window=12, literal=8, extended=False), dict(window=15, literal=8, extended=True),
         dict(window=10, literal=8, extended=True, lazy_matching=True),
         dict(window=10, literal=8, extended=False, lazy_matching=True)])
    add("synth_text:65536", wl.synth_text(1, 65536), base)
    add("synth_text:777", wl.synth_text(6, 777), base)
    add("lcg_runs:512", wl.lcg_runs(24, 512), base + [dict(window=8, literal=8, extended=True)])
    add("stress:8192", wl.stress(6, 8192), [dict(window=w, literal=8, extended=e) for w in (8, 10, 12) for e in (False, True)])
    for n in (0, 1, 2, 3, 15, 16, 17, 31, 32, 33):
        add(f"synth_text:{n}", wl.synth_text(2, n), base)
    tel_dict = wl.telemetry_dictionary(ref.initialize_dictionary(256, 7))
    add("telemetry:256", wl.telemetry(24, 256), [dict(window=8, literal=7, extended=True)], tel_dict, "telemetry")
    add("telemetry:256", wl.telemetry(8, 256), [dict(window=8, literal=7, extended=False),
                                                dict(window=8, literal=7, extended=True)])
    # EXCESS_BITS: a byte >= 0x80 under literal=7
    bad = wl.telemetry(4, 256).copy()
    bad[1, 40] = 0xC3
    bad[2, 0] = 0x80
    bad[3, 255] = 0xFF
    add("telemetry_bad:256", bad, [dict(window=8, literal=7, extended=True)], tel_dict, "telemetry")
    # 5/6-bit literals on masked text (dictionary tables for small literals; min_pattern 3 at w>=11..)
    txt = wl.synth_text(3, 1500)
    add("synth_text&31:1500", txt & 31, [dict(window=w, literal=5, extended=e) for w in (8, 10, 11, 13) for e in (False, True)])
    add("synth_text&63:1500", txt & 63, [dict(window=w, literal=6, extended=True) for w in (10, 12, 13)])
    return dict(note="outputs of oracle/_ref/libtamp_ref.so (reference C, -O3, TAMP_LAZY_MATCHING=1)",
                telemetry_dictionary=b64(tel_dict), cases=cases)


def device_vectors(ref: Ref):
    recs = []
    vdir = os.path.join(REFROOT, "devices", "vectors")
    for name in sorted(os.listdir(vdir)):
        data = open(os.path.join(vdir, name), "rb").read()
        st, out, consumed = ref.decompress(data, cap=1 << 16)
        recs.append(dict(name=name, cite=f"devices/vectors/{name} (replayed by devices/common/tamp_bench.c:228-273)",
                         data=b64(data), status=st, output=b64(out), consumed=consumed))
    return recs


def streaming(ref: Ref):
    """Op scripts on ONE reference compressor object: the shapes of tests/test_compressor_decompressor.py:312-556
    (reset_dictionary_*, append_mode_roundtrip, double_flush_does_not_reset) on this repo's synthetic text, plus
    seeded random scripts.  ``expected`` is what the reference wrote; ``decodes`` says whether the reference decoder
    turns it back into the concatenated writes (not the case when a write_token=False flush pads mid-stream)."""
    import random

    text = bytes(wl.synth_text(1, 6000, first_index=77)[0])
    runs = bytes(wl.lcg_runs(1, 3000, first_index=5)[0])
    hello, bye = b"Hello world! " * 20, b"Goodbye world! " * 20
    W, F, R, CL = (lambda b: ("write", b)), (lambda t: ("flush", t)), ("reset",), ("close",)
    scripts = [
        ("reset_dictionary_basic", ":312-327", dict(dictionary_reset=True),
         [W(text[:200]), F(True), R, W(text[200:3000]), F(False)]),
        ("reset_dictionary_no_prior_flush", ":329-345", dict(dictionary_reset=True), [W(hello), R, W(bye), F(False)]),
        ("reset_dictionary_multiple", ":347-366", dict(dictionary_reset=True),
         [W(text[:500]), R, W(text[500:1000]), R, W(text[1000:1500]), F(False)]),
        ("reset_dictionary_immediate", ":368-382", dict(dictionary_reset=True), [R, W(text[:800]), F(False)]),
        ("reset_dictionary_v1", ":384-399", dict(dictionary_reset=True, extended=False),
         [W(text[:300]), F(True), R, W(text[300:900]), F(False)]),
        ("reset_dictionary_lazy", ":401-418", dict(dictionary_reset=True, lazy_matching=True),
         [W(text[:300]), F(True), R, W(text[300:900]), F(False)]),
        ("reset_dictionary_small_window", ":420-436", dict(dictionary_reset=True, window=8),
         [W(text[:700]), R, W(text[700:1400]), F(False)]),
        ("reset_dictionary_literal7", ":438-456", dict(dictionary_reset=True, literal=7),
         [W(text[:700]), R, W(text[700:1400]), F(False)]),
        ("reset_dictionary_rle_boundary", ":458-474", dict(dictionary_reset=True),
         [W(b"A" * 100), R, W(b"A" * 100 + b"B" * 50), F(False)]),
        ("append_session_1", ":510-530", dict(dictionary_reset=True), [W(hello[:39]), CL]),
        ("append_session_2", ":510-530", dict(dictionary_reset=True, append=True), [W(bye[:45]), CL]),
        ("double_flush_does_not_reset", ":532-555", dict(dictionary_reset=True),
         [W(hello), F(True), F(True), F(True), W(bye), F(False)]),
        ("flush_tokens_no_reset_conf", "tests/test_compressor.py:241-246 (flush() default)", dict(),
         [W(text[:100]), F(True), W(text[100:117]), F(True), F(True), W(text[117:1000]), CL]),
        ("split_writes_equal_one_shot", "tamp/_c_compressor.pyx:70-116", dict(),
         [W(text[i : i + 37]) for i in range(0, 3000, 37)] + [F(False)]),
        ("rle_across_flush", "compressor.c:745-762", dict(), [W(runs[:777]), F(True), W(runs[777:2000]), F(True), CL]),
        ("empty_stream_close", "compressor.c:784", dict(), [CL]),
        ("empty_stream_close_reset_conf", "compressor.c:784", dict(dictionary_reset=True), [CL]),
        ("custom_dictionary_then_reset", "compressor.c:871-873", dict(dictionary_reset=True, window=9,
                                                                      dictionary=text[4000:4512]),
         [W(text[4100:4600]), R, W(text[4100:4600]), CL]),
    ]
    rng = random.Random(20260928)
    srcs = [text, runs, bytes(wl.stress(3, 4096)[1]), bytes(wl.stress(3, 4096)[2])]
    for k in range(48):
        src = rng.choice(srcs)
        dr = rng.random() < 0.5
        conf = dict(window=rng.choice([8, 9, 10, 11, 12, 15]), literal=8, extended=rng.random() < 0.7,
                    dictionary_reset=dr, append=dr and rng.random() < 0.2, lazy_matching=rng.random() < 0.25)
        pos, ops = rng.randrange(0, 1000), []
        for _ in range(rng.randrange(2, 9)):
            x = rng.random()
            if x < 0.55:
                n = rng.choice([0, 1, 2, 3, 15, 16, 17, 31, 40, 100, 300, 700])
                ops.append(W(src[pos : pos + n]))
                pos += n
            elif x < 0.85:
                ops.append(F(rng.random() < 0.7))
            elif dr:
                ops.append(R)
        ops.append(CL)
        scripts.append((f"random_{k:02d}", "seeded script", conf, ops))
    recs = []
    for name, cite, conf, ops in scripts:
        if cite.startswith(":"):
            cite = "tests/test_compressor_decompressor.py" + cite
        st, got = ref.stream_script(ops, **conf)
        plain = b"".join(op[1] for op in ops if op[0] == "write")
        decodes = False
        if st == 0 and not conf.get("append"):
            dst, back, _ = ref.decompress(got, dictionary=conf.get("dictionary"), cap=len(plain) + 64)
            decodes = dst == 2 and back == plain
        jconf = {k: (b64(v) if k == "dictionary" else v) for k, v in conf.items()}
        recs.append(dict(name=name, cite=cite, conf=jconf, status=st, expected=b64(got), decodes=decodes,
                         plain_sha256=sha(plain),
                         ops=[[op[0], b64(op[1])] if op[0] == "write" else list(op) for op in ops]))
    s1 = next(r for r in recs if r["name"] == "append_session_1")
    s2 = next(r for r in recs if r["name"] == "append_session_2")
    cat = base64.b64decode(s1["expected"]) + base64.b64decode(s2["expected"])
    dst, back, _ = ref.decompress(cat, cap=4096)
    assert dst == 2 and back == hello[:39] + bye[:45], (dst, back)
    assert sum(r["decodes"] for r in recs) > 20
    return recs


def decoder_resume(ref: Ref):
    """Call scripts on ONE reference decompressor object (decompressor.c:371-578): each call gets the next `take`
    unconsumed bytes and `cap` bytes of output room -- the output-full / input-exhausted resume shapes of
    tests/test_decompressor.py:99-144, ctests/test_decompressor.c:105-144 and tests/test_compressor_decompressor.py
    (flush / reset streams) on this repo's synthetic inputs, plus seeded random scripts.  Recorded per call:
    status, bytes written (base64), bytes consumed."""
    import random

    text = bytes(wl.synth_text(1, 3000, first_index=91)[0])
    runs = bytes(wl.lcg_runs(1, 2500, first_index=9)[0])
    zeros = bytes(700) + b"abcabcabcabc" * 30 + bytes(300)
    W, F, R, CL = (lambda b: ("write", b)), (lambda t: ("flush", t)), ("reset",), ("close",)

    def stream(plain, **kw):
        st, blob = ref.stream_script([W(plain), CL], **kw)
        assert st == 0
        return blob

    sources = {
        "text_w10": (stream(text), {}),
        "text_v1_w9": (stream(text, window=9, extended=False), {}),
        "runs_w10": (stream(runs), {}),
        "runs_w8": (stream(runs, window=8), {}),
        "zeros_w10": (stream(zeros), {}),
        "text_l7_w12": (stream(bytes(b & 127 for b in text), window=12, literal=7), {}),
    }
    st, blob = ref.stream_script([W(text[:500]), F(True), W(runs[:400]), R, W(zeros[:800]), F(True), F(True), W(text[500:900]), CL],
                                 dictionary_reset=True)
    assert st == 0
    sources["flush_and_reset"] = (blob, {})
    dic = (text[1000:1256] * 8)[:1024]
    st, blob = ref.stream_script([W(text[:1500]), CL], dictionary=dic)
    assert st == 0
    sources["custom_dictionary"] = (blob, {"dictionary": dic})

    fixed = [
        ("one_byte_at_a_time", "text_w10", [(1, 4096)] * 1700),
        ("output_one_byte", "runs_w10", [(4096, 1)] * 2600),
        ("output_full_resume_small", "zeros_w10", [(4096, 7)] * 300),
        ("output_full_resume_16", "text_v1_w9", [(4096, 16)] * 200),
        ("both_tiny", "runs_w8", [(2, 3)] * 1500),
        ("zero_room_calls", "text_w10", [(5, 0), (5, 10), (0, 10), (5, 0)] * 50),
        ("whole_then_empty_calls", "flush_and_reset", [(1 << 20, 1 << 16), (0, 16), (0, 0)]),
        ("flush_reset_small_calls", "flush_and_reset", [(3, 5)] * 1500),
        ("custom_dictionary_chunks", "custom_dictionary", [(50, 64)] * 200),
        ("literal7_chunks", "text_l7_w12", [(9, 33)] * 400),
    ]
    recs = []

    def record(name, src, script, conf=None, window_bits=15, mutate=None):
        blob, kw = sources[src]
        data = blob
        if mutate is not None:
            b = bytearray(blob)
            b[mutate[0] % len(b)] ^= 1 << mutate[1]
            data = bytes(b)
        if conf is not None:
            data = data[1 + (data[0] & 1) :]
        r0, calls = ref.decode_script(data, script, conf=conf, window_bits=window_bits, dictionary=kw.get("dictionary"))
        # drop the tail of calls that neither consume nor produce (keeps the file small)
        while len(calls) > 1 and calls[-1][1] == b"" and calls[-1][2] == 0 and calls[-2][1] == b"" and calls[-2][2] == 0:
            calls.pop()
        script = script[: len(calls)]
        recs.append(dict(name=name, source=src, data=b64(data), conf=list(conf) if conf is not None else None,
                         window_bits=window_bits, dictionary=b64(kw["dictionary"]) if "dictionary" in kw else None,
                         script=[list(s) for s in script], init=r0,
                         calls=[[r, b64(out), k] for r, out, k in calls]))

    for name, src, script in fixed:
        record(name, src, script)
    record("conf_given_skip_header", "text_w10", [(7, 50)] * 400, conf=(10, 8, False, True, False))
    record("window_bits_too_small", "text_l7_w12", [(10, 10)] * 3, window_bits=10)
    record("stashed_header_byte", "flush_and_reset", [(1, 10), (0, 10), (1, 10), (100, 10)])
    rng = random.Random(20260928)
    names = sorted(sources)
    for k in range(40):
        src = rng.choice(names)
        script = [(rng.choice([0, 1, 2, 3, 5, 9, 40, 1000]), rng.choice([0, 1, 2, 3, 7, 20, 64, 300])) for _ in range(rng.randrange(5, 400))]
        script.append((1 << 20, 1 << 15))
        mutate = (rng.randrange(1, 4000), rng.randrange(8)) if rng.random() < 0.3 else None
        record(f"random_{k}", src, script, mutate=mutate, window_bits=rng.choice([15, 15, 12]))
    return recs


def encoder_resume(ref: Ref):
    """Call scripts on ONE reference compressor object below flush granularity: tamp_compressor_sink / _poll /
    _compress / _flush / _compress_and_flush with the caller's output room (compressor.c:532-810) -- the sink/poll
    loops of ctests/test_compressor.c and docs/source/c_library.rst, output buffers that fill up, pieces of any size --
    on this repo's synthetic inputs, plus seeded random scripts.  Recorded per call: status, bytes, consumed."""
    import random

    text = bytes(wl.synth_text(1, 2500, first_index=123)[0])
    runs = bytes(wl.lcg_runs(1, 2000, first_index=11)[0])
    zeros = bytes(600) + b"xyzxyzxyzxyz" * 25 + bytes(200)
    recs = []

    def record(name, ops, **kw):
        r0, calls = ref.encode_script(ops, **kw)
        enc = []
        for op in ops:
            enc.append([op[0]] + [b64(a) if isinstance(a, (bytes, bytearray)) else a for a in op[1:]])
        d = kw.get("dictionary")
        recs.append(dict(name=name, conf={k: (b64(v) if k == "dictionary" and v is not None else v) for k, v in kw.items()},
                         ops=enc, init=r0, calls=[[r, b64(out), k] for r, out, k in calls]))

    def sink_poll(data, cap):  # the loop of docs/source/c_library.rst: sink what fits, poll when full, flush at the end
        ops, pos = [], 0
        while pos < len(data):
            ops.append(("sink", data[pos : pos + 16]))
            pos += 16  # (what the ring does not take is dropped: the scripts only need determinism)
            ops.append(("poll", cap))
        ops.append(("flush", False, 64))
        return ops

    record("sink_poll_text", sink_poll(text[:600], 32))
    record("sink_poll_runs_small_output", sink_poll(runs[:500], 3))
    record("compress_pieces_then_flush", [("compress", text[i : i + 37], 64) for i in range(0, 1500, 37)] + [("flush", True, 16)])
    record("compress_one_byte_pieces", [("compress", runs[i : i + 1], 16) for i in range(400)] + [("flush", False, 16)])
    record("compress_tiny_output", [("compress", zeros, 2)] * 40 + [("flush", True, 1)] * 6 + [("flush", True, 8)])
    record("flush_needs_two_bytes", [("compress", text[:100], 200), ("flush", True, 1), ("flush", True, 0), ("flush", True, 2)])
    record("extended_match_needs_six_bytes", [("compress", zeros[600:900] * 2, 5)] * 30 + [("flush", False, 64)])
    record("compress_and_flush_small_output", [("compress_and_flush", text[:300], True, 40)] * 3 + [("flush", True, 400)])
    record("lazy_pieces", [("compress", text[i : i + 23], 64) for i in range(0, 900, 23)] + [("flush", False, 16)],
           lazy_matching=True)
    record("v1_pieces_w8", [("compress", runs[i : i + 50], 64) for i in range(0, 1000, 50)] + [("flush", True, 16)],
           window=8, extended=False)
    record("literal7_excess_bits", [("compress", bytes(b & 127 for b in text[:100]), 64), ("compress", b"ab\xffcd" * 8, 64),
                                    ("flush", False, 64)], literal=7)
    record("append_and_reset_conf", [("compress", text[:200], 64), ("flush", True, 64), ("flush", True, 64),
                                     ("compress", text[200:260], 64), ("flush", True, 64)], dictionary_reset=True, append=True)
    dic = (text[500:756] * 4)[:1024]
    record("custom_dictionary", [("compress", text[:700], 64)] * 1 + [("flush", False, 64)], dictionary=dic)
    rng = random.Random(4711)
    srcs = [text, runs, zeros]
    for k in range(36):
        kw = dict(window=rng.choice([8, 9, 10, 10, 12]), literal=8, extended=rng.random() < 0.75,
                  lazy_matching=rng.random() < 0.3, dictionary_reset=rng.random() < 0.3)
        src = rng.choice(srcs)
        ops, pos = [], 0
        for _ in range(rng.randrange(5, 80)):
            kind = rng.choice(["compress", "compress", "compress", "poll", "sink", "flush", "compress_and_flush"])
            cap = rng.choice([0, 1, 2, 3, 5, 6, 8, 20]) if rng.random() < 0.3 else rng.choice([64, 300])
            n = rng.choice([0, 1, 3, 15, 16, 17, 40, 200])
            piece = src[pos : pos + n]
            pos = (pos + n) % max(1, len(src) - 200)
            if kind == "compress":
                ops.append(("compress", piece, cap))
            elif kind == "poll":
                ops.append(("poll", cap))
            elif kind == "sink":
                ops.append(("sink", piece))
            elif kind == "flush":
                ops.append(("flush", rng.random() < 0.6, cap))
            else:
                ops.append(("compress_and_flush", piece, rng.random() < 0.6, cap))
        ops.append(("flush", True, 400))
        record(f"random_{k}", ops, **kw)
    return recs


def main():
    ref = Ref()
    assert ref.sizes() == (2, 48, 24), ref.sizes()
    for fname, obj in (
        ("known_answers.json", known_answers(ref)),
        ("dictionaries.json", dictionaries(ref)),
        ("generated.json", generated(ref)),
        ("device_vectors.json", device_vectors(ref)),
        ("streaming.json", streaming(ref)),
        ("decoder_resume.json", decoder_resume(ref)),
        ("encoder_resume.json", encoder_resume(ref)),
    ):
        with open(os.path.join(HERE, fname), "w") as f:
            json.dump(obj, f, indent=0, sort_keys=True)
        print(fname, os.path.getsize(os.path.join(HERE, fname)))


if __name__ == "__main__":
    main()
