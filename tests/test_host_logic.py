"""CPU tier: host-side logic of the Python surface (no device): helpers, packing, sharding, argument checks."""
import io

import numpy as np
import pytest

import tamp_amd
from tamp_amd import batch, sharding
from conftest import load_golden


def test_bit_size_and_min_pattern():
    # tamp/__init__.py:18-23,66-70
    assert [tamp_amd.bit_size(v) for v in (0, 1, 2, 255, 256, (1 << 31) - 1)] == [0, 1, 2, 8, 9, 31]
    assert tamp_amd.bit_size(1 << 31) == -1  # the reference's loop stops at 32 iterations
    assert tamp_amd.compute_min_pattern_size(10, 8) == 2
    assert tamp_amd.compute_min_pattern_size(12, 5) == 3
    with pytest.raises(ValueError):
        tamp_amd.compute_min_pattern_size(16, 8)
    with pytest.raises(ValueError):
        tamp_amd.compute_min_pattern_size(10, 4)


def test_initialize_dictionary_surface():
    d = load_golden("dictionaries.json")
    import os

    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtamp_amd.so not built")
    out = tamp_amd.initialize_dictionary(256)  # tests/test_pseudorandom.py:22-24
    assert isinstance(out, bytearray) and out.hex() == d["first256_literal8"]
    buf = bytearray(256)
    assert tamp_amd.initialize_dictionary(buf) is buf and buf.hex() == d["first256_literal8"]
    assert tamp_amd.initialize_dictionary(256, seed=0) == bytearray(256)
    assert tamp_amd.initialize_dictionary(256, seed=1) != bytearray(256)
    assert tamp_amd.initialize_dictionary(256, seed=3758097560).hex() == d["first256_literal8"]
    with pytest.raises(ValueError):
        tamp_amd.initialize_dictionary(256, literal=4)


def test_compress_bound_and_packing():
    assert tamp_amd.compress_bound(4096, 8) == 4609
    assert tamp_amd.compress_bound(0, 8) == 1
    assert tamp_amd.compress_bound(256, 7, dictionary_reset=True) == 2 + 256
    flat, off, ln = tamp_amd.pack_streams([b"abc", b"", b"defgh"])
    assert flat.tobytes() == b"abcdefgh" and off.tolist() == [0, 3, 3] and ln.tolist() == [3, 0, 5]
    offs, total = batch._slab_offsets(np.array([5, 0, 7], dtype=np.uint32))
    assert offs.tolist() == [0, 5, 5] and total == 12


def test_surface_argument_errors_need_no_device():
    import os

    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtamp_amd.so not built")
    with pytest.raises(ValueError):  # tests/test_compressor.py:196-205
        tamp_amd.Compressor(io.BytesIO(), window=9, literal=7, dictionary=bytearray(256))
    with pytest.raises(ValueError):  # tests/test_compressor.py:420-433
        tamp_amd.Compressor(io.BytesIO(), literal=4)
    with pytest.raises(ValueError):
        tamp_amd.Compressor(io.BytesIO(), window=16)
    with pytest.raises(ValueError):
        tamp_amd.open(io.BytesIO(), "rw")
    with pytest.raises(ValueError):  # append needs dictionary_reset and no custom dictionary (compressor.c:209)
        tamp_amd.Compressor(io.BytesIO(), append=True)
    with pytest.raises(ValueError):
        tamp_amd.Compressor(io.BytesIO(), dictionary_reset=True, append=True, dictionary=bytearray(1024))
    with pytest.raises(ValueError):  # tamp_compressor_reset_dictionary -> TAMP_INVALID_CONF (compressor.c:846)
        tamp_amd.Compressor(io.BytesIO()).reset_dictionary()
    with pytest.raises(ValueError):
        tamp_amd.compress_batch([b"x"], window=8, dictionary=bytes(100))


def test_partition_streams():
    assert sharding.partition_streams([4096] * 8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert sharding.partition_streams([4096] * 10, 4) == [(0, 3), (3, 5), (5, 8), (8, 10)]
    parts = sharding.partition_streams([1, 1, 1000, 1, 1, 1], 3)
    assert parts[0][0] == 0 and parts[-1][1] == 6 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert sharding.partition_streams([], 3) == [(0, 0)] * 3
    assert sharding.partition_streams([5, 5], 4)[-1][1] == 2
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 5000, 1000)
    for ws in (1, 2, 3, 8):
        parts = sharding.partition_streams(lens, ws)
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        sums = [int(lens[a:b].sum()) for a, b in parts]
        assert max(sums) - min(sums) <= 2 * 5000
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    b, e, lo, hi = sharding.shard_for_rank(off, lens, 1, 4)
    assert lo == int(off[b]) and hi == int(off[e - 1] + lens[e - 1])


def test_header_constants_match_python_side():
    """Flags that cross the ctypes boundary by value: the header is the source of truth."""
    import os
    import re

    from tamp_amd import _lib

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tamp_amd.h")).read()
    m = re.search(r"#define\s+TAMP_AMD_WINDOW_BITS_EXACT\s+(0x[0-9a-fA-F]+)", hdr)
    assert m and int(m.group(1), 16) == _lib.WINDOW_BITS_EXACT
    for name, val in (("TAMP_AMD_MEM_HOST", _lib.MEM_HOST), ("TAMP_AMD_MEM_DEVICE", _lib.MEM_DEVICE),
                      ("TAMP_AMD_NO_DEVICE", _lib.NO_DEVICE), ("TAMP_AMD_BAD_ARGUMENT", _lib.BAD_ARGUMENT)):
        m = re.search(name + r"\s*=\s*(-?\d+)", hdr)
        assert m and int(m.group(1)) == val, name


def test_bench_reads_committed_pmc_summaries():
    """bench.py's roofline.traffic comes from the PMC passes committed under profiles/ (FETCH x2 + WRITE, in bytes)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    t, source = mod.pmc_traffic_bytes()
    assert t is not None and 3e8 < t < 1e11  # at least the 268 MB of input; not absurd
    assert "profiles/" in source


def test_corpus_split_and_sharding_tile_the_file():
    """configs[2] host logic: a file cut into 4 KiB streams (short tail kept) and the streams cut into contiguous ranges per
    GPU cover every byte exactly once, for every world size bench.py is launched with."""
    from tamp_amd import workloads as wl

    blob = bytes(range(256)) * 391 + b"tail!"  # 100,101 bytes: 24 full chunks + one of 1,797
    flat, off, ln = wl.split_fixed(blob, 4096)
    assert len(ln) == 25 and int(ln[-1]) == len(blob) - 24 * 4096 and flat.tobytes() == blob
    assert (off == np.arange(25, dtype=np.uint64) * 4096).all()
    f2, o2, l2 = wl.split_fixed(blob, 4096, keep_tail=False)
    assert len(l2) == 24 and f2.size == 24 * 4096
    for world in (1, 2, 4, 8, 32):
        ranges = sharding.partition_streams(ln, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == 25
        got = b""
        for (b, e), nxt in zip(ranges, ranges[1:] + [(25, 25)]):
            assert e == nxt[0]
            if e > b:
                lo, hi = int(off[b]), int(off[e - 1] + ln[e - 1])
                got += flat[lo:hi].tobytes()
        assert got == blob
    rows = wl.tile_rows(blob, 60, 4096)
    assert rows.shape == (60, 4096) and rows[24].tobytes() == blob[:4096] and rows[59].tobytes() == blob[11 * 4096 : 12 * 4096]
    assert set(wl.ENWIK8_PINS) >= {"len", "v1_sha256", "extended_sha256", "v1_size", "extended_size"}


def test_bench_host_helpers():
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    threads, affinity, quota = bench.host_threads()
    assert 1 <= threads <= affinity and (quota is None or threads <= quota)
    assert isinstance(bench.cpu_model(), str)
    traffic, src = bench.pmc_traffic_bytes()
    assert traffic is None or traffic > 0


def test_cli_dictionary_rule_and_arguments(tmp_path):
    """tamp/cli/main.py:90-105: a dictionary file of the window's size is taken as it is, a shorter one is copied to the END of
    the seeded default, a longer one is an error; option names and ranges of the two commands."""
    import os

    from tamp_amd import _lib, cli

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtamp_amd.so not built")
    full = bytes(range(256)) * 4
    p = tmp_path / "full.bin"
    p.write_bytes(full)
    assert cli.load_dictionary(p, 10, 8, True) == bytearray(full)
    q = tmp_path / "raw.bin"
    q.write_bytes(b"hello world")
    d = cli.load_dictionary(q, 10, 7, True)
    seeded = tamp_amd.initialize_dictionary(1024, literal=7)
    assert len(d) == 1024 and d[-11:] == b"hello world" and d[:-11] == seeded[:-11]
    assert cli.load_dictionary(q, 10, 7, False)[:-11] == tamp_amd.initialize_dictionary(1024, literal=8)[:-11]  # v1: literal-8 table
    big = tmp_path / "big.bin"
    big.write_bytes(bytes(2000))
    with pytest.raises(ValueError):
        cli.load_dictionary(big, 10, 8, True)
    ap = cli.build_parser()
    a = ap.parse_args(["compress", "-i", "x", "-o", "y", "-w", "12", "-l", "7", "--lazy-matching", "--no-extended"])
    assert (a.input, a.output, a.window, a.literal, a.lazy_matching, a.extended) == ("x", "y", 12, 7, True, False)
    a = ap.parse_args(["decompress", "in.tamp", "out.bin", "-d", "dict.bin"])
    assert (a.input_pos, a.output_pos, a.dictionary, a.window, a.extended) == ("in.tamp", "out.bin", "dict.bin", 10, True)
    for bad in (["compress", "-w", "7"], ["compress", "-l", "9"], ["decompress", "--lazy-matching"]):
        with pytest.raises(SystemExit):
            ap.parse_args(bad)


def test_compress_kernels_keep_everything_in_registers(tmp_path):
    """The compress kernels run at the register limit of their occupancy target; a spilled VGPR inside a loop costs a
    scratch access per lane and iteration (round 1: 40 % extra HBM traffic).  Compile the device code with the resource
    remarks on and require 0 spilled VGPRs / 0 B of scratch for every instantiation of tamp_compress_kernel -- except the
    two run-aware builds, which since round 4 aim at SEVEN workgroups per CU (72 VGPRs): a handful of values that live
    across a whole epoch were spilled there in round 4 (1.10 x the algorithmic HBM bytes); round 5's cooperative
    find_extended_match took the per-lane verification loops out of the walk and they spill nothing any more (1.04 x);
    round 6's bucket loop needs fewer registers still and the builds now aim at EIGHT per CU (64 VGPRs), spilling nothing.
    The block-mode build (round 5) reserves 68 B of private segment that no instruction touches: checked in the assembly.
    (hipcc cross-compiles without a GPU)."""
    import os
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tamp_amd", "csrc")
    p = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-I" + os.path.join(root, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-c", "tamp_capi.hip", "-o", str(tmp_path / "dev.o")],
                       cwd=src, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", p.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "tamp_compress_kernel" not in name:
            continue
        seen += 1
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        if "ILb1ELb0ELb1E" in name:  # PACKED, not LAZY, RUNS: eight workgroups per CU since the end of round 6 (64 VGPRs)
            assert spill == 0 and scratch == 0, (name, spill, scratch)
            assert "Occupancy [waves/SIMD]: 8" in b, name
        elif name.endswith("Lb1ELb1EEEvNS_12CompressArgsE"):  # block mode (LOOP, BLOCKM)
            assert spill == 0, (name, spill, scratch)
        else:
            assert spill == 0 and scratch == 0, (name, spill, scratch)
    assert seen == 9, seen  # (six builds of rounds 3-4, DESIGN.md 3.2, + block mode: lean, run-aware generic, run-aware 2^10)
    names = [b.split()[0] for b in blocks if "tamp_compress_kernel" in b.split()[0]]
    assert sum("Lb1ELb0EEEvNS_12CompressArgsE" in n or "Lb1ELb1EEEvNS_12CompressArgsE" in n for n in names) == 8, names  # persistent-grid builds
    # ... and no FLAT memory instruction in the kernels whose control words live in LDS: a volatile generic pointer makes
    # every access one (system scope + full wait), which is what the explicit LDS pointers of DESIGN.md 3.9 removed
    p = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-I" + os.path.join(root, "include"),
                        "-S", "tamp_capi.hip", "-o", str(tmp_path / "dev.s")], cwd=src, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    asm = (tmp_path / "dev.s").read_text()
    parts = re.split(r"\n(_ZN[^\n:]*):[^\n]*\n", asm)
    checked = 0
    for i in range(1, len(parts) - 1, 2):
        name, body = parts[i], parts[i + 1]
        if not any(k in name for k in ("tamp_compress_kernel", "tamp_decode_resolve_kernel", "tamp_decode_parse_kernel")):
            continue
        body = body.split(".end_amdhsa_kernel")[0]
        checked += 1
        assert "flat_load" not in body and "flat_store" not in body, name
        if "tamp_compress_kernel" in name:
            assert "scratch_load" not in body and "scratch_store" not in body, name  # (nothing lives in private memory)
            # the wavefront priorities of DESIGN.md 3.14: scan lowest, short phases above it, the walk on top
            assert all(f"s_setprio {k}" in body for k in (0, 2, 3)), name
    assert checked >= 8


def test_frozen_corpora_match_their_manifest():
    """tests/golden/corpus_{prose,python}.txt.xz: the real-text inputs of the GPU parity tests and of bench.py's
    `real_text` figures are committed data, not whatever the machine happens to hold."""
    from tamp_amd import workloads as wl

    for name in ("prose", "python", "markup"):
        raw = wl.frozen_corpus(name)  # raises when the SHA-256 of the manifest does not match
        assert len(raw) == 3 << 20
        assert wl.real_text(name, 1 << 20) == raw[: 1 << 20]
        assert wl.real_text(name, 64 << 20) == raw  # never topped up from the machine


def test_corpus_probe_finds_plain_and_zipped_files_and_reports_what_it_tried(tmp_path):
    """bench.py looks for the metric's own corpus (enwik8) by itself -- $TAMP_CORPUS, a list of directories (plain or
    .zip), one guarded download -- and puts what it tried into config.corpus_probe.  Here: the
    probe against temp directories, with a 1,000-byte stand-in for the 100,000,000-byte file; no network is touched."""
    import zipfile

    from tamp_amd import workloads as wl

    want = 1000
    empty, plain, zipped, wrong = (tmp_path / n for n in ("empty", "plain", "zipped", "wrong"))
    for d in (empty, plain, zipped, wrong):
        d.mkdir()
    path, rec = wl.probe_corpus(env={}, dirs=[str(empty)], fetch=False, want_len=want)
    assert path is None and rec["found"] is None and rec["fetch"] == "not attempted"
    assert rec["tried"] == [str(empty / "enwik8"), str(empty / "enwik8.zip")]
    (wrong / "enwik8").write_bytes(b"x" * (want - 1))  # wrong length: not the corpus
    (plain / "enwik8").write_bytes(bytes(range(250)) * 4)
    path, rec = wl.probe_corpus(env={}, dirs=[str(empty), str(wrong), str(plain)], fetch=False, want_len=want)
    assert path == str(plain / "enwik8") and rec["found"] == path and str(wrong / "enwik8") in rec["tried"]
    with zipfile.ZipFile(zipped / "enwik8.zip", "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("readme.txt", "not it")
        z.writestr("enwik8", bytes(range(250)) * 4)
    out = tmp_path / "unpacked"
    path, rec = wl.probe_corpus(env={}, dirs=[str(zipped)], fetch=False, fetch_to=str(out), want_len=want)
    assert path == str(out / "enwik8") and open(path, "rb").read() == bytes(range(250)) * 4
    # $TAMP_CORPUS comes first
    path, rec = wl.probe_corpus(env={"TAMP_CORPUS": str(plain / "enwik8")}, dirs=[str(empty)], fetch=False, want_len=want)
    assert path == str(plain / "enwik8") and rec["tried"][0] == path
    # the default list and the URL are the ones the bench documents; the real length is enwik8's
    assert wl.ENWIK8_URL.endswith("/dc/enwik8.zip") and wl.ENWIK8_PINS["len"] == 100_000_000
    assert all(not d.startswith("/root/reference") for d in wl.CORPUS_PROBE_DIRS)  # (never read at run time)


def test_round5_advice_items(tmp_path):
    """the corpus probe never fetches unless told to; a build-dictionary delimiter is taken verbatim as the
    reference does (tamp/cli/build_dictionary.py:696-699: `content.split(delimiter.encode())`), an empty corpus is an
    error, and a selection that fits nothing falls back to the SMALLEST tabulated size, not the full dictionary."""
    import inspect

    from tamp_amd import build_dictionary as bd, workloads as wl

    assert inspect.signature(wl.probe_corpus).parameters["fetch"].default is False
    src = tmp_path / "m.txt"
    src.write_bytes("aé1éb\\nc".encode())
    assert bd.read_corpus(src, "é") == [b"a", b"1", "b\\nc".encode()]
    assert bd.read_corpus(src, "\\n") == ["aé1éb".encode(), b"c"]  # a literal backslash-n is two bytes, not a newline
    (tmp_path / "empty.txt").write_bytes(b"\n\n")
    with pytest.raises(ValueError):
        bd.build_dictionary_cli(tmp_path / "empty.txt", tmp_path / "d.bin", window=8, literal=7, total=lambda d: 0, quiet=True)
    # a target fill below anything the miner can place: the smallest tabulated size is written and marked
    corpus = [b'{"id":"dev-%05d","ok":true}' % i for i in range(200)]
    (tmp_path / "c.txt").write_bytes(b"\n".join(corpus))
    lines = []
    res = bd.build_dictionary_cli(tmp_path / "c.txt", tmp_path / "d.bin", window=8, literal=7, trim_threshold=4, target_fill=0.004,
                                  total=lambda d: 1000 if d is None else 1000 - sum(1 for b in d[-64:] if b in b'{"id:dev-ok,true}'),
                                  log=lambda *a: lines.append(" ".join(map(str, a))))
    assert 0 < res["dictionary_bytes"] <= res["tradeoff"][0][0] and any("<-- selected" in x for x in lines)
    assert len((tmp_path / "d.bin").read_bytes()) == res["dictionary_bytes"]


def test_standin_rows_for_configs2_are_frozen():
    """The stand-in for BASELINE configs[2] (enwik8 is out of reach): 24,414 x 4 KiB rows drawn from the three frozen
    corpora with a fixed seed -- the same rows on every box, all three corpora present."""
    from tamp_amd import workloads as wl

    a, b = wl.standin_rows(512), wl.standin_rows(512)
    assert a.shape == (512, 4096) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert wl.CONFIGS2_STREAMS == 24414
    chunks = {bytes(r[:64]) for r in a}
    for name in ("prose", "markup", "python"):
        flat = wl.real_text(name, frozen_only=True)
        heads = {flat[i * 4096: i * 4096 + 64] for i in range(len(flat) // 4096)}
        assert chunks & heads, name


def test_build_dictionary_pipeline_with_a_cpu_evaluator(tmp_path, oracle):
    """`python -m tamp_amd build-dictionary` (tamp/cli/build_dictionary.py:706-927): the mining, packing,
    tradeoff table and knee on a corpus of 600 telemetry messages, with the whole-corpus evaluation -- one GPU batch launch
    per dictionary in the product -- replaced by the oracle here (CPU tier).  The file holds the effective bytes only,
    `cli.load_dictionary` puts them at the end of a seeded window, and that window beats the seeded default by a wide margin."""
    from tamp_amd import build_dictionary as bd, cli, workloads as wl

    rows = wl.telemetry(600, 256)
    corpus = [bytes(r).rstrip(b" ") for r in rows]
    src = tmp_path / "messages.txt"
    src.write_bytes(b"\n".join(corpus))
    assert bd.read_corpus(src) == corpus
    d = tmp_path / "as_files"
    d.mkdir()
    for i, s in enumerate(corpus[:5]):
        (d / f"{i:03d}.bin").write_bytes(s)
    (d / "empty.bin").write_bytes(b"")
    assert bd.read_corpus(d) == corpus[:5]
    with pytest.raises(ValueError):
        bd.read_corpus(tmp_path / "nothing-here")

    flat = np.frombuffer(b"".join(corpus), dtype=np.uint8)
    ln = np.array([len(s) for s in corpus], np.uint32)
    off = np.zeros(len(ln), np.uint64)
    off[1:] = np.cumsum(ln[:-1].astype(np.uint64))

    def total(dictionary):
        r = oracle.compress_batch(flat, off, ln, window=8, literal=7, extended=True, threads=8,
                                  dictionary=None if dictionary is None else bytes(dictionary))
        assert (r.status == 0).all()
        return int(r.out_len.astype(np.int64).sum()) - len(ln)

    out = tmp_path / "dictionary.bin"
    lines = []
    res = bd.build_dictionary_cli(src, out, window=8, literal=7, trim_threshold=4, total=total, log=lambda *a: lines.append(" ".join(map(str, a))))
    blob = out.read_bytes()
    assert 0 < len(blob) == res["dictionary_bytes"] <= 256
    assert res["with_dictionary"] < 0.75 * res["baseline"], res
    window = cli.load_dictionary(out, 8, 7, True)
    assert len(window) == 256 and bytes(window[-len(blob):]) == blob and total(window) == res["with_dictionary"]
    assert any("<-- selected" in x for x in lines) and any(x.startswith("With dict:") for x in lines)
    sizes = [s for s, _ in res["tradeoff"]]
    assert sizes == sorted(sizes) and res["knee"] in sizes
    # the pieces: savings table, packing order, knee
    assert bd.bits_saved(2, 2, 10, 8, True) == 2 * 9 - (2 + 10) and bd.bits_saved(1, 2, 10, 8, True) == 0
    assert bd.bits_saved(14, 2, 10, 8, True) == 14 * 9 - (7 + (2 - 1) + 3 + 10)  # extended match: symbol 13, code 0, 3 bits, offset
    assert bd.bits_saved(14, 2, 10, 8, False) == 14 * 9 - (9 + 10)
    w, eff = bd.pack([(b"late", 100.0, 0.9), (b"early", 100.0, 0.1), (b"dense", 400.0, 0.9)], 8, 8, True)
    assert eff == 14 and bytes(w[-5:]) == b"dense" and bytes(w[-9:-5]) == b"late" and bytes(w[-14:-9]) == b"early"
    assert bd.find_knee([(0, 1000), (10, 500), (20, 480), (30, 470)]) == 10
    assert bd.find_knee([(0, 1000), (10, 900), (20, 800), (30, 700)]) == 30
    a = cli.build_parser().parse_args(["build-dictionary", "corpus/", "-o", "d.bin", "-w", "8", "-l", "7", "-t", "4", "-f", "0.5", "-q", "--no-extended"])
    assert (a.input, a.output, a.window, a.literal, a.trim_threshold, a.target_fill, a.quiet, a.extended) == ("corpus/", "d.bin", 8, 7, 4, 0.5, True, False)


def test_helpers_against_values_recorded_from_the_reference():
    """tests/golden/helpers.json (make_helpers_golden.py): bit_size, compute_min_pattern_size and initialize_dictionary
    with non-default seeds, as the reference's own Python definitions answer (tamp/__init__.py:18-70)."""
    import hashlib

    gold = load_golden("helpers.json")
    for v, want in gold["bit_size"].items():
        assert tamp_amd.bit_size(int(v)) == want, v
    for key, want in gold["min_pattern"].items():
        w, l = map(int, key.split(","))
        assert tamp_amd.compute_min_pattern_size(w, l) == want
    for case in gold["seeded"]:
        got = bytes(tamp_amd.initialize_dictionary(case["size"], seed=case["seed"], literal=case["literal"]))
        assert got[:16].hex() == case["head"] and hashlib.sha256(got).hexdigest() == case["sha256"], case


def test_import_tamp_resolves_to_this_package():
    """The drop-in name: ``import tamp`` gives the reference's public names, each the very object tamp_amd exports."""
    import tamp

    for name in ("compress", "decompress", "Compressor", "Decompressor", "TextCompressor", "TextDecompressor", "open",
                 "initialize_dictionary", "compute_min_pattern_size", "bit_size", "ExcessBitsError"):
        assert getattr(tamp, name) is getattr(tamp_amd, name), name


def test_bench_flags_of_round_5_exist_and_the_probe_is_opt_in():
    """`bench.py --help` (no GPU needed): the corpus probe and the download are opt-in flags, the configs[2] stand-in and the
    shard fraction are there."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    for flag in ("--probe-corpus", "--fetch-corpus", "--corpus-standin", "--shard-fraction", "--corpus"):
        assert flag in out.stdout, flag
    src = open(os.path.join(root, "bench.py")).read()
    assert "wl.probe_corpus(env={}, fetch=(args.fetch_corpus and rank == 0))" in src  # (never without --probe-corpus, never fetching by default)


def test_long_stream_chunk_starts_settle_on_the_token_boundaries(oracle):
    """The idea behind tamp_decompress_long_kernel.hpp, restated on the CPU: cut a v1 stream's bits into chunks, parse every chunk
    from a GUESSED start, hand each chunk's exit to the next chunk as its start, repeat -- the fixed point is the sequential
    parse's own boundaries (chunk 0 starts behind the header), text gets there in two or three rounds, and a run of one token
    (a periodic bit stream never re-synchronises) a chunk per round.  Token grammar: decompressor.c:431-575 (v1)."""
    import tamp_amd
    from tamp_amd import workloads as wl

    codes_lo, codes_hi, nbits = 0x2B2624140B080300, 0x00AB27AA9594544B, 0x979998877765532
    table = {"0": 0}
    for sy in range(1, 15):
        l = ((nbits >> (4 * sy)) & 15) - 1
        code = ((codes_lo >> (8 * sy)) if sy < 8 else (codes_hi >> (8 * (sy - 8)))) & 0xFF
        table["1" + format(code & ((1 << (l - 1)) - 1), f"0{l - 1}b") if l > 1 else "1"] = sy
    assert len(table) == 15 and abs(sum(2.0 ** -len(k) for k in table) - 1.0) < 1e-12  # a complete prefix code

    def token(bits, t, wbits, lbits):  # -> bits the token at t takes (0: the stream ends), bytes it yields
        n = len(bits)
        if t >= n:
            return 0, 0
        if bits[t] == "1":
            return (1 + lbits, 1) if t + 1 + lbits <= n else (0, 0)
        k = 1
        while True:
            if t + 1 + k > n:
                return 0, 0
            sym = table.get(bits[t + 1 : t + 1 + k])
            if sym is not None:
                break
            k += 1
        if sym == 14:  # FLUSH: to the byte boundary
            u = 1 + k
            return u + (-(t + u)) % 8, 0
        if t + 1 + k + wbits > n:
            return 0, 0
        return 1 + k + wbits, sym + minp

    C = 512
    for name, data, window in (("text", bytes(wl.synth_text(1, 30_000)[0]), 10), ("runs", bytes(wl.lcg_runs(1, 20_000)[0]), 8),
                               ("zeros", bytes(40_000), 10), ("period", (b"abcdefghijklmnopqrstuvw" * 2000)[:40_000], 12)):
        st, blob = oracle.stream_script([("write", data[: len(data) // 2]), ("flush", True), ("write", data[len(data) // 2 :]), ("close",)],
                                        window=window, literal=8, extended=False)
        assert st == 0
        minp = tamp_amd.compute_min_pattern_size(window, 8)  # (host helper: a pure function, tamp/__init__.py:54-70)
        bits = "".join(format(b, "08b") for b in blob)
        # the sequential parse: boundaries and output size
        bounds, t, total = set(), 8, 0
        while True:
            bounds.add(t)
            k, nb = token(bits, t, window, 8)
            if not k:
                break
            t += k
            total += nb
        assert total == len(data), name
        N = (len(bits) + C - 1) // C
        g = [i * C for i in range(N + 1)]
        g[0] = 8
        rounds = 0
        while True:
            nxt = list(g)
            for i in range(N):
                t = g[i]
                while t is not None and t < (i + 1) * C:
                    k, _ = token(bits, t, window, 8)
                    t = t + k if k else None
                nxt[i + 1] = t
            rounds += 1
            if nxt == g:
                break
            g = nxt
            assert rounds <= N + 2, name
        assert all(x is None or x in bounds for x in g[:-1]), name  # every settled start is a true token boundary
        if name in ("text", "runs"):
            assert rounds <= 4, (name, rounds)


def test_long_stream_groups_tail_maps_and_window_pos_blocks_model():
    """Round 6, the long-stream decoder's step 3 restated on the CPU (tamp_decompress_long_kernel.hpp): a token stream is cut
    into groups; every group is resolved ALONE -- a byte whose source lies in front of the group stays "byte j of the window in
    front" -- and leaves a tail map; the maps compose into the window in front of every group; a second pass fills in the
    externals.  With RLE / extended-match tokens (decompressor.c:114-273: fewer bytes written to the window than produced,
    clipped at the ring's end) window_pos comes from a pass over those tokens in blocks: every block from every start value,
    one look-up per block.  Checked against a plain sequential decoder, random token streams, small windows."""
    import random

    def sequential(tokens, W, dic):
        win, wp, out = list(dic), 0, []
        for t in tokens:
            if t[0] == "lit":
                src, wr = [t[1]], 1
            elif t[0] == "copy":
                src = [win[t[1] + i] for i in range(t[2])]
                wr = t[2]
            elif t[0] == "rle":
                src = [win[(wp - 1) % W]] * t[1]
                wr = min(t[1], 8, W - wp)
            else:  # ext
                src = [win[t[1] + i] for i in range(t[2])]
                wr = min(t[2], W - wp)
            out += src
            for b in src[:wr]:
                win[wp] = b
                wp = (wp + 1) % W
        return out

    def window_pos_in_blocks(tokens, W, block):
        """bytes every token writes to the window: tamp_long_wp_kernel's three modes"""
        specials = [(i, t) for i, t in enumerate(tokens) if t[0] in ("rle", "ext")]
        entries, last = [], 0  # (gap of plain bytes in front, kind, produced)
        plain = [0]
        for t in tokens:
            plain.append(plain[-1] + ((1 if t[0] == "lit" else t[2]) if t[0] in ("lit", "copy") else 0))
        for i, t in specials:
            entries.append((plain[i] - plain[last], t[0], t[1] if t[0] == "rle" else t[2]))
            last = i + 1

        def run(wp, ents, written=None):
            for gap, kind, L in ents:
                wp = (wp + gap) % W
                w = min(min(L, 8) if kind == "rle" else L, W - wp)
                if written is not None:
                    written.append(w)
                wp = (wp + w) % W
            return wp

        blocks = [entries[i:i + block] for i in range(0, len(entries), block)]
        tables = [[run(wp, b) for wp in range(W)] for b in blocks]  # MODE 0
        starts, wp = [], 0
        for tab in tables:  # MODE 1
            starts.append(wp)
            wp = tab[wp]
        written = []
        for b, s in zip(blocks, starts):  # MODE 2
            run(s, b, written)
        return {i: w for (i, _), w in zip(specials, written)}

    def grouped(tokens, W, dic, group_out, wp_block):
        written = window_pos_in_blocks(tokens, W, wp_block)
        # groups of whole tokens
        groups, cur, size = [], [], 0
        for i, t in enumerate(tokens):
            n = 1 if t[0] == "lit" else (t[1] if t[0] == "rle" else t[2])
            if cur and size + n > group_out:
                groups.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += n
        groups.append(cur)
        V = 0  # bytes written to the window in front of the group
        resolved, maps = [], []
        for g in groups:
            rot = V % W
            out, virt = [], []  # per output byte: ("b", byte) / ("o", output position in the group) / ("x", j); virtual -> output position
            for i in g:
                t = tokens[i]
                Vj = len(virt)
                if t[0] == "lit":
                    src, wr = [("b", t[1])], 1
                else:
                    n = t[1] if t[0] == "rle" else t[2]
                    src = []
                    for k in range(n):
                        idx = (Vj - 1) % W if t[0] == "rle" else (t[1] - rot + k) % W  # rotated ring index read
                        back = (Vj - 1 - idx) % W
                        src.append(("x", idx) if back >= Vj else ("o", virt[Vj - 1 - back]))
                    wr = n if t[0] == "copy" else written[i]
                base = len(out)
                out += src
                virt += [base + k for k in range(wr)]
            # pointers resolved inside the group
            for p in range(len(out)):
                while out[p][0] == "o":
                    out[p] = out[out[p][1]]
            n_written = len(virt)
            tail = []
            for k in range(W):
                v = n_written + k - W
                tail.append(out[virt[v]] if v >= 0 else ("x", n_written + k))
            resolved.append(out)
            maps.append(tail)
            V += n_written
        # the window in front of every group: the maps composed from the fresh decoder's window
        win, fronts = list(dic), []
        for m in maps:
            fronts.append(win)
            win = [e[1] if e[0] == "b" else win[e[1]] for e in m]
        res = []
        for out, front in zip(resolved, fronts):
            res += [e[1] if e[0] == "b" else front[e[1]] for e in out]
        return res

    rng = random.Random(6)
    for W in (16, 32, 64):
        for trial in range(60):
            dic = [rng.randrange(256) for _ in range(W)]
            tokens = []
            for _ in range(rng.randrange(1, 400)):
                u = rng.random()
                if u < 0.45:
                    tokens.append(("lit", rng.randrange(256)))
                elif u < 0.75:
                    n = rng.randrange(2, min(W, 16))
                    tokens.append(("copy", rng.randrange(0, W - n + 1), n))
                elif u < 0.9:
                    tokens.append(("rle", rng.randrange(2, 40)))
                else:
                    n = rng.randrange(2, W)
                    tokens.append(("ext", rng.randrange(0, W - n + 1), n))
            want = sequential(tokens, W, dic)
            for group_out, wp_block in ((37, 5), (150, 64), (10**9, 10**9)):
                assert grouped(tokens, W, dic, group_out, wp_block) == want, (W, trial, group_out)
