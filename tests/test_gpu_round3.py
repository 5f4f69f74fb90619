"""Round-3 GPU tests: very long single streams (the one-workgroup path bench.corpus_pins takes for enwik8), the
one-process N-shard launch path of bench.py, and a sweep of ragged / special streams at small windows.  All `-m gpu`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tamp_amd

    return tamp_amd


@pytest.fixture(scope="module")
def checker():
    from oracle.checker import Oracle, Ref

    return Ref() if Ref.available() else Oracle()


def _tiled(nbytes: int) -> bytes:
    """Real text of exactly ``nbytes``: the frozen prose and Python corpora alternating, cut at odd places so that the
    copies do not line up with the 4 KiB / 64-position grids of the kernels."""
    from tamp_amd import workloads as wl

    prose, py = wl.real_text("prose"), wl.real_text("python")
    parts, total, k = [], 0, 0
    while total < nbytes:
        src = prose if k % 2 == 0 else py
        cut = len(src) - 4099 * (k % 7) - 1
        parts.append(src[:cut])
        total += cut
        k += 1
    return b"".join(parts)[:nbytes]


def test_one_16_mib_stream_matches_reference(ta, checker):
    """ONE stream of 16 MiB through one workgroup, both formats, byte for byte against the reference C (the largest
    one-shot stream of round 2 was 300 KB)."""
    blob = _tiled(16 << 20)
    flat = np.frombuffer(blob, dtype=np.uint8)
    off, ln = np.zeros(1, np.uint64), np.array([len(blob)], np.uint32)
    for ext in (True, False):
        want = checker.compress_batch(flat, off, ln, window=10, literal=8, extended=ext)
        got = ta.compress_batch(flat, off, ln, window=10, literal=8, extended=ext)
        assert int(got.status[0]) == 0 and int(want.status[0]) == 0
        assert got.stream(0) == want.stream(0), f"16 MiB stream differs (extended={ext})"
        back = ta.decompress_batch([want.stream(0)], out_cap=len(blob) + 8)
        assert int(back.status[0]) == 2 and back.stream(0) == blob


def test_corpus_pins_path_on_a_100_000_000_byte_file(ta, checker):
    """bench.corpus_pins -- the code that will claim the enwik8 whole-file pins -- on a file of enwik8's size: one
    100,000,000-byte stream per format, compared with the reference C on the same bytes (a tiled real-text file has no
    published pin, so `sha256_matches_reference` must come out False and `matches_checker` True)."""
    import torch

    sys.path.insert(0, ROOT)
    import bench

    blob = _tiled(100_000_000)
    out = bench.corpus_pins(None, blob, torch, np, reference=checker)
    assert out["checked"] is True and out["bytes"] == 100_000_000
    for tag in ("v1", "extended"):
        assert out[tag]["status"] == 0
        assert out[tag]["matches_checker"] is True, (tag, out[tag])
        assert out[tag]["sha256_matches_reference"] is False  # not enwik8: the published pin must not be claimed
        assert out[tag]["size"] == out[tag]["checker_size"]


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_one_process_eight_shards_keep_the_device_fed():
    """`python bench.py --gpus 8` as ONE process issuing eight launches per step (north_star's launch model), all shards on
    this box's one device: the wall clock of a step must not exceed the kernels' own time by more than 5 % -- a host-side
    launch bottleneck (allocation, fill kernels, Python per shard) would show up here -- and the per-shard enqueue cost
    is reported."""
    line = _bench(["--gpus", "8", "--streams", "8192", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                  {"TAMP_BENCH_ONE_DEVICE": "1"})
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["streams_total"] == 8 * 8192 and cfg["all_streams_ok"]
    assert cfg["host_launch_us_per_shard"] > 0
    assert line["ms_per_step"] <= 1.05 * cfg["sum_kernel_ms_per_step"], (line["ms_per_step"], cfg)
    # a launch must cost the host far less than the kernel it starts (8,192 streams run ~0.9 ms)
    assert cfg["host_launch_us_per_shard"] < 300, cfg


def test_corpus_strong_scaling_path_with_four_shards(tmp_path):
    """--corpus with --gpus 4 (strong scaling: contiguous stream ranges balanced by bytes), all shards on one device."""
    from tamp_amd import workloads as wl

    p = tmp_path / "corpus.txt"
    p.write_bytes(wl.real_text("prose")[: (2 << 20) + 321])
    line = _bench(["--gpus", "4", "--corpus", str(p), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                  {"TAMP_BENCH_ONE_DEVICE": "1"})
    assert line["scaling"] == "strong" and line["n_gpus"] == 4
    assert line["config"]["streams_total"] == 513 and line["config"]["all_streams_ok"]
    assert line["ms_per_step"] <= 1.05 * line["config"]["sum_kernel_ms_per_step"] + 0.2


def test_ragged_special_and_real_text_streams_small_windows(ta, checker):
    """Same bytes as the reference on real text cut into 4 KiB streams, ragged lengths (0..9,000 bytes), runs and periodic
    data, windows 2^8..2^10, both formats.  (Round 3 ran these cases against the opt-in tile-ring encoder, which round 4
    removed -- slower on every input; they stay as a sweep of the one encoder.)"""
    from tamp_amd import workloads as wl

    rng = np.random.default_rng(11)
    prose = wl.real_text("prose")
    lens = np.concatenate([np.arange(0, 24), rng.integers(1, 9000, 120)]).astype(np.uint32)
    off = np.zeros(len(lens), np.uint64)
    off[1:] = np.cumsum(lens[:-1])
    ragged = (np.frombuffer(prose[: int(lens.sum())], dtype=np.uint8), off, lens)
    specials = [bytes(5000), b"ab" * 3000, bytes([7]) * 300 + b"xyz" + bytes([7]) * 3000, (b"x" * 20 + b"hello world ") * 300,
                bytes(rng.integers(0, 4, 6000, dtype=np.uint8)), bytes(rng.integers(0, 256, 6000, dtype=np.uint8))]
    sl = np.array([len(x) for x in specials], np.uint32)
    so = np.zeros(len(sl), np.uint64)
    so[1:] = np.cumsum(sl[:-1])
    cases = [ragged, (np.frombuffer(b"".join(specials), dtype=np.uint8), so, sl)]
    for name in ("prose", "python"):
        cases.append(wl.split_fixed(wl.real_text(name)[: (1 << 20) + 777], 4096))
    for flat, o, l in cases:
        for window in (8, 9, 10):
            for ext in (True, False):
                want = checker.compress_batch(flat, o, l, window=window, literal=8, extended=ext, threads=8)
                got = ta.compress_batch(flat, o, l, window=window, literal=8, extended=ext, max_in_len=int(l.max()))
                for i in range(len(l)):
                    assert got.stream(i) == want.stream(i) and int(got.status[i]) == int(want.status[i]), (window, ext, i)


def test_split_decoder_scratch_failure_falls_back_and_trim_releases(ta, checker, monkeypatch):
    """the split decoder's record slab (hundreds of MB for 4 KiB streams) is cached per HIP stream; when the
    device cannot supply it the slice shrinks, and below 4,096 streams the batch is decoded by the lane / wave decoders
    instead of failing the call.  tamp_amd_trim() hands the cached slabs back."""
    from tamp_amd import workloads as wl

    n, L = 6000, 4096
    rows = wl.synth_text(n, L)
    off, ln = wl.csr_for_fixed(n, L)
    comp = ta.compress_batch(rows.reshape(-1), off, ln, window=10, literal=8, max_in_len=L)
    streams = [comp.stream(i) for i in range(n)]

    def decode_and_check():
        back = ta.decompress_batch(streams, out_cap=L + 8)
        assert (np.asarray(back.status) == 2).all()
        for i in range(0, n, 37):
            assert back.stream(i) == rows[i].tobytes(), i

    ta.trim(0)
    decode_and_check()                       # split decoder, slab allocated
    assert ta.trim(0) > 0                    # ... and released
    assert ta.trim(0) == 0
    monkeypatch.setenv("TAMP_AMD_SPLIT_FAIL_ABOVE", str(40 << 20))  # every slab above 40 MB is refused: the slice halves
    decode_and_check()
    monkeypatch.setenv("TAMP_AMD_SPLIT_FAIL_ABOVE", "1")            # nothing can be allocated: lane / wave decoders
    ta.trim(0)
    decode_and_check()
    monkeypatch.delenv("TAMP_AMD_SPLIT_FAIL_ABOVE")
    decode_and_check()


def test_numpy_integer_capacities_and_workspace_reuse(ta, checker):
    """out_cap as a numpy integer scalar (`isinstance(out_cap, int)` sent it down the tensor path), and the
    `reuse=` workspace of steady-state callers: same bytes, same buffers."""
    import torch

    from tamp_amd import workloads as wl

    n, L = 512, 4096
    rows = wl.synth_text(n, L)
    off, ln = wl.csr_for_fixed(n, L)
    dev = torch.device("cuda:0")
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t, len_t = torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev)
    want = checker.compress_batch(rows.reshape(-1), off, ln, window=10, literal=8, extended=True, threads=8)
    cap = np.int64(ta.compress_bound(L, 8))
    r1 = ta.compress_batch(data, off_t, len_t, out_cap=cap, max_in_len=L)
    r2 = ta.compress_batch(data, off_t, len_t, out_cap=np.uint32(cap), max_in_len=L, reuse=r1)
    torch.cuda.synchronize()
    assert r2.out.data_ptr() == r1.out.data_ptr() and r2.out_len.data_ptr() == r1.out_len.data_ptr()
    for i in range(n):
        assert r2.stream(i) == want.stream(i)
    with pytest.raises((TypeError, ValueError, AttributeError)):
        ta.compress_batch(data, off_t, len_t, out_cap=True, max_in_len=L)  # bool is not a capacity
    back = ta.decompress_batch(r2.out, r2.out_off, r2.out_len, out_cap=np.int32(L + 8))
    assert bool((back.status == 2).all().item())
    host = ta.compress_batch(rows.reshape(-1), off, ln, out_cap=np.int64(cap), max_in_len=L)  # host path, integer capacity
    assert host.stream(5) == want.stream(5)


def test_compressor_write_streams_like_the_reference_object(ta):
    """tamp.Compressor.write() (tamp/_c_compressor.pyx:74-118): bytes leave during the call and the stream is the
    reference object's, with every write handed over as a piece (PIECE_MIN = 1) -- runs and extended matches growing
    across calls, pieces shorter than the 16-byte ring, both formats, FLUSH tokens in between; the counts returned are
    the reference's up to the bits its object holds back until the next poll."""
    import io

    from oracle.checker import Ref
    from tamp_amd import workloads as wl

    if not Ref.available():
        pytest.skip("oracle/_ref (the reference C built in place) is the only checker with object-level calls")
    ref = Ref()
    rng = np.random.default_rng(2024)
    prose, py = wl.real_text("prose"), wl.real_text("python")
    runs = (b"x" * 700 + b"header\n" + b" " * 300 + b"y" * 5 + b"-" * 77 + b"\n") * 6
    sources = [prose[40_000:140_000], py[10_000:90_000], runs, bytes(3000) + prose[:5000] + bytes(600)]
    old_min = ta.Compressor.PIECE_MIN
    ta.Compressor.PIECE_MIN = 1
    try:
        for si, src in enumerate(sources):
            for ext in (True, False):
                for trial in range(3):
                    ops, pos = [], 0
                    while pos < len(src):
                        k = int(rng.choice([1, 3, 15, 16, 17, 40, 300, 5000, 20000]))
                        ops.append(("write", src[pos : pos + k]))
                        pos += k
                        if rng.random() < 0.08:
                            ops.append(("flush", bool(rng.integers(0, 2))))
                    ops.append(("close",))
                    want_counts = []
                    rc, want = ref.stream_script(ops, window=10, literal=8, extended=ext, counts=want_counts)
                    assert rc == 0
                    f = io.BytesIO()
                    c = ta.Compressor(f, window=10, literal=8, extended=ext)
                    got_counts = []
                    for op in ops:
                        if op[0] == "write":
                            got_counts.append(c.write(op[1]))
                        elif op[0] == "flush":
                            got_counts.append(c.flush(op[1]))
                        else:
                            got_counts.append(c.close())
                    assert f.getvalue() == want, (si, ext, trial)
                    # The reference flushes whole bytes at the START of a poll (compressor.c:549-551), so its object sits on
                    # the last token's bits (and on the header until the first poll) when a call returns; a piece hands every
                    # whole byte over at once.  Same stream, up to four bytes earlier; equal again at every flush point.
                    gc, wc = np.cumsum(got_counts), np.cumsum(want_counts)
                    for k, op in enumerate(ops):
                        if op[0] == "write":
                            assert 0 <= gc[k] - wc[k] <= 4, (si, ext, trial, k, got_counts[k], want_counts[k])
                        else:
                            assert gc[k] == wc[k], (si, ext, trial, k)
    finally:
        ta.Compressor.PIECE_MIN = old_min
    # the default thresholds: large writes leave in bounded pieces, small ones wait; the stream is the same
    f = io.BytesIO()
    with ta.Compressor(f) as c:
        n1 = c.write(prose[:300_000])
        n2 = c.write(prose[300_000:300_100])
        assert n1 > 0 and n2 == 0
    assert f.getvalue() == ta.compress(prose[:300_100])
    assert bytes(ta.decompress(f.getvalue())) == prose[:300_100]


def test_reference_named_object_takes_large_calls_as_pieces(ta, monkeypatch):
    """tamp_compressor_compress[_cb] on a reference-named object (include/tamp_compat.h): calls of 64 KiB and more go to the
    batch kernel as pieces, smaller ones to the token-level resume kernel -- mixed on ONE object the stream is still the
    reference object's, and the progress callback (common.h:184-210) fires per piece and can abort."""
    import ctypes as C

    from oracle.checker import Ref
    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    if not Ref.available():
        pytest.skip("needs oracle/_ref")
    ref = Ref()
    lib = _lib.load()

    class TampConf(C.Structure):
        _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                    ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1),
                    ("lazy_matching", C.c_uint16, 1)]

    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
    sz = C.POINTER(C.c_size_t)
    lib.tamp_compressor_init.restype = C.c_int8
    lib.tamp_compressor_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tamp_compressor_compress_cb.restype = C.c_int8
    lib.tamp_compressor_compress_cb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_void_p, C.c_size_t, sz, CB, C.c_void_p]
    lib.tamp_compressor_flush.restype = C.c_int8
    lib.tamp_compressor_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_bool]
    no_cb = C.cast(None, CB)
    data = wl.real_text("prose")[100_000:100_000 + 900_000] + b" " * 500 + wl.real_text("python")[:300_000]
    sizes = [70_000, 5, 16, 200_000, 3, 100, 131_072, 15, 65_536, 1, 400_000]
    for ext in (1, 0):
        conf = TampConf(window=10, literal=8, extended=ext)
        window, comp = (C.c_ubyte * 1024)(), (C.c_ubyte * 48)()
        assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
        got, pos, ops = bytearray(), 0, []
        for k in sizes + [len(data)]:
            piece = data[pos : pos + k]
            pos += len(piece)
            if not piece:
                break
            ops.append(("write", piece))
            out = (C.c_ubyte * (len(piece) * 2 + 4096))()
            buf = (C.c_ubyte * len(piece)).from_buffer_copy(piece)
            w, c = C.c_size_t(0), C.c_size_t(0)
            assert lib.tamp_compressor_compress_cb(comp, out, len(out), C.byref(w), buf, len(piece), C.byref(c), no_cb, None) == 0
            assert c.value == len(piece)
            got += bytes(out[: w.value])
        out, w = (C.c_ubyte * 64)(), C.c_size_t(0)
        assert lib.tamp_compressor_flush(comp, out, 64, C.byref(w), False) == 0
        got += bytes(out[: w.value])
        rc, want = ref.stream_script(ops + [("flush", False)], window=10, literal=8, extended=bool(ext))
        assert rc == 0 and bytes(got) == want, ext
    # progress per piece, abort with a custom code
    monkeypatch.setenv("TAMP_AMD_PROGRESS_PIECE_MB", "1")
    conf = TampConf(window=10, literal=8, extended=1)
    window, comp = (C.c_ubyte * 1024)(), (C.c_ubyte * 48)()
    assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
    seen = []
    cb = CB(lambda user, done, total: (seen.append((done, total)), 0 if done < (2 << 20) else 105)[1])
    big = data[: (3 << 20) // 2 * 2][: 3 << 20] if len(data) >= (3 << 20) else (data * 3)[: 3 << 20]
    buf = (C.c_ubyte * len(big)).from_buffer_copy(big)
    out = (C.c_ubyte * (len(big) * 2))()
    w, c = C.c_size_t(0), C.c_size_t(0)
    assert lib.tamp_compressor_compress_cb(comp, out, len(out), C.byref(w), buf, len(big), C.byref(c), cb, None) == 105
    assert seen == [(1 << 20, 3 << 20), (2 << 20, 3 << 20)] and c.value == 2 << 20


def test_pieces_with_custom_dictionary_append_and_excess_bits(ta):
    """Pieces in the corners of the streaming surface: a custom dictionary (the first piece starts from it, later ones from
    the carried window), `append=True` (FLUSH marker instead of a header), 7-bit literals with an offending byte in a later
    piece (ExcessBitsError as in the reference's write), windows 2^8 and 2^12."""
    import io

    from oracle.checker import Ref
    from tamp_amd import workloads as wl

    if not Ref.available():
        pytest.skip("needs oracle/_ref")
    ref = Ref()
    rng = np.random.default_rng(77)
    text = wl.real_text("python")[500_000:560_000]
    old_min = ta.Compressor.PIECE_MIN
    ta.Compressor.PIECE_MIN = 1
    try:
        for kw in (dict(window=8, dictionary=bytes(rng.integers(32, 127, 256, dtype=np.uint8))),
                   dict(window=12, dictionary=(text[:3000] + bytes(4096))[:4096]),
                   dict(window=10, dictionary_reset=True, append=True),
                   dict(window=9, extended=False)):
            ops, pos = [], 0
            while pos < len(text):
                k = int(rng.choice([5, 16, 33, 700, 9000]))
                ops.append(("write", text[pos : pos + k]))
                pos += k
                if rng.random() < 0.1:
                    ops.append(("flush", True))
            ops.append(("close",))
            rc, want = ref.stream_script(ops, literal=8, **{"lazy_matching": False, **kw})
            assert rc == 0
            f = io.BytesIO()
            c = ta.Compressor(f, literal=8, **kw)
            for op in ops:
                if op[0] == "write":
                    c.write(op[1])
                elif op[0] == "flush":
                    c.flush(op[1])
                else:
                    c.close()
            assert f.getvalue() == want, kw
        # literal=7: the byte 0xE9 sits in the third piece
        seven = bytes(b & 0x7F for b in text[:6000])
        bad = seven[:4000] + b"\xe9" + seven[4001:]
        f = io.BytesIO()
        c = ta.Compressor(f, literal=7)
        c.write(bad[:1500])
        c.write(bad[1500:3000])
        with pytest.raises(ta.ExcessBitsError):
            c.write(bad[3000:])
            c.flush()
    finally:
        ta.Compressor.PIECE_MIN = old_min


def _ragged_batch(rng, n, max_len, empties=True):
    """n streams of real text with a heavy-tailed length distribution (a few long ones among many short), some empty."""
    from tamp_amd import workloads as wl

    src = np.frombuffer(wl.real_text("python") + wl.real_text("prose"), dtype=np.uint8)
    ln = np.minimum((rng.pareto(1.2, n) * 200).astype(np.int64) + 1, max_len)
    if empties:
        ln[rng.integers(0, n, max(1, n // 50))] = 0
    start = rng.integers(0, len(src) - max_len, n)
    off = np.zeros(n, np.uint64)
    off[1:] = np.cumsum(ln[:-1])
    flat = np.concatenate([src[a:a + k] for a, k in zip(start, ln)]) if ln.sum() else np.zeros(1, np.uint8)
    return flat, off, ln.astype(np.uint32)


@pytest.mark.parametrize("conf", [
    dict(window=10, literal=8, extended=True),                        # run-aware build, 256 threads, one stream per claim
    dict(window=10, literal=8, extended=True, lazy_matching=True),    # lazy build
    dict(window=15, literal=8, extended=True),                        # u16 entries
    dict(window=8, literal=8, extended=False),
])
def test_persistent_grid_hands_out_every_stream_once(ta, checker, conf, monkeypatch):
    """The compress builds that run as a persistent grid (workgroups take streams from a counter until none is left): batches
    whose size is no multiple of the claim, smaller than one claim, larger than the grid, with empty streams (no walk: the
    next claim is fetched behind the stream instead of during it) and heavy-tailed lengths -- every stream byte for byte
    against the reference, and the same bytes again with one workgroup per claim (TAMP_AMD_STATIC_GRID)."""
    rng = np.random.default_rng(11)
    for n, max_len in ((1, 5000), (5, 300), (17, 700), (1003, 900), (1537, 6000), (9001, 3000)):
        flat, off, ln = _ragged_batch(rng, n, max_len)
        ckw = {("lazy" if k == "lazy_matching" else k): v for k, v in conf.items()}
        want = checker.compress_batch(flat, off, ln, threads=16, **ckw)
        got = ta.compress_batch(flat, off, ln, **conf)
        assert np.array_equal(np.asarray(got.status), np.asarray(want.status))
        for i in range(n):
            assert got.stream(i) == want.stream(i), (conf, n, i, int(ln[i]))
    flat, off, ln = _ragged_batch(rng, 2500, 2000)
    ref = ta.compress_batch(flat, off, ln, **conf)
    monkeypatch.setenv("TAMP_AMD_STATIC_GRID", "1")
    again = ta.compress_batch(flat, off, ln, **conf)
    monkeypatch.delenv("TAMP_AMD_STATIC_GRID")
    assert all(ref.stream(i) == again.stream(i) for i in range(2500))


def test_persistent_grid_short_claims_of_sixteen(ta, checker):
    """One-wavefront launches of the persistent builds (short messages with lazy matching / the run-aware hint) claim sixteen
    streams per fetch: sizes around the claim, and the plain short-message build (one workgroup per stream) next to them."""
    from tamp_amd import workloads as wl

    rng = np.random.default_rng(12)
    for n in (1, 15, 16, 17, 31, 33, 4099):
        rows = wl.synth_text(n, 256)
        ln = rng.integers(0, 257, n).astype(np.uint32)
        off = (np.arange(n, dtype=np.uint64) * 256)
        for kw in (dict(lazy_matching=True), dict(run_aware=True), dict()):
            conf = dict(window=8, literal=8, extended=True)
            conf.update({k: v for k, v in kw.items() if k != "run_aware"})
            ckw = {("lazy" if k == "lazy_matching" else k): v for k, v in conf.items()}
            want = checker.compress_batch(rows.reshape(-1), off, ln, threads=16, **ckw)
            got = ta.compress_batch(rows.reshape(-1), off, ln, max_in_len=256, **conf, **({"run_aware": True} if "run_aware" in kw else {}))
            for i in range(n):
                assert got.stream(i) == want.stream(i), (kw, n, i)


def test_short_runs_with_crowded_buckets_are_matched_on_arrival(ta, checker):
    """Extended format: positions inside short runs of one byte (2..6 ahead, the previous byte the same) whose bigram bucket
    is crowded are left out of the match phase and matched by the walk when a token happens to end inside the run
    (Walk::best_on_demand).  Text made of short runs of a few bytes with words in between -- crowded buckets, tokens ending
    everywhere -- at several windows, plus runs at the window's ends and the same data in the v1 format."""
    rng = np.random.default_rng(21)
    words = [b"self", b"return", b"def", b"x", b"if", b"else:", b"ab", b"for i in", b"=", b"()", b"\n"]
    n, L = 384, 4096
    rows = np.zeros((n, L), np.uint8)
    for i in range(n):
        parts, total = [], 0
        fill = bytes([rng.choice([32, 32, 32, 45, 48])])
        while total < L:
            run = fill * int(rng.integers(2, 9)) if rng.random() < 0.7 else bytes([rng.choice([32, 45, 48])]) * int(rng.integers(2, 14))
            w = words[int(rng.integers(0, len(words)))]
            parts += [run, w]
            total += len(run) + len(w)
        rows[i] = np.frombuffer(b"".join(parts)[:L], np.uint8)
    off = np.arange(n, dtype=np.uint64) * L
    ln = np.full(n, L, np.uint32)
    ln[::7] = rng.integers(1, L, len(ln[::7]))
    for window in (8, 10, 12):
        for ext in (True, False):
            want = checker.compress_batch(rows.reshape(-1), off, ln, window=window, literal=8, extended=ext, threads=16)
            got = ta.compress_batch(rows.reshape(-1), off, ln, window=window, literal=8, extended=ext)
            for i in range(n):
                assert got.stream(i) == want.stream(i), (window, ext, i)


def test_bench_line_keeps_the_contract():
    """The driver's contract for bench.py (one JSON line): every field it reads, the roofline and cpu_baseline objects, the
    live counter passes, the real-text rates next to the headline."""
    line = _bench(["--steps", "3", "--warmup", "1", "--cpu-sample", "2048"], {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["unit"] == "MB/s" and line["dtype"] == "u8" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert line["scaling"] == "weak" and "configs[1]" in line["config"]["workload"] and "model" not in line["config"]
    assert abs(line["value"] - 65536 * 4096 / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert roof["kernel_ms"] <= line["ms_per_step"] * 1.02
    # traffic: measured in the run (or the committed pass): between 1 x and 1.5 x the algorithmic bytes
    assert roof["algorithmic_bytes_per_launch"] <= roof["traffic"] <= 1.5 * roof["algorithmic_bytes_per_launch"], roof
    assert 30_000 < roof["valu_per_stream"] < 80_000 and 0.5 < roof["valu_busy"] <= 1.0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["unit"] == "MB/s" and cpu["value"] > 0
    assert cpu["parity"].startswith("bit-exact"), cpu
    rt = line["config"]["real_text_MBps"]
    assert rt["prose"] > 1000 and rt["python"] > 1000
    assert line["also"]["decompress_round_trip"] == "bit-exact"
