"""GPU tier (-m gpu), round 2: real text, the metric's own workload shape, full-size BASELINE configs, regression
fixtures and bounded randomised runs -- everything through the C ABI, checked against the reference C (oracle/_ref, when
its prebuilt library travelled to the box) or this repo's restatement of it (oracle/), bit for bit.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT, load_golden, unb64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tamp_amd
    from tamp_amd import _lib

    lib = _lib.load()
    assert lib.tamp_amd_device_count() >= 1, "no HIP device visible"
    return tamp_amd


@pytest.fixture(scope="module")
def checker():
    """The reference C itself where its prebuilt library is present, else the restatement."""
    from oracle.checker import Oracle, Ref

    return Ref() if Ref.available() else Oracle()


def _streams_of(res, n):
    olen = np.asarray(res.out_len.cpu() if hasattr(res.out_len, "cpu") else res.out_len)
    ooff = np.asarray(res.out_off.cpu() if hasattr(res.out_off, "cpu") else res.out_off)
    out = res.out.cpu().numpy() if hasattr(res.out, "cpu") else res.out
    return [out[int(ooff[i]) : int(ooff[i]) + int(olen[i])].tobytes() for i in range(n)]


# --------------------------------------------------------------------------------------------------------------
# configs[2] shape on real text: a corpus cut into independent 4 KiB streams, both formats, both kernel builds
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("corpus", ["prose", "python"])
def test_real_text_4k_chunks_match_reference(ta, checker, corpus):
    from tamp_amd import workloads as wl

    blob = wl.real_text(corpus, 6 << 20)
    assert len(blob) >= 3 << 20, "the frozen corpus (tests/golden/corpus_*.txt.xz) is part of the tree"
    # every stream of the first 2 MiB, then every fifth chunk of the rest: ~750 streams, the short tail included
    head, rest = blob[: (2 << 20) + 777], blob[(2 << 20) + 777 :]
    flat, off, ln = wl.split_fixed(head, 4096, keep_tail=True)
    assert ln[-1] == 777 and len(ln) == 513
    k = len(rest) // 4096
    extra = np.frombuffer(rest[: k * 4096], dtype=np.uint8).reshape(k, 4096)[::5]
    flat = np.concatenate([flat, extra.reshape(-1)])
    off = np.concatenate([off, off[-1] + np.uint64(777) + np.arange(len(extra), dtype=np.uint64) * np.uint64(4096)])
    ln = np.concatenate([ln, np.full(len(extra), 4096, dtype=np.uint32)])
    n = len(ln)
    for ext in (True, False):
        want = checker.compress_batch(flat, off, ln, window=10, literal=8, extended=ext, threads=8)
        for run_aware in (None, False, True):
            got = ta.compress_batch(flat, off, ln, window=10, literal=8, extended=ext, max_in_len=4096, run_aware=run_aware)
            assert (np.asarray(got.status) == 0).all()
            gs = _streams_of(got, n)
            for i in range(n):
                assert gs[i] == want.stream(i), (corpus, ext, run_aware, i)
        # and back: decode what the reference produced
        back = ta.decompress_batch([want.stream(i) for i in range(n)], out_cap=4096 + 8)  # (room left: status 2, not 1)
        assert (np.asarray(back.status) == 2).all()
        bs = _streams_of(back, n)
        for i in range(n):
            assert bs[i] == flat[int(off[i]) : int(off[i]) + int(ln[i])].tobytes(), (corpus, ext, i)


def test_real_text_one_long_stream_and_other_windows(ta, checker):
    """configs[0] with a real-text stand-in: the first 64 KiB (and 300 KB) of prose as ONE stream, both formats, lazy
    matching too; other windows / literal sizes on 4 KiB chunks."""
    from tamp_amd import workloads as wl

    blob = wl.real_text("prose", 1 << 20)
    assert len(blob) >= 300_000, "the frozen corpus (tests/golden/corpus_prose.txt.xz) is part of the tree"
    for size in (65536, 300_000):
        data = np.frombuffer(blob[:size], dtype=np.uint8)
        off, ln = np.zeros(1, np.uint64), np.array([size], np.uint32)
        for ext in (True, False):
            for lazy in (False, True):
                want = checker.compress_batch(data, off, ln, window=10, literal=8, extended=ext, lazy=lazy)
                got = ta.compress_batch(data, off, ln, window=10, literal=8, extended=ext, lazy_matching=lazy)
                assert int(got.status[0]) == 0 and _streams_of(got, 1)[0] == want.stream(0), (size, ext, lazy)
    rows = wl.tile_rows(blob[:400 * 4096], 400)
    off, ln = wl.csr_for_fixed(400, 4096)
    for window, literal in ((8, 8), (9, 8), (11, 8), (12, 8), (15, 8), (10, 7)):
        flat = (rows & 0x7F if literal == 7 else rows).reshape(-1)
        for ext in (True, False):
            want = checker.compress_batch(flat, off, ln, window=window, literal=literal, extended=ext, threads=8)
            got = ta.compress_batch(flat, off, ln, window=window, literal=literal, extended=ext, max_in_len=4096)
            gs = _streams_of(got, 400)
            for i in range(400):
                assert gs[i] == want.stream(i) and int(got.status[i]) == int(want.status[i]), (window, literal, ext, i)


def test_corpus_hook_of_bench(ta, tmp_path):
    """bench.py --corpus (configs[2]): any file is cut into 4 KiB streams, every stream checked against the reference,
    and the whole-file pins are only claimed for a file of enwik8's size."""
    from tamp_amd import workloads as wl

    blob = wl.real_text("prose", 3 << 20)
    assert len(blob) >= (1 << 20), "the frozen corpus (tests/golden/corpus_prose.txt.xz) is part of the tree"
    p = tmp_path / "corpus.txt"
    p.write_bytes(blob[: (1 << 20) + 123])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--corpus", str(p), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["streams_total"] == 257 and line["scaling"] == "strong" and line["data"].startswith("real: corpus.txt")
    assert line["cpu_baseline"]["parity"].startswith("bit-exact, 257 streams")
    assert line["corpus_pins"]["checked"] is False


# --------------------------------------------------------------------------------------------------------------
# multi-GPU launch path of bench.py on a one-GPU box: shards of one device, no RCCL
# --------------------------------------------------------------------------------------------------------------
def test_bench_gpus_2_runs_as_invoked_on_one_device(ta):
    env = dict(os.environ, TAMP_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--streams", "4096", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["streams_total"] == 8192 and line["config"]["all_streams_ok"]
    assert "nccl" not in line["config"]["parallelism"] and line["value"] > 0


def test_two_shards_same_code_as_bench_match_oracle(ta, oracle):
    """The N=2 path of bench.py (one Shard per device, its own hipStream, asynchronous launches) on two shards of one
    device: bytes of every stream against the oracle, shard 1 generated from its own first_index."""
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from tamp_amd import workloads as wl

    n, slen = 600, 4096
    shards = []
    for r in range(2):
        rows = wl.synth_text(n, slen, first_index=r * n)
        off, ln = wl.csr_for_fixed(n, slen)
        sh = bench.Shard(torch, r, torch.device("cuda", 0), rows.reshape(-1), off, ln, slen, dict(window=10, literal=8, extended=True))
        sh.rows = rows
        shards.append(sh)
    res = [sh.launch(record=True) for sh in shards]  # both in flight before either is waited for
    for sh in shards:
        sh.sync()
    assert shards[0].stream.cuda_stream != shards[1].stream.cuda_stream
    for sh, r in zip(shards, res):
        off, ln = wl.csr_for_fixed(n, slen)
        want = oracle.compress_batch(sh.rows.reshape(-1), off, ln, window=10, literal=8, extended=True, threads=8)
        gs = _streams_of(r, n)
        for i in range(n):
            assert gs[i] == want.stream(i), (sh.index, i)
        assert sh.events[0][0].elapsed_time(sh.events[0][1]) > 0


# --------------------------------------------------------------------------------------------------------------
# host-memory copy-back: nothing but produced bytes is written when the slabs do not tile the output buffer
# --------------------------------------------------------------------------------------------------------------
def test_host_copy_back_touches_only_produced_bytes(ta, oracle, monkeypatch):
    import ctypes as C

    from tamp_amd import _lib
    from tamp_amd import workloads as wl

    lib = _lib.load()
    n, slen = 300, 1500
    rows = wl.synth_text(n, slen)
    off, ln = wl.csr_for_fixed(n, slen)
    cap1 = ta.compress_bound(slen, 8)
    want = oracle.compress_batch(rows.reshape(-1), off, ln, window=10, literal=8, extended=True, threads=4)
    conf = _lib.TampAmdConf(window=10, literal=8, extended=1)
    monkeypatch.setenv("TAMP_AMD_HOST_CHUNK_STREAMS", "64")
    monkeypatch.setenv("TAMP_AMD_HOST_CHUNK_MB", "1")
    stride = cap1 + 40  # a gap of 40 bytes behind every slab
    for layout in ("gaps", "permuted"):
        order = np.arange(n) if layout == "gaps" else np.random.default_rng(5).permutation(n)
        out_off = (order.astype(np.uint64) * np.uint64(stride)).astype(np.uint64)
        cap = np.full(n, cap1, dtype=np.uint32)
        out = np.full(n * stride + 8, 0xEE, dtype=np.uint8)
        out_len = np.zeros(n, np.uint32)
        status = np.zeros(n, np.int8)
        # single device path (staged copy-back, then the per-stream fallback), then TAMP_AMD_ALL_DEVICES with three shards
        for fan, staged in (("0", True), ("0", False), ("3", True)):
            out[:] = 0xEE
            monkeypatch.setenv("TAMP_AMD_FANOUT", fan)
            if staged:
                monkeypatch.delenv("TAMP_AMD_NO_STAGED_COPYBACK", raising=False)
            else:
                monkeypatch.setenv("TAMP_AMD_NO_STAGED_COPYBACK", "1")
            p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            rc = lib.tamp_batch_compress(C.byref(conf), None, p(rows), p(off), p(ln), p(out), p(out_off), p(cap), p(out_len),
                                         p(status), n, slen, _lib.MEM_HOST, -1 if fan != "0" else 0, None)
            assert rc == 0 and (status == 0).all()
            touched = np.zeros(out.size, dtype=bool)
            for i in range(n):
                o, k = int(out_off[i]), int(out_len[i])
                assert out[o : o + k].tobytes() == want.stream(i), (layout, fan, i)
                touched[o : o + k] = True
            assert (out[~touched] == 0xEE).all(), (layout, fan, "bytes outside the produced output were written")


def test_decoder_rejects_streams_the_bit_counters_cannot_hold(ta):
    """in_len >= 2^29: the one-shot batch decoders report TAMP_AMD_BAD_ARGUMENT for that stream and produce nothing; the
    neighbour in the same batch decodes normally."""
    import torch

    dev = torch.device("cuda", 0)
    good = ta.compress(b"foo foo foo")
    big = 1 << 29
    data = torch.zeros(big + 64, dtype=torch.uint8, device=dev)
    data[big : big + len(good)] = torch.frombuffer(bytearray(good), dtype=torch.uint8).to(dev)
    in_off = torch.tensor([0, big], dtype=torch.int64, device=dev)
    in_len = torch.tensor([big, len(good)], dtype=torch.int32, device=dev)
    for mode in ("wave", "lane", "global", "split"):
        os.environ["TAMP_AMD_DECODER"] = mode
        try:
            r = ta.decompress_batch(data, in_off, in_len.view(torch.int32), out_cap=64)
        finally:
            del os.environ["TAMP_AMD_DECODER"]
        st = r.status.cpu().numpy()
        assert int(st[0]) == -21 and int(r.out_len[0]) == 0 and int(r.in_consumed[0]) == 0, mode
        assert int(st[1]) == 2 and r.stream(1) == b"foo foo foo", mode


# --------------------------------------------------------------------------------------------------------------
# regression fixtures and bounded randomised differential runs (driver-observed, not prose)
# --------------------------------------------------------------------------------------------------------------
def test_fuzz_regression_fixtures(ta):
    for c in load_golden("fuzz_regressions.json")["cases"]:
        data = unb64(c["input"])
        for run_aware in (None, False, True):
            got = ta.compress_batch([data], run_aware=run_aware, **c["conf"])
            assert int(got.status[0]) == c["status"] and got.stream(0) == unb64(c["expected"]), (c["name"], run_aware)
        back = ta.decompress_batch([unb64(c["expected"])], out_cap=len(data) + 8)
        assert int(back.status[0]) == 2 and back.stream(0) == data, c["name"]


@pytest.mark.parametrize("tool,seconds", [("fuzz_gpu.py", 25), ("fuzz_stream_gpu.py", 15), ("fuzz_resume_gpu.py", 15),
                                          ("fuzz_encoder_resume_gpu.py", 15)])
def test_bounded_randomised_differential(ta, tool, seconds):
    """tools/fuzz_*.py for a bounded budget each: compress (all modes, both builds) and decompress (all three decoders),
    streaming scripts, decoder objects, encoder objects -- against the oracle / live reference objects."""
    if tool == "fuzz_encoder_resume_gpu.py":
        from oracle.checker import Ref

        if not Ref.available():
            pytest.skip("needs the prebuilt reference library (oracle/_ref)")
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, FUZZ_SEED=str(20260929))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds)], capture_output=True, text=True,
                         timeout=seconds * 6 + 240, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "ok" in out.stdout.lower(), out.stdout[-500:]


# --------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] and one GPU's share of configs[4] at FULL size: size-independent properties + sampled oracle
# --------------------------------------------------------------------------------------------------------------
def test_full_size_config4_decode_one_million_streams(ta, oracle):
    """configs[3]: 1,048,576 pre-compressed 4 KiB streams, windows 2^8..2^12 interleaved, decoded in ONE call.
    Properties: every status 2, every length 4096, consumed == stream length, sampled streams equal their text, the
    compressed sample equals the oracle's bytes, and a checksum of the whole output equals the checksum of the text."""
    import torch

    from tamp_amd import workloads as wl

    dev = torch.device("cuda", 0)
    n, L = 1 << 20, 4096
    per = n // 5 + 1
    wsel = torch.arange(n, device=dev) % 5 + 8
    olen = torch.zeros(n, dtype=torch.int64, device=dev)
    parts = {}
    for w in range(8, 13):
        ids = torch.nonzero(wsel == w).flatten()
        rows = wl.synth_text(len(ids), L, first_index=w * 1000003)
        off, ln = wl.csr_for_fixed(len(ids), L)
        r = ta.compress_batch(torch.from_numpy(rows.reshape(-1)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev),
                              torch.from_numpy(ln.astype(np.int32)).to(dev), window=w, max_in_len=L)
        assert bool((r.status == 0).all().item())
        k = 48  # the compressor's own parity at this window, sampled against the oracle
        want = oracle.compress_batch(rows[:k].reshape(-1), off[:k], ln[:k], window=w, literal=8, extended=True, threads=8)
        gs = _streams_of(r, k)
        assert all(gs[i] == want.stream(i) for i in range(k)), w
        olen[ids] = r.out_len.to(torch.int64)
        parts[w] = (ids, rows, r)
        assert len(ids) <= per
    in_off = torch.cumsum(olen, 0) - olen
    slab = torch.empty(int(olen.sum().item()) + 64, dtype=torch.uint8, device=dev)
    text_sum = 0
    for w, (ids, rows, r) in parts.items():
        lens, src, dst = r.out_len.to(torch.int64), r.out_off.to(torch.int64), in_off[ids]
        rep = torch.repeat_interleave(torch.arange(len(ids), device=dev), lens)
        within = torch.arange(int(lens.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
        slab[dst[rep] + within] = r.out[src[rep] + within]
        del rep, within
        text_sum += int(rows.astype(np.uint64).sum())
    C = L + 8  # a little room: a slab that is exactly full reports TAMP_OUTPUT_FULL, as the reference does
    d = ta.decompress_batch(slab, in_off, olen.to(torch.int32), out_cap=C, timing=True)
    assert bool((d.status == 2).all().item()) and bool((d.out_len == L).all().item())
    assert bool((d.in_consumed.to(torch.int64) == olen).all().item())
    assert int(d.out[: n * C].view(n, C)[:, :L].sum(dtype=torch.int64).item()) == text_sum  # checksum over all 4 GiB
    for w, (ids, rows, r) in parts.items():
        step = max(1, len(ids) // 300)
        sel = ids[::step][:300].tolist()
        for j, s in enumerate(sel):
            assert d.out[s * C : s * C + L].cpu().numpy().tobytes() == rows[j * step].tobytes(), (w, s)
    print(f"\nconfigs[3] full size: {n} streams, {int(olen.sum().item()) / 2**30:.2f} GiB in, 4 GiB out, decode {d.kernel_ms:.2f} ms "
          f"= {n * L / d.kernel_ms / 1e6:.1f} GB/s out")


def test_full_size_config5_one_gpu_share(ta, oracle):
    """configs[4], one GPU's share at 8 GPUs: 2,097,152 x 256 B telemetry messages, shared custom dictionary, window=8
    literal=7 extended (header 0x16); three messages carry a byte >= 0x80 -> TAMP_EXCESS_BITS exactly there; the first
    4,096 equal the oracle's bytes; everything decodes back to the input."""
    import torch

    from tamp_amd import workloads as wl

    dev = torch.device("cuda", 0)
    n, L = 1 << 21, 256
    dct = wl.telemetry_dictionary(bytes(ta.initialize_dictionary(256, literal=7)))
    rows = wl.telemetry(n, L).copy()
    bad_ids = [5, n // 3, n - 2]
    for b in bad_ids:
        rows[b, 40] = 0xC3
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t = torch.arange(n, dtype=torch.int64, device=dev) * L
    len_t = torch.full((n,), L, dtype=torch.int32, device=dev)
    r = ta.compress_batch(data, off_t, len_t, window=8, literal=7, dictionary=dct, max_in_len=L, timing=True)
    st = r.status.cpu().numpy()
    assert np.nonzero(st != 0)[0].tolist() == sorted(bad_ids) and all(int(st[b]) == -2 for b in bad_ids)
    assert int(r.out[0]) == 0x16
    off, ln = wl.csr_for_fixed(4096, L)
    want = oracle.compress_batch(rows[:4096].reshape(-1), off, ln, window=8, literal=7, dictionary=dct, threads=8)
    gs = _streams_of(r, 4096)
    assert all(gs[i] == want.stream(i) and int(st[i]) == int(want.status[i]) for i in range(4096))
    C = L + 8
    back = ta.decompress_batch(r.out, r.out_off, r.out_len, out_cap=C, dictionary=dct, timing=True)
    good = torch.ones(n, dtype=torch.bool, device=dev)
    good[torch.tensor(bad_ids, device=dev)] = False
    assert bool((back.status[good] == 2).all().item()) and bool((back.out_len[good] == L).all().item())
    assert bool((back.out[: n * C].view(n, C)[:, :L][good] == data.view(n, L)[good]).all().item())
    print(f"\nconfigs[4] share: {n} x {L} B compress {r.kernel_ms:.2f} ms = {n * L / r.kernel_ms / 1e6:.1f} GB/s in, "
          f"decode {back.kernel_ms:.2f} ms = {n * L / back.kernel_ms / 1e6:.1f} GB/s out")


# --------------------------------------------------------------------------------------------------------------
# boundary: tamp_compress_stream with bounded memory, progress callbacks
# --------------------------------------------------------------------------------------------------------------
def test_compress_stream_bounded_buffer_and_progress_callback(ta, oracle, monkeypatch):
    """tamp_compress_stream (compressor.h:338, compressor.c:891-955) on an input longer than its host buffer: fed buffer
    by buffer to ONE compressor object and flushed once -- the bytes of the one-shot call -- with the progress callback
    (common.h:184-210) fired once per buffer as (bytes consumed so far, 0); tamp_compressor_compress_cb reports
    (consumed, total) and a non-zero return aborts with that code."""
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

    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
    sz = C.POINTER(C.c_size_t)
    lib.tamp_compressor_init.restype = C.c_int8
    lib.tamp_compressor_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tamp_compress_stream.restype = C.c_int8
    lib.tamp_compress_stream.argtypes = [C.c_void_p] * 5 + [sz, sz, CB, C.c_void_p]
    lib.tamp_compressor_compress_cb.restype = C.c_int8
    lib.tamp_compressor_compress_cb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_void_p, C.c_size_t, sz, CB, C.c_void_p]
    mem_read = C.cast(lib.tamp_stream_mem_read, C.c_void_p)
    mem_write = C.cast(lib.tamp_stream_mem_write, C.c_void_p)
    monkeypatch.setenv("TAMP_AMD_STREAM_BUFFER_MB", "1")
    text = wl.synth_text(1, (2 << 20) + 70_001, first_index=77)[0].tobytes()  # three buffers: 1 MiB, 1 MiB, the rest
    for ext in (1, 0):
        st, want = oracle.compress(text, extended=bool(ext))
        conf = TampConf(window=10, literal=8, extended=ext)
        window, comp = (C.c_ubyte * 1024)(), (C.c_ubyte * 48)()
        assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
        src = (C.c_ubyte * len(text)).from_buffer_copy(text)
        dst = (C.c_ubyte * (len(text) + 4096))()
        rd, wr = MemReader(C.addressof(src), len(text), 0), MemWriter(C.addressof(dst), len(text) + 4096, 0)
        seen = []
        cb = CB(lambda user, done, total: (seen.append((done, total)), 0)[1])
        cin, cout = C.c_size_t(0), C.c_size_t(0)
        assert lib.tamp_compress_stream(comp, mem_read, C.byref(rd), mem_write, C.byref(wr), C.byref(cin), C.byref(cout), cb, None) == 0
        assert (cin.value, cout.value) == (len(text), len(want)) and bytes(dst[: wr.pos]) == want, ext
        assert seen == [(1 << 20, 0), (2 << 20, 0), (len(text), 0)]
    # callback of the object-level call: (consumed, total); a custom code in [100, 127] aborts and is passed through
    conf = TampConf(window=10, literal=8, extended=1)
    assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
    seen = []
    cb = CB(lambda user, done, total: (seen.append((done, total)), 101)[1])
    piece = text[:5000]
    out, w, k = (C.c_ubyte * 8192)(), C.c_size_t(0), C.c_size_t(0)
    buf = (C.c_ubyte * len(piece)).from_buffer_copy(piece)
    assert lib.tamp_compressor_compress_cb(comp, out, 8192, C.byref(w), buf, len(piece), C.byref(k), cb, None) == 101
    assert seen == [(5000, 5000)] and k.value == 5000


def test_bench_under_torch_distributed_run_two_ranks_one_device(ta):
    """The launch line the driver uses for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), two ranks sharing
    the one device of this box: gloo for the barrier / MAX only, totals summed over ranks, one JSON line from rank 0."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TAMP_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--streams", "4096"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["streams_total"] == 8192 and line["config"]["all_streams_ok"]
    assert "gloo" in line["config"]["parallelism"] and "cpu_baseline" not in line


def test_command_line_round_trip(ta, oracle, tmp_path):
    """python -m tamp_amd compress / decompress (tamp/cli/main.py:115-232): files and pipes, an undersized dictionary file
    (raw effective bytes copied to the end of the seeded default), empty input -> "No data provided."."""
    from tamp_amd import workloads as wl

    text = wl.synth_text(1, 20000, first_index=5)[0].tobytes()
    src, comp, back, dct = tmp_path / "in.txt", tmp_path / "out.tamp", tmp_path / "back.txt", tmp_path / "d.bin"
    src.write_bytes(text)
    dct.write_bytes(text[:300])
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a, **k: subprocess.run([sys.executable, "-m", "tamp_amd", *a], capture_output=True, timeout=300, env=env, cwd=ROOT, **k)  # noqa: E731
    r = run("compress", "-i", str(src), "-o", str(comp), "-w", "9", "--lazy-matching")
    assert r.returncode == 0, r.stderr
    assert comp.read_bytes() == oracle.compress(text, window=9, lazy_matching=True)[1]
    r = run("decompress", str(comp), str(back))
    assert r.returncode == 0 and back.read_bytes() == text
    r = run("compress", "-d", str(dct), "-w", "10", "-l", "7", input=text)  # stdin -> stdout, raw dictionary
    assert r.returncode == 0, r.stderr
    full = ta.initialize_dictionary(1024, literal=7)
    full[-300:] = text[:300]
    assert r.stdout == oracle.compress(text, window=10, literal=7, dictionary=bytes(full))[1]
    r2 = run("decompress", "-d", str(dct), "-l", "7", input=r.stdout)
    assert r2.returncode == 0 and r2.stdout == text
    empty = tmp_path / "empty"
    empty.write_bytes(b"")
    r = run("compress", "-i", str(empty))
    assert r.returncode == 1 and b"No data provided." in r.stderr


# --------------------------------------------------------------------------------------------------------------
# epoch cut at long runs of one byte (tamp_compress_kernel.hpp: the block ends just inside the first such run)
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("window,slen", [(8, 256), (8, 700), (10, 4096), (12, 6000)])
def test_long_runs_anywhere_in_the_block_match_oracle(ta, oracle, monkeypatch, window, slen):
    """Text with runs of 30..300 equal bytes dropped at random places (several per stream, some back to back, some at
    the very start / end, some repeated so that a match can cover them): extended format, run-cut threshold at its
    default, at its smallest and off -- the bytes may not depend on it."""
    import torch
    from tamp_amd import workloads as wl

    rng = np.random.default_rng(window * 1000 + slen)
    n = 384
    rows = wl.synth_text(n, slen, first_index=77).copy()
    for i in range(n):
        k = int(rng.integers(0, 5))
        for _ in range(k):
            ln = int(rng.integers(30, 300))
            at = int(rng.integers(0, slen))
            if rng.random() < 0.15:
                at = 0
            if rng.random() < 0.15:
                at = max(0, slen - ln)
            rows[i, at : at + ln] = rng.choice([0x20, 0x00, 0x3D, 0x61])
        if i % 7 == 0 and slen >= 700:  # the same run + tail twice: the second one can be matched instead of run-coded
            rows[i, 300:360] = 0x2D
            rows[i, 360:380] = rows[i, 100:120]
            rows[i, 500:560] = 0x2D
            rows[i, 560:580] = rows[i, 100:120]
    off, ln_ = wl.csr_for_fixed(n, slen)
    dev = torch.device("cuda", 0)
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
    len_t = torch.from_numpy(ln_.astype(np.int32)).to(dev)
    want = oracle.compress_batch(rows.reshape(-1), off, ln_, window=window, literal=8, extended=True, threads=8)
    for cut in (None, "2", "0"):
        if cut is None:
            monkeypatch.delenv("TAMP_AMD_CUT_RUN", raising=False)
        else:
            monkeypatch.setenv("TAMP_AMD_CUT_RUN", cut)
        for runs in ("0", "1"):
            monkeypatch.setenv("TAMP_AMD_RUNS", runs)
            r = ta.compress_batch(data, off_t, len_t, window=window, literal=8, extended=True, max_in_len=slen)
            gs = _streams_of(r, n)
            for i in range(n):
                assert gs[i] == want.stream(i), (cut, runs, i)


# --------------------------------------------------------------------------------------------------------------
# split decoder: the streams it hands on (lag list full, output beyond its LDS rows) still come back right
# --------------------------------------------------------------------------------------------------------------
def test_split_decoder_leftovers_go_to_the_wave_decoder(ta, oracle, monkeypatch):
    """One batch mixing ordinary text with (a) streams of a hundred RLE runs longer than 8 bytes -- every one a lag, more
    than the 48 the parse kernel lists per stream --, (b) a stream that decodes to more than 16,384 bytes, (c) streams
    whose output capacity cuts a run / an extended match short.  Forced split decoder and the launcher's own choice."""
    from tamp_amd import workloads as wl

    rng = np.random.default_rng(5)
    plain = [wl.synth_text(1, 5000, first_index=40 + i)[0].tobytes() for i in range(300)]
    laggy = []
    for i in range(6):
        parts = []
        for k in range(100):
            parts.append(bytes([65 + (k + i) % 20]) * int(rng.integers(10, 40)))
            parts.append(wl.synth_text(1, 64, first_index=1000 + 100 * i + k)[0].tobytes()[: int(rng.integers(3, 30))])
        laggy.append(b"".join(parts))
    big = [wl.synth_text(1, 40000, first_index=7)[0].tobytes()]
    datas = plain[:150] + laggy + big + plain[150:]
    comps = []
    for dta in datas:
        st, comp = oracle.compress(dta)
        assert st == 0
        comps.append(comp)
    for mode in ("split", None):
        if mode:
            monkeypatch.setenv("TAMP_AMD_DECODER", mode)
        else:
            monkeypatch.delenv("TAMP_AMD_DECODER", raising=False)
        for cap in (50000, 5000, 4990):
            res = ta.decompress_batch(comps, out_cap=cap)
            for i, comp in enumerate(comps):
                st, want, consumed = oracle.decompress(comp, cap=cap)
                assert (int(res.status[i]), res.stream(i)) == (st, want), (mode, cap, i)
                assert int(res.in_consumed[i]) == consumed, (mode, cap, i)
