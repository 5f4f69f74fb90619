"""Round 5, GPU tier: ONE long v1 stream spread over all workgroups (block mode, tamp_compress_kernel<.., BLOCKM>,
DESIGN.md section 3.10) against the reference C / the oracle -- through the batch call, the one-shot tamp.compress(), the
reference-named object -- and the throughput bar of VERDICT round 4, item 7."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tamp_amd

    return tamp_amd


@pytest.fixture(scope="module")
def checker():
    from oracle.checker import Oracle, Ref

    return Ref() if Ref.available() else Oracle()


def _text(n, kind):
    from tamp_amd import workloads as wl

    if kind == "synth":
        return wl.synth_text((n + 4095) // 4096, 4096).reshape(-1)[:n].copy()
    blob = wl.real_text(kind)
    return np.frombuffer((blob * ((n + len(blob) - 1) // len(blob)))[:n], dtype=np.uint8).copy()


def _want(checker, flat, **kw):
    return checker.compress_batch(flat, np.zeros(1, np.uint64), np.array([flat.size], np.uint32), **kw).stream(0)


@pytest.mark.parametrize("kind", ["prose", "python", "synth"])
def test_block_mode_matches_the_reference_around_block_boundaries(ta, checker, kind, monkeypatch):
    """One v1 stream of 256 KiB and more is matched block by block on all workgroups, the blocks' entry offsets and bit
    positions come from a serial scan, and every block ORs its bits into place (compressor.c:532-660 restated per block):
    bytes equal to the reference C for lengths on, one past and just short of a block boundary, and equal to what the
    one-workgroup batch kernel produces for the same call (TAMP_AMD_BLOCK_MIN=0 switches block mode off)."""
    for n in (262144, 262145, 262144 + 1023, 300001, (1 << 20) + 777):
        flat = _text(n, kind)
        want = _want(checker, flat, window=10, literal=8, extended=False)
        got = ta.compress_batch([flat.tobytes()], window=10, literal=8, extended=False)
        assert int(got.status[0]) == 0 and got.stream(0) == want, (kind, n)
    monkeypatch.setenv("TAMP_AMD_BLOCK_MIN", "0")
    assert ta.compress_batch([flat.tobytes()], window=10, literal=8, extended=False).stream(0) == want


def test_block_mode_windows_dictionary_and_tight_room(ta, checker):
    """Every window the packed index serves (2^8 .. 2^14; 2^15 and the other formats stay with the one-workgroup kernel and
    must still be right), a custom dictionary, and output room that ends inside the stream: TAMP_OUTPUT_FULL with the
    exact prefix, as tamp_compressor_compress_and_flush leaves it (compressor.c:65-75)."""
    flat = _text(300001, "prose")
    for w in (8, 9, 11, 12, 13, 14, 15):
        want = _want(checker, flat, window=w, literal=8, extended=False)
        got = ta.compress_batch([flat.tobytes()], window=w, literal=8, extended=False)
        assert int(got.status[0]) == 0 and got.stream(0) == want, w
    for kw in (dict(extended=True), dict(extended=False, literal=7), dict(extended=False, lazy_matching=True)):
        ascii_only = flat & 0x7F
        ckw = {("lazy" if k == "lazy_matching" else k): v for k, v in kw.items()}  # (the checker's name for it)
        want = _want(checker, ascii_only, window=10, **{"literal": 8, **ckw})
        assert ta.compress_batch([ascii_only.tobytes()], window=10, **{"literal": 8, **kw}).stream(0) == want, kw
    dic = bytes(range(256)) * 4
    want = _want(checker, flat, window=10, literal=8, extended=False, dictionary=dic)
    assert ta.compress_batch([flat.tobytes()], window=10, literal=8, extended=False, dictionary=dic).stream(0) == want
    want = _want(checker, flat, window=10, literal=8, extended=False)
    for cap in (len(want), len(want) - 1, 4097, 1000, 3, 1):
        g = ta.compress_batch([flat.tobytes()], window=10, literal=8, extended=False, out_cap=cap)
        assert int(g.status[0]) == (0 if cap >= len(want) else 1)
        assert g.stream(0) == want[: min(cap, len(want))], cap


def test_block_mode_on_device_tensors_at_odd_output_offsets(ta, checker):
    """Device-resident call: the stream's slab starts 1, 2, 3 bytes off a dword inside the caller's output tensor; the
    emitters' aligned dword stores / atomics must not touch a byte in front of it or behind the stream's last byte."""
    import torch

    dev = torch.device("cuda:0")
    flat = _text(400000, "markup")
    want = _want(checker, flat, window=10, literal=8, extended=False)
    d = torch.from_numpy(flat).to(dev)
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    ln = torch.tensor([flat.size], dtype=torch.int32, device=dev)
    lib_conf = dict(window=10, literal=8, extended=False, max_in_len=flat.size)
    for shift in (1, 2, 3):
        from tamp_amd import _lib
        from tamp_amd.batch import _conf, _ptr

        lib = _lib.load()
        cap = len(want) + 5
        out = torch.full((cap + 64,), 0xEE, dtype=torch.uint8, device=dev)
        out_off = torch.tensor([shift], dtype=torch.int64, device=dev)
        out_cap = torch.tensor([cap], dtype=torch.int32, device=dev)
        out_len = torch.zeros(1, dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int8, device=dev)
        conf = _conf(10, 8, False, None, False, False, None)
        rc = lib.tamp_batch_compress(C.byref(conf), None, _ptr(d), _ptr(off), _ptr(ln), _ptr(out), _ptr(out_off), _ptr(out_cap),
                                     _ptr(out_len), _ptr(status), 1, flat.size, _lib.MEM_DEVICE, 0,
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        assert int(status[0]) == 0 and int(out_len[0]) == len(want)
        assert host[shift: shift + len(want)].tobytes() == want
        assert (host[:shift] == 0xEE).all()  # nothing in front of the slab
        # (behind the stream's last byte the slab was zero-filled up to its capacity; behind the slab: untouched)
        assert (host[shift + cap:] == 0xEE).all()
    del lib_conf


def test_one_shot_compress_and_reference_named_object_take_block_mode(ta, checker):
    """tamp.compress(data, extended=False) of 256 KiB and more is the batch call on one stream; a fresh reference-named
    object handed a whole v1 stream in one tamp_compressor_compress_and_flush call takes the same route and afterwards
    holds the window the reference's object would hold: the next segment on the same object continues the stream."""
    from tamp_amd import _lib

    flat = _text(700003, "prose")
    want = _want(checker, flat, window=10, literal=8, extended=False)
    assert bytes(ta.compress(flat.tobytes(), extended=False)) == want
    assert bytes(ta.compress(flat.tobytes(), window=12, extended=False)) == _want(checker, flat, window=12, literal=8, extended=False)

    lib = _lib.load()

    class TampConf(C.Structure):
        _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                    ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1),
                    ("lazy_matching", C.c_uint16, 1)]

    lib.tamp_compressor_init.restype = C.c_int8
    lib.tamp_compressor_compress_and_flush.restype = C.c_int8
    lib.tamp_compressor_compress_and_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                                       C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_bool]
    conf = TampConf(window=10, literal=8, extended=0)
    window = (C.c_ubyte * 1024)()
    c = (C.c_ubyte * 48)()
    assert lib.tamp_compressor_init(c, C.byref(conf), window) == 0
    out = (C.c_ubyte * (len(want) + 4096))()
    written, consumed = C.c_size_t(0), C.c_size_t(0)
    data = flat.tobytes()
    src = (C.c_ubyte * len(data)).from_buffer_copy(data)
    r = lib.tamp_compressor_compress_and_flush(c, out, len(out), C.byref(written), src, len(data), C.byref(consumed), False)
    assert (r, consumed.value) == (0, len(data)) and bytes(out[: written.value]) == want
    tail = _text(5000, "python").tobytes()
    src2 = (C.c_ubyte * len(tail)).from_buffer_copy(tail)
    r = lib.tamp_compressor_compress_and_flush(c, out, len(out), C.byref(written), src2, len(tail), C.byref(consumed), False)
    from oracle.checker import Oracle

    st2, both = Oracle().stream_script([("write", data), ("flush", False), ("write", tail), ("flush", False)],
                                       window=10, literal=8, extended=False)
    assert (r, st2, consumed.value) == (0, 0, len(tail)) and both[: len(want)] == want
    assert bytes(out[: written.value]) == both[len(want):]


def test_a_100_000_000_byte_v1_stream_at_a_gigabyte_per_second(ta, checker):
    """`tamp_amd_compress` of a 100,000,000-byte v1 stream >= 1 GB/s, bit-exact against the reference
    C (a frozen corpus tiled to enwik8's length: the file itself is not on the box).  Wall clock of the host-memory call --
    copies in and out included -- and the kernels' own time on device-resident data."""
    import time

    import torch

    from tamp_amd import _lib

    flat = _text(100_000_000, "prose")
    want = _want(checker, flat, window=10, literal=8, extended=False)
    lib = _lib.load()
    lib.tamp_amd_compress.restype = C.c_int8
    lib.tamp_amd_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.c_int]
    from tamp_amd.batch import _conf

    conf = _conf(10, 8, False, None, False, False, None)
    out = np.zeros(len(want) + 4096, dtype=np.uint8)
    written = C.c_size_t(0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = lib.tamp_amd_compress(C.byref(conf), None, out.ctypes.data, out.size, C.byref(written), flat.ctypes.data, flat.size, 0)
        best = min(best, time.perf_counter() - t0)
        assert r == 0 and written.value == len(want)
    assert out[: written.value].tobytes() == want
    assert flat.size / best / 1e9 >= 1.0, f"tamp_amd_compress (host memory): {flat.size / best / 1e9:.2f} GB/s"
    dev = torch.device("cuda:0")
    d = torch.from_numpy(flat).to(dev)
    ms = []
    for _ in range(3):
        g = ta.compress_batch(d, torch.zeros(1, dtype=torch.int64, device=dev), torch.tensor([flat.size], dtype=torch.int32, device=dev),
                              window=10, literal=8, extended=False, max_in_len=flat.size, timing=True)
        ms.append(float(g.kernel_ms))
    assert g.stream(0) == want
    assert flat.size / (min(ms) * 1e-3) / 1e9 >= 4.0, f"device-resident: {min(ms):.1f} ms"
    print(f"100,000,000-byte v1 stream: host call {flat.size / best / 1e9:.2f} GB/s, kernels {min(ms):.2f} ms = {flat.size / min(ms) / 1e6:.2f} GB/s")


def test_expensive_streams_first_changes_the_schedule_not_the_results(ta, checker, monkeypatch):
    """Round 5: batches of a few rounds of the persistent grid are claimed most-expensive-first (score = aligned dwords of
    four equal bytes; the compress kernel reads gathered tables, sizes and statuses are scattered back).  Same bytes, sizes
    and statuses per stream with the ordering forced on and off -- on the configs[2] stand-in's first 3,052 streams (one
    GPU's share at eight) and on a ragged batch with empty streams and output room that ends inside some of them."""
    from tamp_amd import workloads as wl

    rows = wl.standin_rows(3052, 4096)
    off, ln = wl.csr_for_fixed(len(rows), 4096)
    flat = rows.reshape(-1)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TAMP_AMD_LPT", mode)
        r = ta.compress_batch(flat, off, ln, window=10, literal=8, extended=True, max_in_len=4096)
        got[mode] = ([r.stream(i) for i in range(len(ln))], r.status.copy(), r.out_len.copy())
    assert got["0"][0] == got["1"][0] and (got["0"][1] == got["1"][1]).all() and (got["0"][2] == got["1"][2]).all()
    k = 300
    want = checker.compress_batch(flat[: k * 4096], off[:k], ln[:k], window=10, literal=8, extended=True)
    assert all(got["1"][0][i] == want.stream(i) for i in range(k))
    # ragged: lengths 0 .. 9,000, every seventh stream with room for only 100 bytes
    rng = np.random.default_rng(7)
    lens = rng.integers(0, 9000, 2500).astype(np.uint32)
    lens[::97] = 0
    offs = np.zeros(len(lens), np.uint64)
    offs[1:] = np.cumsum(lens[:-1].astype(np.uint64))
    blob = np.frombuffer((wl.real_text("python") * 8)[: int(lens.sum())], dtype=np.uint8)
    caps = np.array([100 if i % 7 == 0 else int(n) * 9 // 8 + 16 for i, n in enumerate(lens)], dtype=np.uint32)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TAMP_AMD_LPT", mode)
        r = ta.compress_batch(blob, offs, lens, window=10, literal=8, extended=True, out_cap=caps)
        res[mode] = ([r.stream(i) for i in range(len(lens))], r.status.copy(), r.out_len.copy())
    assert res["0"][0] == res["1"][0] and (res["0"][1] == res["1"][1]).all() and (res["0"][2] == res["1"][2]).all()
    wantr = checker.compress_batch(blob, offs, lens, out_cap=caps, window=10, literal=8, extended=True)
    # (streams that ran out of room: the status is the reference's; how many bytes of the cut-off stream are delivered is
    # pinned elsewhere -- tests/test_gpu_parity.py::test_restricted_output)
    assert (res["1"][1] == wantr.status).all()
    assert all(res["1"][0][i] == wantr.stream(i) for i in range(len(lens)) if wantr.status[i] == 0 and i % 5 == 0)


def test_block_mode_calls_in_flight_on_two_streams_keep_their_own_tables(ta, checker):
    """Two long v1 streams compressed on two HIP streams without a wait in between: every call's per-block tables live in a
    scratch buffer of ITS stream (a shared one would be overwritten by the second call while the first still runs)."""
    import torch

    dev = torch.device("cuda:0")
    a, b = _text(3_000_001, "prose"), _text(2_500_003, "python")
    wa, wb = (_want(checker, x, window=10, literal=8, extended=False) for x in (a, b))
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        for d, st in ((da, s1), (db, s2)):
            res.append(ta.compress_batch(d, torch.zeros(1, dtype=torch.int64, device=dev), torch.tensor([d.numel()], dtype=torch.int32, device=dev),
                                         window=10, literal=8, extended=False, max_in_len=int(d.numel()), stream=st.cuda_stream))
    torch.cuda.synchronize()
    for i, r in enumerate(res):
        assert r.stream(0) == (wa if i % 2 == 0 else wb), i


def test_a_handful_of_long_v1_streams_take_block_mode_one_after_the_other(ta, checker):
    """Up to 64 streams per call, every one of them 256 KiB or more: each is spread over all workgroups in turn (one call,
    no batch of one-workgroup streams); with a short one among them the batch kernel takes the whole call.  Same bytes."""
    parts = [_text(n, k) for n, k in ((300_000, "prose"), (1_234_567, "python"), (262_144, "synth"), (700_001, "markup"))]
    want = [_want(checker, p, window=10, literal=8, extended=False) for p in parts]
    got = ta.compress_batch([p.tobytes() for p in parts], window=10, literal=8, extended=False)
    assert [int(x) for x in got.status] == [0] * 4 and [got.stream(i) for i in range(4)] == want
    mixed = parts + [_text(5000, "prose")]
    want.append(_want(checker, mixed[-1], window=10, literal=8, extended=False))
    got = ta.compress_batch([p.tobytes() for p in mixed], window=10, literal=8, extended=False)
    assert [got.stream(i) for i in range(5)] == want
