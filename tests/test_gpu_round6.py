"""Round 6, GPU tier: what the round changed outside the long-stream decoder (tests/test_gpu_long_decode.py has that).

* Batch calls on a side stream: the tables and the output slab a call makes with torch come into being ON that stream
  (tamp_amd/batch.py) -- made on torch's current stream they raced the launch: eight decode calls on eight streams in a row
  returned wrong bytes, and a loop of them faulted the device.
* The one-compare index-entry layout serves every u32-entry default-parse build (tamp_compress_kernel.hpp, kEntV2): windows
  2^8 .. 2^14, short messages (512 buckets), block mode -- against the reference C / the oracle.
"""
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


def test_calls_on_side_streams_make_their_tables_on_those_streams(ta):
    import torch
    from tamp_amd import workloads as wl

    n, L, parts = 16384, 4096, 8
    dev = torch.device("cuda:0")
    rows = wl.synth_text(n, L)
    off, ln = wl.csr_for_fixed(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t, len_t = torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ln.astype(np.int32)).to(dev)
    want = ta.compress_batch(data, off_t, len_t, window=10, max_in_len=L)
    torch.cuda.synchronize()
    k = n // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for _ in range(3):  # (no synchronisation anywhere between the calls)
        comp = [ta.compress_batch(data, off_t[i * k:(i + 1) * k], len_t[i * k:(i + 1) * k], window=10, max_in_len=L, out_cap=int(want.out_off[1]),
                                  stream=s.cuda_stream) for i, s in enumerate(streams)]
        dec = [ta.decompress_batch(want.out, want.out_off[i * k:(i + 1) * k], want.out_len[i * k:(i + 1) * k], out_cap=L + 8,
                                   stream=s.cuda_stream) for i, s in enumerate(streams)]
    torch.cuda.synchronize()
    for i in range(parts):
        assert bool((comp[i].out_len == want.out_len[i * k:(i + 1) * k]).all())
        for j in (0, 1, k // 2, k - 1):
            assert comp[i].stream(j) == want.stream(i * k + j)
        assert bool((dec[i].status == 2).all())
        got = dec[i].out.view(k, L + 8)[:, :L].contiguous().cpu().numpy().tobytes()
        assert got == rows[i * k:(i + 1) * k].tobytes()


@pytest.mark.parametrize("window", [8, 9, 11, 12, 13, 14])
def test_one_compare_entry_layout_at_every_window(ta, checker, window):
    from tamp_amd import workloads as wl

    blob = wl.real_text("markup") + wl.real_text("python")
    streams = [blob[i * 6000:(i + 1) * 6000 + 777] for i in range(48)] + [blob[:40_000], b"ab" * 3000, bytes(5000)]
    for ext in (True, False):
        got = ta.compress_batch(streams, window=window, extended=ext)
        for i, s in enumerate(streams):
            st, out = checker.compress(s, window=window, extended=ext)[:2]
            assert int(got.status[i]) == st and got.stream(i) == out, (window, ext, i)
    # short messages: the one-wavefront build with 512 buckets
    msgs = [blob[i * 211:(i * 211) + 150 + (i % 90)] for i in range(600)]
    got = ta.compress_batch(msgs, window=window, literal=8)
    for i in (0, 1, 7, 123, 599):
        st, out = checker.compress(msgs[i], window=window)[:2]
        assert int(got.status[i]) == st and got.stream(i) == out
