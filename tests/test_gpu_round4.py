"""Round-4 GPU tests: the eight-shard launch paths at the bench's FULL shard size (65,536 streams of 4 KiB per shard, all
shards on this box's one device), the third frozen corpus (markup), and the bench's corpus probe as invoked.  All `-m gpu`."""
import ctypes as C
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


def _bench(args, env_extra, timeout=1200):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_eight_full_size_shards_one_process_as_invoked():
    """`TAMP_BENCH_ONE_DEVICE=1 python bench.py --gpus 8` at the DEFAULT 65,536 streams per shard (8 x 256 MiB of input +
    8 output slabs on one device): the step's wall clock stays within 3 % of the sum of the eight kernels' own times --
    nothing on the host (allocation, table fills, Python per shard) sits between the launches -- and every stream of every
    shard reports TAMP_OK.  No scaling figure follows from this: eight shards on ONE device measure the launch path only."""
    line = _bench(["--gpus", "8", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], {"TAMP_BENCH_ONE_DEVICE": "1"})
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["streams_total"] == 8 * 65536 and cfg["all_streams_ok"]
    assert line["scaling"] == "weak" and "configs[1]" in cfg["workload"]
    assert line["ms_per_step"] <= 1.03 * cfg["sum_kernel_ms_per_step"], (line["ms_per_step"], cfg)
    assert 0 < cfg["host_launch_us_per_shard"] < 300, cfg
    assert cfg["corpus_probe"]["found"] is None and cfg["corpus_probe"]["tried"] == []  # (the probe is opt-in: --probe-corpus)


def test_eight_full_size_shards_keep_their_own_slabs_and_bytes(ta, checker):
    """The bench's Shard objects, eight of them on cuda:0 at 65,536 streams each, three interleaved steps: every shard reuses
    ITS OWN output slab and tables from step to step (no allocation per launch, no sharing between shards), launches on
    its own stream, and its bytes -- a sample of 96 streams per shard, first / middle / last -- are the reference's."""
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from tamp_amd import workloads as wl

    dev = torch.device("cuda", 0)
    n, L = 65536, 4096
    shards = []
    for r in range(8):
        rows = wl.synth_text(n, L, first_index=r * n)
        off, ln = wl.csr_for_fixed(n, L)
        sh = bench.Shard(torch, r, dev, rows.reshape(-1), off, ln, L, dict(window=10, literal=8, extended=True))
        sh.sample = np.concatenate([np.arange(32), np.arange(n // 2, n // 2 + 32), np.arange(n - 32, n)])
        sh.want = checker.compress_batch(np.ascontiguousarray(rows[sh.sample]).reshape(-1), *wl.csr_for_fixed(96, L),
                                         window=10, literal=8, extended=True, threads=8)
        shards.append(sh)
        del rows
    ptrs = []
    for step in range(3):
        res = [sh.launch(record=True) for sh in shards]
        for sh in shards:
            sh.sync()
        ptrs.append([(r.out.data_ptr(), r.out_len.data_ptr(), r.status.data_ptr()) for r in res])
    assert ptrs[1] == ptrs[2], "a shard allocated a new slab or new tables between two steps"
    assert len({p[0] for p in ptrs[2]}) == 8 and len({sh.stream.cuda_stream for sh in shards}) == 8
    for sh, r in zip(shards, res):
        assert bool((r.status == 0).all().item()), sh.index
        for k, i in enumerate(sh.sample):
            assert r.stream(int(i)) == sh.want.stream(k), (sh.index, int(i))


def test_all_devices_c_abi_eight_host_threads_full_size(ta, checker, monkeypatch):
    """`device = TAMP_AMD_ALL_DEVICES` through the C ABI with host buffers: 8 x 65,536 streams of 4 KiB (2 GiB) cut into
    eight contiguous shards, a host thread each (TAMP_AMD_FANOUT=8 maps them onto this box's one device).  Every status is
    TAMP_OK, a strided sample of 2,048 streams equals the reference, and the whole output decodes back to the input."""
    from tamp_amd import _lib, workloads as wl

    lib = _lib.load()
    n, L = 8 * 65536, 4096
    rows = wl.synth_text(n, L)
    off, ln = wl.csr_for_fixed(n, L)
    capv = ta.compress_bound(L, 8)
    cap = np.full(n, capv, np.uint32)
    out_off = np.arange(n, dtype=np.uint64) * np.uint64(capv)
    out = np.zeros(n * capv, np.uint8)
    out_len = np.zeros(n, np.uint32)
    status = np.full(n, -99, np.int8)
    conf = _lib.TampAmdConf(window=10, literal=8, extended=1)
    monkeypatch.setenv("TAMP_AMD_FANOUT", "8")
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.tamp_batch_compress(C.byref(conf), None, p(rows), p(off), p(ln), p(out), p(out_off), p(cap), p(out_len),
                                 p(status), n, L, _lib.MEM_HOST, -1, None)
    assert rc == 0 and (status == 0).all()
    pick = np.arange(0, n, n // 2048)
    want = checker.compress_batch(np.ascontiguousarray(rows[pick]).reshape(-1), *wl.csr_for_fixed(len(pick), L), window=10,
                                  literal=8, extended=True, threads=8)
    for k, i in enumerate(pick):
        o = int(out_off[i])
        assert out[o : o + int(out_len[i])].tobytes() == want.stream(k), int(i)
    # shard boundaries: the streams either side of every cut
    for b in range(1, 8):
        for i in (b * 65536 - 1, b * 65536):
            w = checker.compress_batch(rows[i], np.zeros(1, np.uint64), np.array([L], np.uint32), window=10, literal=8, extended=True)
            o = int(out_off[i])
            assert out[o : o + int(out_len[i])].tobytes() == w.stream(0), i
    for b in range(8):  # (decoded back shard by shard: bounded host memory)
        sl = slice(b * 65536, (b + 1) * 65536)
        lo = int(out_off[sl][0])
        hi = int(out_off[sl][-1]) + capv
        back = ta.decompress_batch(out[lo:hi], out_off[sl] - np.uint64(lo), out_len[sl], out_cap=L + 8)
        assert (np.asarray(back.status) == 2).all() and (np.asarray(back.out_len) == L).all(), b
        assert (np.asarray(back.out)[: 65536 * (L + 8)].reshape(65536, L + 8)[:, :L] == rows[sl]).all(), b


def test_markup_corpus_matches_reference(ta, checker):
    """The third frozen corpus (tests/golden/corpus_markup.txt.xz: HTML of the GDB manual, the closest thing to wiki
    markup on the image): all 768 chunks of 4 KiB, both formats, plus one 3 MiB stream, byte for byte."""
    from tamp_amd import workloads as wl

    blob = wl.real_text("markup")
    flat, off, ln = wl.split_fixed(blob, 4096)
    for ext in (True, False):
        want = checker.compress_batch(flat, off, ln, window=10, literal=8, extended=ext, threads=8)
        got = ta.compress_batch(flat, off, ln, window=10, literal=8, extended=ext, max_in_len=4096)
        for i in range(len(ln)):
            assert got.stream(i) == want.stream(i), (ext, i)
    one = np.frombuffer(blob, dtype=np.uint8)
    w1 = checker.compress_batch(one, np.zeros(1, np.uint64), np.array([len(blob)], np.uint32), window=10, literal=8, extended=True)
    g1 = ta.compress_batch(one, np.zeros(1, np.uint64), np.array([len(blob)], np.uint32), window=10, literal=8, extended=True)
    assert g1.stream(0) == w1.stream(0)


def test_bench_corpus_probe_is_opt_in(tmp_path):
    """a default `python bench.py` neither looks for enwik8 nor touches the network -- its headline is
    configs[1] whatever lies on the box; `--probe-corpus` (no download without --fetch-corpus) says what it tried in
    config.corpus_probe, and a find would be measured under also.configs2_corpus without changing the headline."""
    line = _bench(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {})
    probe = line["config"]["corpus_probe"]
    assert probe["found"] is None and probe["tried"] == [] and probe["fetch"] == "not attempted"
    assert "configs[1]" in line["config"]["workload"] and line["scaling"] == "weak"
    line = _bench(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--probe-corpus"], {})
    probe = line["config"]["corpus_probe"]
    assert probe["found"] is None and any(t.endswith("/enwik8") for t in probe["tried"]) and probe["fetch"] == "not attempted"
    assert "configs[1]" in line["config"]["workload"]


def test_bench_strong_scaling_standin_as_invoked():
    """`bench.py --corpus-standin [--shard-fraction F]` runs configs[2]'s shape (24,414 x 4 KiB of
    real text, strong scaling) on the frozen corpora; one GPU's share at 8 GPUs (3,052 streams) is timed as invoked."""
    whole = _bench(["--corpus-standin", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], {})
    assert whole["scaling"] == "strong" and whole["config"]["streams_total"] == 24414 and whole["config"]["all_streams_ok"]
    assert "STAND-IN" in whole["config"]["workload"]
    part = _bench(["--corpus-standin", "--shard-fraction", "8", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], {})
    assert 3040 <= part["config"]["streams_total"] <= 3064 and part["config"]["all_streams_ok"]
    # an eighth of the work -- but NOT an eighth of the time: a 3,052-stream batch is 1.7 rounds of the 1,792-workgroup grid
    # and no faster than its slowest stream (one workgroup per stream: 1.4 ms for the heaviest chunk of Python source,
    # tools/solo_latency.py).  Measured 3.1-3.3 x; the bench line carries the figure (also.strong_scaling_standin).
    assert part["ms_per_step"] < whole["ms_per_step"] / 2.5


def test_build_dictionary_cli_end_to_end(tmp_path, ta):
    """`python -m tamp_amd build-dictionary` as invoked (tamp/cli/build_dictionary.py:706-927) on 4,000
    telemetry messages, every whole-corpus evaluation a GPU batch launch; then `compress -d` / `decompress -d` with the
    file it wrote: the messages come back, and they are smaller than without the dictionary."""
    from tamp_amd import cli, workloads as wl

    rows = wl.telemetry(4000, 256)
    corpus = [bytes(r).rstrip(b" ") for r in rows]
    src = tmp_path / "messages.txt"
    src.write_bytes(b"\n".join(corpus))
    out = tmp_path / "dictionary.bin"
    p = subprocess.run([sys.executable, "-m", "tamp_amd", "build-dictionary", str(src), "-o", str(out), "-w", "8", "-l", "7"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "With dict:" in p.stderr and "<-- selected" in p.stderr
    blob = out.read_bytes()
    assert 0 < len(blob) <= 256
    window = cli.load_dictionary(out, 8, 7, True)
    plain = sum(len(ta.compress(s, window=8, literal=7)) for s in corpus[:200])
    with_d = sum(len(ta.compress(s, window=8, literal=7, dictionary=bytearray(window))) for s in corpus[:200])
    assert with_d < 0.8 * plain, (plain, with_d)
    msg = tmp_path / "one.bin"
    msg.write_bytes(corpus[7])
    assert cli.main(["compress", str(msg), str(tmp_path / "one.tamp"), "-w", "8", "-l", "7", "-d", str(out)]) == 0
    # (a raw, undersized dictionary file needs the stream's window / literal on the decompress side too, as in the
    # reference: tamp/cli/main.py:208-216)
    assert cli.main(["decompress", str(tmp_path / "one.tamp"), str(tmp_path / "one.out"), "-w", "8", "-l", "7", "-d", str(out)]) == 0
    assert (tmp_path / "one.out").read_bytes() == corpus[7]


def test_short_messages_through_the_split_decoder(ta, monkeypatch):
    """Round 4: batches of short messages take the split decoder -- the parse keeps the whole compressed message in its
    ring, RESOLVE runs one WAVEFRONT per stream (out_cap up to 2 KiB; four bytes per thread above it as well).  Telemetry
    messages (window 2^8 + shared custom dictionary, the BASELINE configs[4] shape; default window 2^10), short text,
    messages of 1..40 bytes, truncated streams, and every kind of restricted output -- status, bytes and consumed count
    against the oracle's decoder, forced (`split`) and as the launcher picks (`auto`)."""
    import random

    from oracle.checker import Oracle
    from tamp_amd import workloads as wl

    oracle = Oracle()
    rng = random.Random(41)
    tel = wl.telemetry(600, 256)
    d8 = wl.telemetry_dictionary(bytes(ta.initialize_dictionary(256, literal=7)))
    groups = []  # (streams, dictionary, max_window_bits)
    comp = [oracle.compress(tel[i].tobytes(), window=8, literal=7, dictionary=d8)[1] for i in range(300)]
    groups.append((comp + [c[: rng.randrange(1, len(c))] for c in comp[:40]], d8, 15))
    comp = [oracle.compress(tel[300 + i].tobytes())[1] for i in range(300)]  # default window, seeded dictionary
    groups.append((comp + [c[: rng.randrange(1, len(c))] for c in comp[:40]], None, 15))
    short = [wl.synth_text(1, rng.randrange(1, 41), first_index=i)[0].tobytes() for i in range(200)]
    text = [wl.synth_text(1, rng.randrange(200, 2000), first_index=500 + i)[0].tobytes() for i in range(100)]
    runs = [wl.lcg_runs(1, rng.randrange(100, 1500), first_index=i)[0].tobytes() for i in range(60)]  # RLE / extended tokens, lags
    groups.append(([oracle.compress(x, window=9)[1] for x in short + text + runs], None, 15))
    groups.append(([oracle.compress(x, window=10, extended=False)[1] for x in text[:60] + runs[:30]], None, 10))
    checked = 0
    for mode in ("split", "auto"):
        if mode == "split":
            monkeypatch.setenv("TAMP_AMD_DECODER", "split")
        else:
            monkeypatch.delenv("TAMP_AMD_DECODER", raising=False)
        for streams, d, mwb in groups:
            for cap in (2048, 600, 264, 256, 255, 100, 17, 1, 0):
                res = ta.decompress_batch(streams, out_cap=cap, dictionary=d, max_window_bits=mwb)
                for i, c in enumerate(streams):
                    want = oracle.decompress(c, dictionary=d, cap=cap, max_window_bits=mwb)
                    assert (int(res.status[i]), res.stream(i), int(res.in_consumed[i])) == want, (mode, cap, i, len(c))
                    checked += 1
            # ragged caps in one batch (the largest decides RESOLVE's build)
            caps = np.array([rng.randrange(0, 1200) for _ in streams], dtype=np.uint32)
            res = ta.decompress_batch(streams, out_cap=caps, dictionary=d, max_window_bits=mwb)
            for i, c in enumerate(streams):
                want = oracle.decompress(c, dictionary=d, cap=int(caps[i]), max_window_bits=mwb)
                assert (int(res.status[i]), res.stream(i), int(res.in_consumed[i])) == want, (mode, "ragged", i)
    assert checked > 20000
