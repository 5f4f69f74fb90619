#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W            one process drives N devices, one hipStream each
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
                                                             one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE)

Workload, default (BASELINE.json configs[1]): per GPU 65,536 independent 4 KiB synthetic-text
streams, window=10 literal=8, library-default extended=1, batch-compressed by the HIP kernels with the inputs already
resident in HBM.  One "step" = one batch launch over all streams of every shard.  Streams shard embarrassingly (weak
scaling, NO collective on the data path, RCCL is never initialised): shard r owns stream indices [r*65536, (r+1)*65536).
With --corpus PATH (or $TAMP_CORPUS) the workload is configs[2]: the file cut into independent 4 KiB streams (the short
tail kept as a last stream), the streams cut into N contiguous ranges balanced by bytes (strong scaling); every stream
is checked against the reference C, and a 100,000,000-byte file is also compressed as ONE stream and compared with the
reference's whole-file pins for enwik8 (/root/reference/tests/test_dataset_regression.py:38-43, README.md:266).
Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_TAG = "r6"     # profiles/<tag>_pmc_{fetch,write,sq,sq2}_counter_collection.csv feed roofline.traffic / valu_*


class Shard:
    """One contiguous range of streams resident on one device, with its own HIP stream."""

    def __init__(self, torch, index, device, flat, in_off, in_len, max_len, conf_kw):
        import numpy as np
        import tamp_amd

        self.index, self.device = index, device
        self.n = int(len(in_len))
        self.in_bytes = int(np.asarray(in_len, dtype=np.uint64).sum())
        self.max_len = int(max_len)
        with torch.cuda.device(device):
            self.stream = torch.cuda.Stream(device=device)
            self.data = torch.from_numpy(np.ascontiguousarray(flat)).to(device)
            self.off = torch.from_numpy(np.asarray(in_off, dtype=np.int64)).to(device)
            self.len = torch.from_numpy(np.asarray(in_len, dtype=np.int32)).to(device)
            self.cap = tamp_amd.compress_bound(self.max_len, 8)  # uniform slabs: the launch needs no device round trip
        self.kw = dict(conf_kw, max_in_len=self.max_len, out_cap=self.cap)
        self.torch = torch
        self.events = []
        self.last = None       # the previous step's result: its slab and tables are the next step's workspace
        self.host_s = 0.0      # host time spent inside launch() (enqueue cost of this shard), timed region only
        self.launches = 0

    def launch(self, record=False, **over):
        """Enqueue one batch compress on this shard's stream (asynchronous); optionally bracket it with HIP events."""
        import tamp_amd

        torch = self.torch
        t0 = time.perf_counter()
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            res = tamp_amd.compress_batch(self.data, self.off, self.len, stream=self.stream.cuda_stream,
                                          reuse=None if over else self.last, **dict(self.kw, **over))
            if record:
                e1.record(self.stream)
                self.events.append((e0, e1))
        if not over:
            self.last = res
        if record:
            self.host_s += time.perf_counter() - t0
            self.launches += 1
        return res

    def sync(self):
        self.stream.synchronize()


def host_threads():
    """CPUs this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, quota) if quota else n), n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (synthetic workload)")
    ap.add_argument("--stream-len", type=int, default=4096)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--extended", type=int, default=1)
    ap.add_argument("--corpus", default=os.environ.get("TAMP_CORPUS"), help="file to cut into 4 KiB streams (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="timed loop only (profiling runs)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="take roofline.traffic / valu_* from the committed passes instead of profiling a short run now")
    ap.add_argument("--cpu-sample", type=int, default=32768, help="streams timed on the host cores")
    ap.add_argument("--probe-corpus", action="store_true",
                    help="look for enwik8 in ./ ~/ /data /tmp (plain or .zip); if found it is measured under also.configs2_corpus -- "
                         "the headline stays configs[1] (only --corpus / $TAMP_CORPUS switch the headline workload)")
    ap.add_argument("--fetch-corpus", action="store_true",
                    help="with --probe-corpus: one guarded download attempt of the URL the reference's Makefile uses (rank 0 only)")
    ap.add_argument("--no-corpus-probe", action="store_true", help=argparse.SUPPRESS)  # (round-4 flag: the probe is opt-in now)
    ap.add_argument("--corpus-standin", action="store_true",
                    help="headline workload = the 24,414-stream real-text stand-in for configs[2] (strong scaling over --gpus)")
    ap.add_argument("--shard-fraction", type=int, default=1,
                    help="with --corpus-standin on one GPU: time only the first 1/F of the streams (one GPU's share at F GPUs)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import tamp_amd
    from tamp_amd import partition_streams
    from tamp_amd import workloads as wl

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = env_world > 1  # one rank per GPU under torch.distributed.run
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    world = env_world if launched else max(1, args.gpus)
    if launched and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    one_device = os.environ.get("TAMP_BENCH_ONE_DEVICE") == "1"  # every shard on cuda:0 (launch-path test on a 1-GPU box)
    ndev = torch.cuda.device_count()
    if not one_device and not launched and world > ndev:
        raise SystemExit(f"--gpus {world} but only {ndev} device(s) visible (TAMP_BENCH_ONE_DEVICE=1 maps every shard to cuda:0)")
    dist = None
    if launched:
        # The data path needs no collective: the process group exists for the barrier and the MAX of elapsed time only,
        # over gloo (CPU); RCCL is never initialised.
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    my_shards = [rank] if launched else list(range(world))

    def device_of(r):
        if one_device:
            return torch.device("cuda", 0)
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if launched else r)

    conf_kw = dict(window=args.window, literal=8, extended=bool(args.extended))
    # The metric is quoted on enwik8 (BASELINE.json), which neither image holds.  The headline workload NEVER depends on
    # what happens to lie on the box: it is configs[1] unless --corpus / $TAMP_CORPUS / --corpus-standin
    # ask for something else.  --probe-corpus (opt-in) looks for the file in the usual places -- no network unless
    # --fetch-corpus -- and a find is measured under also.configs2_corpus; what was tried is config.corpus_probe.
    corpus_probe = {"tried": [], "found": None, "fetch": "not attempted",
                    "note": "opt-in: --probe-corpus [--fetch-corpus] looks for enwik8; --corpus PATH / $TAMP_CORPUS run configs[2] on a file"}
    probe_found = None
    if not args.corpus and args.probe_corpus:
        probe_found, corpus_probe = wl.probe_corpus(env={}, fetch=(args.fetch_corpus and rank == 0))
        if dist is not None:
            dist.barrier()
    corpus_blob = None
    standin = bool(args.corpus_standin) and not args.corpus
    if args.corpus:
        corpus_blob = open(args.corpus, "rb").read()
        flat, in_off, in_len = wl.split_fixed(corpus_blob, args.stream_len, keep_tail=True)
        ranges = partition_streams(in_len, world)
    elif standin:
        # configs[2]'s shape on the frozen corpora: 24,414 x 4 KiB real-text streams, contiguous ranges balanced by bytes
        # over the GPUs (strong scaling); --shard-fraction F on one GPU times the first of F such ranges
        rows = wl.standin_rows(wl.CONFIGS2_STREAMS, args.stream_len)
        flat = rows.reshape(-1)
        in_off, in_len = wl.csr_for_fixed(len(rows), args.stream_len)
        ranges = partition_streams(in_len, world * max(1, args.shard_fraction))[:world] if world == 1 else partition_streams(in_len, world)
    shards = []
    for r in my_shards:
        if corpus_blob is not None or standin:
            b, e = ranges[r]
            lo = int(in_off[b]) if e > b else 0
            hi = int(in_off[e - 1] + in_len[e - 1]) if e > b else 0
            shards.append(Shard(torch, r, device_of(r), flat[lo:hi], in_off[b:e] - np.uint64(lo), in_len[b:e],
                                args.stream_len, conf_kw))
        else:
            rows = wl.synth_text(args.streams, args.stream_len, first_index=r * args.streams)
            off, ln = wl.csr_for_fixed(args.streams, args.stream_len)
            sh = Shard(torch, r, device_of(r), rows.reshape(-1), off, ln, args.stream_len, conf_kw)
            sh.rows = rows
            shards.append(sh)

    def fence():
        for sh in shards:
            sh.sync()
        if dist is not None:
            dist.barrier()
        for sh in shards:
            sh.sync()

    res = None
    for _ in range(args.warmup):
        res = [sh.launch() for sh in shards]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = [sh.launch(record=True) for sh in shards]
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    in_bytes = sum(sh.in_bytes for sh in shards)      # this process, one step
    out_bytes = sum(int(r.out_len.to(torch.int64).sum().item()) for r in res)
    status_ok = all(bool((r.status == 0).all().item()) for r in res)
    n_streams = sum(sh.n for sh in shards)
    totals = [in_bytes, out_bytes, n_streams, int(status_ok)]
    if dist is not None:
        tt = torch.tensor(totals, dtype=torch.int64)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        totals = tt.tolist()
        totals[3] = int(totals[3] == world)
    job_in, job_out, job_streams, job_ok = totals
    value = job_in * args.steps / elapsed / 1e6

    # dominant kernel: HIP events recorded around every launch of the timed region, on the launch stream
    sh0 = shards[0]
    ev = [a.elapsed_time(b) for sh in shards for (a, b) in sh.events]
    k_ms = float(np.mean([a.elapsed_time(b) for (a, b) in sh0.events])) if sh0.events else float("nan")
    alg_bytes = sh0.in_bytes + int(res[0].out_len.to(torch.int64).sum().item())  # in_len + out_len
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    if standin:
        workload = (f"configs[2] STAND-IN (enwik8 is not on this box): the distinct 4 KiB chunks of the three frozen corpora "
                    f"(prose, markup, Python sources) tiled and shuffled to {wl.CONFIGS2_STREAMS} streams, "
                    f"{'the first 1/%d of them' % args.shard_fraction if args.shard_fraction > 1 else 'all of them'}, window={args.window} "
                    f"literal=8 extended={args.extended}, contiguous stream ranges balanced by bytes over {world} GPU(s), inputs resident in HBM")
    elif corpus_blob is not None:
        workload = (f"configs[2]: {os.path.basename(args.corpus)} ({len(corpus_blob)} B) cut into {job_streams} independent "
                    f"streams of {args.stream_len} B (short tail kept as the last stream), window={args.window} literal=8 "
                    f"extended={args.extended}, contiguous stream ranges balanced by bytes over {world} GPU(s), inputs resident in HBM")
    else:
        workload = (f"configs[1]: {args.streams} x {args.stream_len} B synthetic-text streams per GPU, window={args.window} "
                    f"literal=8 extended={args.extended}, one batch launch per step and GPU, inputs resident in HBM")
    result = {
        "metric": "input MB/s, 4KiB-chunk batch compress window=10 (bit-exact vs C ref)",
        "value": round(value, 2),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if (corpus_blob is not None or standin) else "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": ("real: " + os.path.basename(args.corpus) + " sha256=" + hashlib.sha256(corpus_blob).hexdigest()[:16])
                if corpus_blob is not None else ("real text: frozen corpora of tests/golden (stand-in for enwik8)" if standin else "synthetic"),
        "config": {
            "workload": workload,
            "streams_total": job_streams,
            "stream_len": args.stream_len,
            "parallelism": (f"{world} rank(s), one GPU each (torch.distributed.run; gloo barrier + MAX only)" if launched else
                            f"one process, {world} device(s), one hipStream per device") + ", streams sharded, no collective"
                           + (" [TAMP_BENCH_ONE_DEVICE: all shards on cuda:0]" if one_device else ""),
            "compressed_ratio": round(job_out / max(job_in, 1), 4),
            "all_streams_ok": bool(job_ok),
            # host time inside the launch call per shard and step (enqueue only: the launches are asynchronous); with one
            # process driving N devices the sum over shards must stay well under one kernel time or devices starve
            "host_launch_us_per_shard": round(1e6 * sum(sh.host_s for sh in shards) / max(1, sum(sh.launches for sh in shards)), 1),
            "sum_kernel_ms_per_step": round(float(sum(np.mean([a.elapsed_time(b) for (a, b) in sh.events]) for sh in shards if sh.events)), 4),
            "corpus_probe": corpus_probe,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "tamp_compress_kernel",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None,
            "kernel_ms": round(k_ms, 4),
            "kernel_ms_minmax_all_shards": [round(min(ev), 4), round(max(ev), 4)] if ev else None,
            "timing": f"hipEvent pairs around each of the {len(sh0.events)} launches of the timed region on shard 0's stream",
            "algorithmic_bytes_per_launch": alg_bytes,
            "read_frac": round(sh0.in_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        },
    }
    traffic, src = pmc_traffic_bytes()
    if corpus_blob is None and not standin and args.streams == 65536 and args.stream_len == 4096:
        result["roofline"]["traffic"] = traffic
        result["roofline"]["traffic_source"] = src
        # the real limiter (SURVEY.md 8d's secondary counters): the kernel is bound by VALU issue, not by HBM
        result["roofline"].update(pmc_issue_figures(args.streams, sh0.in_bytes))
        if rank == 0 and world == 1 and not args.no_live_pmc and not args.no_cpu_baseline:
            # ... and measured now: three short rocprofv3 --pmc passes over this very command (own processes, separate
            # passes as the guide prescribes); the committed figures above stay as the fallback and as a cross-check
            live = live_pmc(args, sh0.in_bytes)
            if live.get("traffic"):
                result["roofline"]["traffic_committed"] = traffic
                result["roofline"].update(live)

    sh0.no_live_pmc = args.no_live_pmc
    extras = rank == 0 and not args.no_cpu_baseline
    if extras and world == 1:
        result["cpu_baseline"] = cpu_baseline(args, sh0, res[0], np, corpus_blob is not None)
    if extras and corpus_blob is not None:
        try:
            result["corpus_pins"] = corpus_pins(args, corpus_blob, torch, np)
        except Exception as e:
            result["corpus_pins"] = {"error": repr(e)[:200]}
    if extras and world == 1 and corpus_blob is None and not standin:
        # Not part of the metric (informational, each guarded): the v1 format, the decode of what was just produced,
        # and real text found on this machine next to the synthetic headline.
        also = {}
        try:
            also.update(also_v1_and_decode(sh0, res[0], torch))
        except Exception as e:
            also["error"] = repr(e)[:200]
        try:
            also["real_text"] = also_real_text(args, torch, np)
            # the metric's corpus class next to the synthetic headline (input MB/s, extended format = library default)
            result["config"]["real_text_MBps"] = {
                name: round(1000 * v["extended_GBps"]) for name, v in also["real_text"].items()
                if isinstance(v, dict) and "extended_GBps" in v}
        except Exception as e:
            also["real_text"] = {"error": repr(e)[:200]}
        try:
            also["one_long_stream"] = also_one_long_stream(torch, np)
        except Exception as e:  # noqa: BLE001
            also["one_long_stream"] = {"error": repr(e)[:200]}
        try:
            also["strong_scaling_standin"] = also_strong_scaling(args, torch, np)
        except Exception as e:  # noqa: BLE001
            also["strong_scaling_standin"] = {"error": repr(e)[:200]}
        if probe_found:
            try:
                also["configs2_corpus"] = also_corpus(args, probe_found, torch, np)
            except Exception as e:  # noqa: BLE001
                also["configs2_corpus"] = {"error": repr(e)[:200]}
        try:
            also["baseline_configs"] = also_baseline_configs(torch, np)
        except Exception as e:  # noqa: BLE001 -- the headline line must not die for a side measurement
            also["baseline_configs"] = {"error": repr(e)[:200]}
        result["also"] = also
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def also_configs1(args, torch, np, conf_kw, steps=10, warmup=3):
    """BASELINE configs[1] (65,536 x 4 KiB synthetic text) as a side measurement, for runs whose headline is configs[2]."""
    from tamp_amd import workloads as wl

    rows = wl.synth_text(65536, 4096)
    off, ln = wl.csr_for_fixed(65536, 4096)
    sh = Shard(torch, 0, torch.device("cuda", 0), rows.reshape(-1), off, ln, 4096, conf_kw)
    for _ in range(warmup):
        sh.launch()
    for _ in range(steps):
        res = sh.launch(record=True)
    sh.sync()
    k_ms = float(np.mean([a.elapsed_time(b) for (a, b) in sh.events]))
    return {"workload": "65536 x 4096 B synthetic text, window=10 literal=8 extended=%d" % int(conf_kw["extended"]),
            "kernel_ms": round(k_ms, 4), "input_GBps": round(sh.in_bytes / (k_ms * 1e-3) / 1e9, 2),
            "all_streams_ok": bool((res.status == 0).all().item())}


def also_one_long_stream(torch, np, n=100_000_000):
    """ONE stream of enwik8's length (the reference's own benchmark shape, /root/reference/README.md:309-312) in the v1
    format: its 1,024-position blocks spread over all workgroups (tamp_compress_kernel<.., BLOCKM>).  The
    frozen prose corpus tiled to 100,000,000 bytes, resident in HBM; kernel time of the whole call (zero fill, tables, scan,
    emit); a 4 MiB stream of the same text checked against the reference C (the 100 MB one: tests/test_gpu_round5.py)."""
    import tamp_amd
    from tamp_amd import workloads as wl

    dev = torch.device("cuda", 0)
    blob = wl.real_text("prose")
    flat = np.frombuffer((blob * (n // len(blob) + 1))[:n], dtype=np.uint8).copy()
    d = torch.from_numpy(flat).to(dev)
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    ln = torch.tensor([n], dtype=torch.int32, device=dev)
    ms, r = [], None
    for _ in range(4):
        r = tamp_amd.compress_batch(d, off, ln, window=10, literal=8, extended=False, max_in_len=n, timing=True)
        ms.append(float(r.kernel_ms))
    kind, impl = _checker()
    k = 4 << 20
    want = impl.compress_batch(flat[:k], np.zeros(1, np.uint64), np.array([k], np.uint32), window=10, literal=8, extended=False).stream(0)
    got = tamp_amd.compress_batch(d[:k], off, torch.tensor([k], dtype=torch.int32, device=dev), window=10, literal=8,
                                  extended=False, max_in_len=k).stream(0)
    # ... and back: ONE stream decoded by the whole device (tamp_decompress_long_kernel.hpp), stream and output in HBM
    clen = int(r.out_len[0])
    dms, back = [], None
    for _ in range(3):
        back = tamp_amd.decompress_batch(r.out[:clen], off, torch.tensor([clen], dtype=torch.int32, device=dev), out_cap=n + 64, timing=True)
        dms.append(float(back.kernel_ms))
    same = int(back.status[0]) == 2 and int(back.out_len[0]) == n and bool(torch.equal(back.out[:n], d))
    # the library's DEFAULT format (extended = 1): compressing one such stream stays with one workgroup (its lags make the window
    # depend on the parse), decoding it does not (round 6: RLE / extended-match tokens, window_pos from a pass over the tokens
    # that can lag).  32 MB of the same text, compressed once.
    ne = 32_000_000
    lne = torch.tensor([ne], dtype=torch.int32, device=dev)
    re_ = tamp_amd.compress_batch(d[:ne], off, lne, window=10, literal=8, extended=True, max_in_len=ne, timing=True)
    celen = int(re_.out_len[0])
    xms, xback = [], None
    for _ in range(3):
        xback = tamp_amd.decompress_batch(re_.out[:celen], off, torch.tensor([celen], dtype=torch.int32, device=dev), out_cap=ne + 64, timing=True)
        xms.append(float(xback.kernel_ms))
    xsame = int(xback.status[0]) == 2 and int(xback.out_len[0]) == ne and bool(torch.equal(xback.out[:ne], d[:ne]))
    extended = {"bytes": ne, "format": "extended=1 (the library default), window=10 literal=8", "compress_kernel_ms": round(float(re_.kernel_ms), 1),
                "compress_note": "one workgroup: the extended format's lags make the window depend on the parse",
                "ratio": round(celen / ne, 4), "decode_kernel_ms": round(min(xms[1:]), 2),
                "decode_output_GBps": round(ne / (min(xms[1:]) * 1e-3) / 1e9, 2), "decode_round_trip": "equal" if xsame else "MISMATCH"}
    return {"bytes": n, "extended_format": extended, "format": "v1 (extended=0), window=10 literal=8", "kernel_ms": round(min(ms[1:]), 3),
            "input_GBps": round(n / (min(ms[1:]) * 1e-3) / 1e9, 2), "status": int(r.status[0]),
            "ratio": round(int(r.out_len[0]) / n, 4), "parity_4MiB_stream": ("bit-exact" if got == want else "MISMATCH") + f" vs {kind}",
            "decode_kernel_ms": round(min(dms[1:]), 2), "decode_output_GBps": round(n / (min(dms[1:]) * 1e-3) / 1e9, 2),
            "decode_round_trip": "equal" if same else "MISMATCH"}


def also_strong_scaling(args, torch, np, reps=5):
    """Strong scaling of BASELINE configs[2] PREDICTED on one device: the 24,414-stream real-text
    stand-in is cut into N contiguous ranges balanced by bytes -- what `bench.py --corpus-standin --gpus N` gives each GPU
    -- and every range is timed alone on this device; predicted_speedup_at_N = t(all 24,414) / max over the N ranges.
    It leaves out what only a second device can show (host launch overlap: 31 us per shard)."""
    import tamp_amd
    from tamp_amd import partition_streams
    from tamp_amd import workloads as wl

    dev = torch.device("cuda", 0)
    L = 4096
    rows = wl.standin_rows(wl.CONFIGS2_STREAMS, L)
    off, ln = wl.csr_for_fixed(len(rows), L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
    len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)

    def timed(b, e):
        ms = []
        o = off_t[b:e] - off_t[b]
        d = data[int(off[b]): int(off[e - 1] + ln[e - 1])]
        for _ in range(reps + 1):
            r = tamp_amd.compress_batch(d, o, len_t[b:e], max_in_len=L, timing=True, window=args.window, literal=8,
                                        extended=bool(args.extended))
            ms.append(float(r.kernel_ms))
        return float(np.median(ms[1:]))

    out = {"streams": len(rows), "stream_len": L, "scaling": "strong (fixed total work: 24,414 streams over N GPUs)",
           "timing": f"kernel time (hipEvents), median of {reps} launches per range, one device"}
    t_all = timed(0, len(rows))
    out["t_ms_all"] = round(t_all, 4)
    out["input_GBps_all"] = round(len(rows) * L / (t_all * 1e-3) / 1e9, 2)
    for N in (2, 4, 8):
        ts = [timed(b, e) for (b, e) in partition_streams(ln, N)]
        out[f"t_ms_shards_at_{N}"] = [round(t, 4) for t in ts]
        out[f"predicted_speedup_at_{N}"] = round(t_all / max(ts), 3)
    return out


def also_corpus(args, path, torch, np):
    """A corpus the opt-in probe found (enwik8): configs[2] on one device as a SIDE measurement -- cut into 4 KiB streams
    (short tail kept), kernel-timed, the first 2,048 streams checked against the reference C, whole-file pins when the
    file has enwik8's length."""
    import tamp_amd
    from tamp_amd import workloads as wl

    blob = open(path, "rb").read()
    flat, off, ln = wl.split_fixed(blob, 4096, keep_tail=True)
    dev = torch.device("cuda", 0)
    data = torch.from_numpy(np.ascontiguousarray(flat)).to(dev)
    off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
    len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
    out = {"file": os.path.basename(path), "bytes": len(blob), "streams": int(len(ln)), "sha256": hashlib.sha256(blob).hexdigest()[:16]}
    kind, impl = _checker()
    for ext in (True, False):
        ms, r = [], None
        for _ in range(4):
            r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=4096, timing=True, window=args.window, literal=8, extended=ext)
            ms.append(float(r.kernel_ms))
        k = min(2048, len(ln))
        want = impl.compress_batch(flat[: int(off[k - 1] + ln[k - 1])], off[:k], ln[:k], threads=host_threads()[0],
                                   window=args.window, literal=8, extended=ext)
        olen, ooff = r.out_len[:k].cpu().numpy(), r.out_off[:k].cpu().numpy()
        gout = r.out[: int(ooff[-1] + olen[-1])].cpu().numpy()
        ok = all(gout[ooff[i]: ooff[i] + olen[i]].tobytes() == want.stream(i) for i in range(k))
        tag = "extended" if ext else "v1"
        out[tag] = {"kernel_ms": round(min(ms[1:]), 4), "input_GBps": round(len(blob) / (min(ms[1:]) * 1e-3) / 1e9, 2),
                    "ratio": round(float(r.out_len.to(torch.int64).sum().item()) / len(blob), 4),
                    "parity": ("bit-exact" if ok else "MISMATCH") + f" (first {k} streams vs {kind})"}
    try:
        out["corpus_pins"] = corpus_pins(args, blob, torch, np)
    except Exception as e:  # noqa: BLE001
        out["corpus_pins"] = {"error": repr(e)[:200]}
    return out


def pmc_traffic_bytes():
    """HBM bytes per launch of the compress kernel from the committed rocprofv3 PMC summaries of this command (bench.py
    cannot run the profiler on itself); (None, reason) if they are missing."""
    import csv

    def mean_kb(name):
        path = os.path.join(ROOT, "profiles", name)
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "tamp_compress" in r["Kernel_Name"]]
        return sum(vals) / len(vals)

    for tag in (PROFILE_TAG, "r2", "r1j"):
        try:
            fetch_kb = mean_kb(f"{tag}_pmc_fetch_counter_collection.csv")
            write_kb = mean_kb(f"{tag}_pmc_write_counter_collection.csv")
        except Exception:
            continue
        return int(2 * fetch_kb * 1024 + write_kb * 1024), (
            f"profiles/{tag}_pmc_{{fetch,write}}_counter_collection.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
            "passes, tools/pmc_run.sh) of this command; bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction) + "
            "WRITE_SIZE KB" + ("" if tag == PROFILE_TAG else " [STALE: captured on an earlier build of the kernel]"))
    return None, "no committed PMC pass"


#: Cycles of its SIMD's issue time one wave64 VALU instruction costs, MEASURED (tools/probe/valu_issue_probe.hip ->
#: profiles/r5_valu_issue_probe.txt, seven wavefronts per SIMD): 4.08-4.13 for every integer class this kernel leans on
#: (v_and/v_lshl with a scalar operand, v_alignbyte, v_bfe, v_cmp, v_cndmask, v_min/v_max, v_mul, every three-operand
#: form, DPP moves); only plain VOP1/VOP2 and/or/xor/add/sub/lshr/mov on VGPR or constant operands pair up across waves
#: (2.2), and interleaved with the others they measured 3.7.  4 is the round figure the ceiling is quoted with.
VALU_CYCLES_PER_WAVE_INSTRUCTION = 4.0
ISSUE_MODEL = ("measured: 4.1 cycles of SIMD issue time per wave64 integer VALU instruction (2.2 only for plain VGPR/constant "
               "VOP1/VOP2 and/or/xor/add/sub/lshr/mov pairing up across waves; 3.7 interleaved), SALU overlaps with VALU of other "
               "waves -- tools/probe/valu_issue_probe.hip, profiles/r5_valu_issue_probe.txt")


def live_pmc(args, in_bytes):
    """HBM bytes and VALU figures of the compress kernel from rocprofv3 --pmc passes run NOW over `bench.py --steps 2
    --warmup 1 --no-cpu-baseline --no-live-pmc` (FETCH_SIZE, WRITE_SIZE and the SQ group each in a pass of their own, with
    --kernel-trace only); {} if rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc",
           "--streams", str(args.streams), "--stream-len", str(args.stream_len), "--window", str(args.window),
           "--extended", str(args.extended)]
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    t0 = time.perf_counter()
    for tag, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                          ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])):
        d = tempfile.mkdtemp(prefix="tamp_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd],
                           cwd="/tmp", env=env, capture_output=True, timeout=240, check=True)
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "tamp_compress" in r["Kernel_Name"]:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                got[k] = sum(v) / len(v)
        except Exception:
            return {}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")
    if any(k not in got for k in need):
        return {}
    cycles = got["GRBM_GUI_ACTIVE"] / 8.0
    return {
        "traffic": int(2 * got["FETCH_SIZE"] * 1024 + got["WRITE_SIZE"] * 1024),
        "traffic_source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ group, one pass each, over "
                          "`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc`; bytes per launch = 2 x FETCH_SIZE KB "
                          "(gfx950 correction) + WRITE_SIZE KB",
        "valu_per_stream": round(got["SQ_INSTS_VALU"] / args.streams),
        "salu_per_stream": round(got.get("SQ_INSTS_SALU", 0) / args.streams),
        # (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles; with eight wavefronts per SIMD the 2.2-cycle class overlaps and the quotient
        # passes 1.0 by a few per cent: reported raw as well, clamped here -- 1.0 = a VALU instruction in flight every cycle)
        "valu_busy": min(1.0, round(got["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles), 3)),
        "valu_busy_raw": round(got["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles), 3),
        "issue_ceiling_GBps": round(in_bytes / (got["SQ_INSTS_VALU"] * VALU_CYCLES_PER_WAVE_INSTRUCTION / (1024 * 2.4e9)) / 1e9, 1),
        "issue_source": "measured in this run (SQ pass above)",
        "issue_model": ISSUE_MODEL,
        "live_pmc_seconds": round(time.perf_counter() - t0, 1),
    }


def pmc_issue_figures(n_streams, in_bytes):
    """VALU wave-instructions per stream, VALU busy and the issue ceiling from the committed SQ counter passes of this
    command (profiles/<tag>_pmc_sq*.csv, tools/pmc_run.sh).  issue_ceiling_GBps = input bytes / (VALU instructions x 4
    cycles / (1024 SIMDs x clock)): what the chip would reach at this instruction count with every issue slot used."""
    import csv

    def means(name):
        acc = {}
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
            if "tamp_compress" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in acc.items()}

    for tag in (PROFILE_TAG, "r2"):
        try:
            m = means(f"{tag}_pmc_sq_counter_collection.csv")
            m.update(means(f"{tag}_pmc_sq2_counter_collection.csv"))
            cycles = m["GRBM_GUI_ACTIVE"] / 8.0  # per XCD
            valu = m["SQ_INSTS_VALU"]
        except Exception:
            continue
        clock_ghz = 2.4
        return {
            "valu_per_stream": round(valu / n_streams),
            "salu_per_stream": round(m.get("SQ_INSTS_SALU", 0) / n_streams),
            "valu_busy": min(1.0, round(m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles), 3)),
            "valu_busy_raw": round(m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles), 3),
            "issue_ceiling_GBps": round(in_bytes / (valu * VALU_CYCLES_PER_WAVE_INSTRUCTION / (1024 * clock_ghz * 1e9)) / 1e9, 1),
            "issue_model": ISSUE_MODEL,
            "issue_source": f"profiles/{tag}_pmc_sq{{,2}}_counter_collection.csv"
                            + ("" if tag == PROFILE_TAG else " [STALE: captured on an earlier build of the kernel]"),
        }
    return {"valu_per_stream": None, "issue_source": "no committed SQ counter pass"}


def _checker():
    from oracle.checker import Oracle, Ref

    return ("reference", Ref()) if Ref.available() else ("port", Oracle())


def cpu_baseline(args, shard, gpu_res, np, from_corpus):
    """The reference C (oracle/_ref, kind "reference") or this repo's restatement (kind "port") timed on the host cores
    this process may use, on a bounded sample of the same workload; its output doubles as the parity check of the GPU
    run.  Method of BASELINE.md section 3: one pthread per usable core, 3 warm runs, best of 5."""
    kind, impl = _checker()
    threads, affinity, quota = host_threads()
    sample = min(args.cpu_sample, shard.n)
    off = shard.off[:sample].cpu().numpy().astype(np.uint64)
    ln = shard.len[:sample].cpu().numpy().astype(np.uint32)
    hi = int(off[-1] + ln[-1])
    sub = shard.data[:hi].cpu().numpy()
    kw = dict(window=args.window, literal=8, extended=bool(args.extended))
    warm = sub[: min(hi, 8 << 20)]
    wn = int(np.searchsorted(off + ln, warm.size, side="right"))
    for _ in range(3):
        if wn:
            impl.compress_batch(warm, off[:wn], ln[:wn], threads=threads, **kw)
    best = None
    for _ in range(5):
        r = impl.compress_batch(sub, off, ln, threads=threads, **kw)
        if best is None or r.seconds < best.seconds:
            best = r
    olen = gpu_res.out_len[:sample].cpu().numpy()
    ooff = gpu_res.out_off[:sample].cpu().numpy()
    gout = gpu_res.out[: int(ooff[-1] + olen[-1])].cpu().numpy()
    mism = -1
    for i in range(sample):
        if gout[ooff[i] : ooff[i] + olen[i]].tobytes() != best.stream(i):
            mism = i
            break
    return {
        "value": round(hi / best.seconds / 1e6, 2),
        "unit": "MB/s",
        "cores": threads,
        "threads": threads,
        "kind": kind,
        "cpu_model": cpu_model(),
        "sample": f"first {sample} of the {shard.n} streams ({hi} B), one pthread per usable CPU = {threads} "
                  f"(affinity mask {affinity}, cgroup quota {quota}, os.cpu_count()={os.cpu_count()}), 3 warm runs, best of 5",
        "per_core": round(hi / best.seconds / 1e6 / threads, 2),
        "parity": (f"bit-exact, {sample} streams" if mism < 0 else f"MISMATCH at stream {mism}"),
    }


def also_v1_and_decode(shard, res, torch):
    import tamp_amd

    out = {}
    ms = []
    for _ in range(3):
        shard.events.clear()
        shard.launch(record=True, extended=False)
        shard.sync()
        ms.append(shard.events[-1][0].elapsed_time(shard.events[-1][1]))
    out["compress_v1_format_MBps"] = round(shard.in_bytes / (min(ms) * 1e-3) / 1e6, 1)
    torch.cuda.synchronize(shard.device)
    back, best = None, 1e30
    with torch.cuda.device(shard.device):
        for _ in range(6):  # (the first call sizes the decoder's scratch slab)
            back = tamp_amd.decompress_batch(res.out, res.out_off, res.out_len, out_cap=shard.max_len, timing=True)
            best = min(best, float(back.kernel_ms))
    n = shard.n
    ok = bool((back.out_len == shard.len).all().item())
    if ok and shard.max_len * n == shard.in_bytes:
        ok = bool(torch.equal(back.out[: shard.in_bytes], shard.data[: shard.in_bytes]))
    comp_bytes = int(res.out_len.to(torch.int64).sum().item())
    gbs = (comp_bytes + shard.in_bytes) / (best * 1e-3) / 1e9
    out["decompress_output_MBps"] = round(shard.in_bytes / (best * 1e-3) / 1e6, 1)
    out["decompress_round_trip"] = "bit-exact" if ok else "MISMATCH"
    traffic, tsrc = pmc_decode_traffic_bytes()
    if n == 65536 and shard.max_len == 4096 and not getattr(shard, "no_live_pmc", False):
        live, lsrc = live_decode_traffic()
        if live:
            traffic, tsrc = live, lsrc
    out["decompress"] = {"kernel_ms": round(best, 4), "roofline": {
        "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
        "algorithmic_bytes_per_launch": comp_bytes + shard.in_bytes,
        "traffic": traffic if (n == 65536 and shard.max_len == 4096) else None, "traffic_source": tsrc,
        "note": "compressed bytes read + bytes written, hipEvents around the decode launch (header pre-pass included)"}}
    return out


def live_decode_traffic():
    """HBM bytes of one decode of the bench batch measured NOW: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (a pass each) over
    tools/dec_traffic.py, summed over the decode's kernels; (None, why) if the profiler is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    script = os.path.join(ROOT, "tools", "dec_traffic.py")
    if not os.path.exists(exe) or not os.path.exists(script):
        return None, "no rocprofv3"
    total_kb = {}
    for tag in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tamp_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", tag, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, script],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", GRAFT_REPO_ROOT=ROOT), capture_output=True, timeout=240, check=True)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if "tamp_" in k and "compress_kernel" not in k:
                        per.setdefault(k, []).append(float(r["Counter_Value"]))
            total_kb[tag] = sum(sum(v) / len(v) for v in per.values())
        except Exception as e:  # noqa: BLE001
            return None, repr(e)[:80]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if not total_kb.get("FETCH_SIZE") or not total_kb.get("WRITE_SIZE"):
        return None, "no counter rows"
    return int(2 * total_kb["FETCH_SIZE"] * 1024 + total_kb["WRITE_SIZE"] * 1024), (
        "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (a pass each) over tools/dec_traffic.py, "
        "2 x FETCH_SIZE KB + WRITE_SIZE KB summed over the decode's kernels")


def pmc_decode_traffic_bytes():
    """HBM bytes of one decode of the bench batch (header pre-pass + parse + resolve + leftovers), from the committed
    FETCH_SIZE / WRITE_SIZE passes over tools/dec_traffic.py (the bench's decode leg alone)."""
    import csv

    def total_kb(name):
        per = {}
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
            k = r["Kernel_Name"]
            if "tamp_" in k and "compress_kernel" not in k:
                per.setdefault(k, []).append(float(r["Counter_Value"]))
        return sum(sum(v) / len(v) for v in per.values())

    try:
        f = total_kb(f"{PROFILE_TAG}_pmc_dec2_fetch_counter_collection.csv")
        w = total_kb(f"{PROFILE_TAG}_pmc_dec2_write_counter_collection.csv")
    except Exception:
        return None, "no committed PMC pass for the decode"
    return int(2 * f * 1024 + w * 1024), (f"profiles/{PROFILE_TAG}_pmc_dec2_{{fetch,write}}_counter_collection.csv: 2 x FETCH_SIZE KB + "
                                          "WRITE_SIZE KB summed over the decode's kernels (tools/dec_traffic.py, tools/pmc_run.sh)")


def also_real_text(args, torch, np):
    """Real text (the frozen fixtures of tests/golden/make_corpus.py; the metric's own corpus, enwik8, is not in the
    image): GB/s of input per corpus, both formats, as many 4 KiB streams as the headline batch has (65,536; the 768
    chunks of a corpus repeated to fill them), with the first 512 streams of each checked against the reference C."""
    import tamp_amd
    from tamp_amd import workloads as wl

    kind, impl = _checker()
    dev = torch.device("cuda", 0)
    n, L = args.streams, 4096  # the headline's batch size (round 2 used 16,384: a quarter as many device fills, ~7 % slower per stream)
    out = {"streams": n, "stream_len": L, "checker": kind}
    sources = {name: wl.real_text(name) for name in wl.REAL_TEXT_SOURCES}
    sources["synthetic (configs[1] text)"] = None
    for name, blob in sources.items():
        if blob is not None and len(blob) < 64 * L:
            out[name] = "not found on this machine"
            continue
        rows = wl.synth_text(n, L) if blob is None else wl.tile_rows(blob, n, L)
        off, ln = wl.csr_for_fixed(n, L)
        data = torch.from_numpy(rows.reshape(-1)).to(dev)
        off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
        len_t = torch.from_numpy(ln.astype(np.int32)).to(dev)
        entry = {"corpus_bytes": None if blob is None else len(blob)}
        for ext in (True, False):
            ms, r = [], None
            for _ in range(3):
                r = tamp_amd.compress_batch(data, off_t, len_t, max_in_len=L, timing=True, window=args.window, literal=8,
                                            extended=ext)
                ms.append(float(r.kernel_ms))
            # every DISTINCT chunk of the corpus (768: the batch repeats them), 2,048 rows of the synthetic text
            k = 2048 if blob is None else min(n, len(blob) // L)
            want = impl.compress_batch(rows[:k].reshape(-1), off[:k], ln[:k], threads=host_threads()[0],
                                       window=args.window, literal=8, extended=ext)
            olen = r.out_len[:k].cpu().numpy()
            ooff = r.out_off[:k].cpu().numpy()
            gout = r.out[: int(ooff[-1] + olen[-1])].cpu().numpy()
            ok = all(gout[ooff[i] : ooff[i] + olen[i]].tobytes() == want.stream(i) for i in range(k))
            tag = "extended" if ext else "v1"
            entry[tag + "_GBps"] = round(n * L / (min(ms) * 1e-3) / 1e9, 2)
            entry[tag + "_ratio"] = round(float(r.out_len.to(torch.int64).sum().item()) / (n * L), 4)
            entry[tag + "_parity"] = ("bit-exact" if ok else "MISMATCH") + f" (first {k} streams = every distinct chunk)" * (blob is not None)
        out[name] = entry
    return out


def also_baseline_configs(torch, np):
    """BASELINE configs[3] and configs[4] at a size that fits the bench's time budget (the full sizes run in
    tests/test_gpu_round2.py and tools/config4.py / config5.py): decode of 262,144 pre-compressed 4 KiB streams with
    windows 2^8..2^12 interleaved, and compress of 1,048,576 256-byte telemetry messages with the shared custom
    dictionary (w=8 l=7 extended); both are checked by decoding back to the input."""
    import tamp_amd
    from tamp_amd import workloads as wl

    dev = torch.device("cuda", 0)
    out = {}
    # ---- configs[3]: the streams are produced by this library's compressor, per window, and laid out as one slab
    n, L = 262144, 4096
    base = wl.synth_text(n // 5 + 1, L, first_index=1 << 20)
    wsel = torch.arange(n, device=dev) % 5 + 8
    olen = torch.zeros(n, dtype=torch.int64, device=dev)
    parts = {}
    for w in range(8, 13):
        ids = torch.nonzero(wsel == w).flatten()
        rows = base[: len(ids)]
        off, ln = wl.csr_for_fixed(len(ids), L)
        r = tamp_amd.compress_batch(torch.from_numpy(rows.reshape(-1)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev),
                                    torch.from_numpy(ln.astype(np.int32)).to(dev), window=w, max_in_len=L)
        parts[w] = (ids, r)
        olen[ids] = r.out_len.to(torch.int64)
    in_off = torch.cumsum(olen, 0) - olen
    slab = torch.empty(int(olen.sum().item()) + 64, dtype=torch.uint8, device=dev)
    for w, (ids, r) in parts.items():
        lens = r.out_len.to(torch.int64)
        rep = torch.repeat_interleave(torch.arange(len(ids), device=dev), lens)
        within = torch.arange(int(lens.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
        slab[in_off[ids][rep] + within] = r.out[r.out_off.to(torch.int64)[rep] + within]
        del rep, within
    ms, d = [], None
    for _ in range(3):
        d = tamp_amd.decompress_batch(slab, in_off, olen.to(torch.int32), out_cap=L + 8, timing=True)
        ms.append(float(d.kernel_ms))
    ok = bool((d.status == 2).all().item()) and bool((d.out_len == L).all().item())
    got = d.out.view(-1)[: n * (L + 8)].view(n, L + 8)
    for w, (ids, r) in parts.items():
        k = min(256, len(ids))
        ok = ok and bool(torch.equal(got[ids[:k], :L].cpu(), torch.from_numpy(base[:k])))
    comp = int(olen.sum().item())
    out["configs[3] decode"] = {"streams": n, "stream_len": L, "windows": "2^8..2^12 interleaved", "kernel_ms": round(min(ms), 3),
                                "output_GBps": round(n * L / (min(ms) * 1e-3) / 1e9, 1),
                                "algorithmic_GBps": round((comp + n * L) / (min(ms) * 1e-3) / 1e9, 1),
                                "round_trip_sample": "equal" if ok else "MISMATCH"}
    del slab, d, got, parts
    # ---- configs[4]: one GPU's share is 2,097,152 messages; half of it here
    n, L = 1 << 20, 256
    dic = wl.telemetry_dictionary(bytes(tamp_amd.initialize_dictionary(256, literal=7)))
    rows = wl.telemetry(n, L)
    data = torch.from_numpy(rows.reshape(-1)).to(dev)
    off_t = torch.arange(n, dtype=torch.int64, device=dev) * L
    len_t = torch.full((n,), L, dtype=torch.int32, device=dev)
    ms, r = [], None
    for _ in range(3):
        r = tamp_amd.compress_batch(data, off_t, len_t, window=8, literal=7, dictionary=dic, max_in_len=L, timing=True)
        ms.append(float(r.kernel_ms))
    back = tamp_amd.decompress_batch(r.out, r.out_off, r.out_len, out_cap=L + 8, dictionary=dic, timing=True)
    got = back.out.view(-1)[: n * (L + 8)].view(n, L + 8)[:, :L]
    ok = bool((r.status == 0).all().item()) and bool((back.out_len == L).all().item()) and bool(torch.equal(got, data.view(n, L)))
    out["configs[4] compress"] = {"messages": n, "message_len": L, "conf": "window=8 literal=7 extended=1, shared custom dictionary",
                                  "kernel_ms": round(min(ms), 3), "input_GBps": round(n * L / (min(ms) * 1e-3) / 1e9, 1),
                                  "messages_per_s": round(n / (min(ms) * 1e-3)),
                                  "ratio": round(float(r.out_len.to(torch.int64).sum().item()) / (n * L), 4),
                                  "decode_output_GBps": round(n * L / (float(back.kernel_ms) * 1e-3) / 1e9, 1),
                                  "round_trip": "equal" if ok else "MISMATCH"}
    return out


def corpus_pins(args, blob, torch, np, reference=None):
    """Whole-file pins of the reference for enwik8: the file as ONE stream, both formats, SHA-256 and size of the output
    (tests/test_dataset_regression.py:38-43, README.md:266), plus the first 100 KB in the v1 format (README.md:336).
    A file of another size has no published pin; with ``reference`` (a checker from oracle/) the same ONE-stream outputs
    are compared with what the reference C produces for this very file (the 100,000,000-byte test of tests/)."""
    import tamp_amd
    from tamp_amd import workloads as wl

    pins = wl.ENWIK8_PINS
    is_enwik8_size = len(blob) == pins["len"]
    if not is_enwik8_size and reference is None:
        return {"checked": False, "why": f"file is {len(blob)} B, enwik8 is {pins['len']} B: no whole-file pin applies"}
    out = {"checked": True, "bytes": len(blob)}
    for ext in (False, True):
        r = tamp_amd.compress_batch([blob], window=10, literal=8, extended=ext)
        got = r.stream(0)
        tag = "extended" if ext else "v1"
        digest = hashlib.sha256(got).hexdigest()
        out[tag] = {"size": len(got), "status": int(r.status[0]), "sha256": digest[:16]}
        if is_enwik8_size:
            out[tag].update(size_pin=pins[tag + "_size"], sha256_matches_reference=digest == pins[tag + "_sha256"])
        if reference is not None:
            flat = np.frombuffer(blob, dtype=np.uint8)
            want = reference.compress_batch(flat, np.zeros(1, np.uint64), np.array([len(blob)], np.uint32), window=10,
                                            literal=8, extended=ext).stream(0)
            out[tag]["matches_checker"] = got == want
            out[tag]["checker_size"] = len(want)
    sizes = {("extended" if ext else "v1"): len(tamp_amd.compress_batch([blob[:100_000]], window=10, literal=8,
                                                                          extended=ext).stream(0)) for ext in (False, True)}
    out["first_100k"] = dict(sizes, size_pin=pins["first_100k_v1_size"],
                             note="README.md:336 quotes one size for the first 100,000 bytes without naming the format")
    return out


if __name__ == "__main__":
    main()
