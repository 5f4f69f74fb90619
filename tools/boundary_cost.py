"""What a single-object call costs at the C boundary: every object-level call is at least one
device launch plus its copies, so the reference-named API is a drop-in for CORRECTNESS and for batch callers, not a faster
single-stream codec.  Measures, through the reference's own names (include/tamp_compat.h):
  * tamp_compressor_sink + tamp_compressor_poll: latency per poll (one token);
  * tamp_compressor_compress_and_flush on a fresh object for 32 B .. 1 MiB: latency and MB/s;
  * tamp_compress_stream over a memory reader with 32 B / 4 KiB / 1 MiB buffers;
  * tamp_batch_compress from host memory for 1 .. 4,096 streams of 4 KiB: where the device overtakes one host core of the
    reference (bench.py cpu_baseline.per_core, ~15 MB/s on the GPU box's EPYC 9575F).
Dev tool (GPU box): python tools/boundary_cost.py"""
import ctypes as C, os, subprocess, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from tamp_amd import _lib, workloads as wl
import tamp_amd

if len(sys.argv) > 2 and sys.argv[1] == "stream":  # child: one tamp_compress_stream measurement with the env's buffer size
    pass
lib = _lib.load()
class TampConf(C.Structure):
    _fields_ = [("window", C.c_uint16, 4), ("literal", C.c_uint16, 4), ("use_custom_dictionary", C.c_uint16, 1),
                ("extended", C.c_uint16, 1), ("dictionary_reset", C.c_uint16, 1), ("append", C.c_uint16, 1), ("lazy_matching", C.c_uint16, 1)]
class MemReader(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
class MemWriter(C.Structure):
    _fields_ = [("data", C.c_void_p), ("capacity", C.c_size_t), ("pos", C.c_size_t)]
CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
sz = C.POINTER(C.c_size_t)
lib.tamp_compressor_init.restype = C.c_int8; lib.tamp_compressor_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.tamp_compressor_sink.restype = None; lib.tamp_compressor_sink.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz]
lib.tamp_compressor_poll.restype = C.c_int8; lib.tamp_compressor_poll.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz]
lib.tamp_compressor_compress_and_flush.restype = C.c_int8
lib.tamp_compressor_compress_and_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz, C.c_void_p, C.c_size_t, sz, C.c_bool]
lib.tamp_compress_stream.restype = C.c_int8; lib.tamp_compress_stream.argtypes = [C.c_void_p] * 5 + [sz, sz, CB, C.c_void_p]
mem_read = C.cast(lib.tamp_stream_mem_read, C.c_void_p); mem_write = C.cast(lib.tamp_stream_mem_write, C.c_void_p)
text = (wl.real_text("prose") * 2)[: 4 << 20]
conf = TampConf(window=10, literal=8, extended=1)

def fresh():
    window, comp = (C.c_ubyte * 1024)(), (C.c_ubyte * 48)()
    assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
    return window, comp

if len(sys.argv) > 2 and sys.argv[1] == "stream":
    n = int(sys.argv[2])
    window, comp = fresh()
    src = (C.c_ubyte * n).from_buffer_copy(text[:n]); dst = (C.c_ubyte * (n + 4096))()
    rd, wr = MemReader(C.addressof(src), n, 0), MemWriter(C.addressof(dst), n + 4096, 0)
    cin, cout = C.c_size_t(0), C.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.tamp_compress_stream(comp, mem_read, C.byref(rd), mem_write, C.byref(wr), C.byref(cin), C.byref(cout), C.cast(None, CB), None)
    dt = time.perf_counter() - t0
    ok = bytes(dst[: wr.pos]) == tamp_amd.compress(text[:n])
    print(f"| `tamp_compress_stream`, {os.environ.get('TAMP_AMD_STREAM_BUFFER_BYTES')}-byte buffer, {n} B of prose | {dt * 1e3:.1f} ms | {n / dt / 1e6:.3f} MB/s | same bytes as one-shot: {ok} |")
    sys.exit(0 if rc == 0 and ok else 1)

tamp_amd.compress(text[:4096])  # (first launch: context, module load)
print("| call | latency | rate | note |\n|---|---|---|---|")
# --- one token per call: sink 16 bytes, poll once ---
window, comp = fresh()
out = (C.c_ubyte * 64)(); w = C.c_size_t(0); k = C.c_size_t(0)
at, lat = 0, []
for i in range(300):
    buf = (C.c_ubyte * 16).from_buffer_copy(text[at: at + 16])
    lib.tamp_compressor_sink(comp, buf, 16, C.byref(k)); at += k.value
    t0 = time.perf_counter()
    lib.tamp_compressor_poll(comp, out, 64, C.byref(w))
    lat.append(time.perf_counter() - t0)
lat = np.array(lat[20:])
print(f"| `tamp_compressor_sink` + `tamp_compressor_poll` (one token) | median {np.median(lat) * 1e6:.0f} us, p90 {np.percentile(lat, 90) * 1e6:.0f} us | {2.65 / np.median(lat) / 1e6:.4f} MB/s at 2.65 B per token | every poll is a launch of the resume kernel + 3 copies |")
# --- one-shot calls on a fresh object ---
for n in (32, 256, 4096, 65536, 1 << 20, 4 << 20):
    src = (C.c_ubyte * n).from_buffer_copy(text[:n]); dst = (C.c_ubyte * (n + n // 8 + 64))()
    ts = []
    for rep in range(5 if n >= (1 << 20) else 20):
        window, comp = fresh()
        t0 = time.perf_counter()
        rc = lib.tamp_compressor_compress_and_flush(comp, dst, len(dst), C.byref(w), src, n, C.byref(k), False)
        ts.append(time.perf_counter() - t0)
        assert rc == 0 and k.value == n
    t = float(np.median(ts[1:]))
    print(f"| `tamp_compressor_compress_and_flush`, {n} B | {t * 1e6:.0f} us | {n / t / 1e6:.2f} MB/s | {'resume kernel (token level)' if n < 2048 else 'batch kernel, one workgroup (segment mode)'} |")
# --- tamp_compress_stream at three buffer sizes (the buffer is read from the environment at call time: child processes) ---
for bufsz, n in ((32, 16384), (4096, 1 << 18), (1 << 20, 4 << 20)):
    r = subprocess.run([sys.executable, __file__, "stream", str(n)], env=dict(os.environ, TAMP_AMD_STREAM_BUFFER_BYTES=str(bufsz)), capture_output=True, text=True)
    print(r.stdout.strip() or ("stream child failed: " + r.stderr[-300:]))
# --- where the batch call overtakes one host core ---
for ns in (1, 4, 16, 64, 256, 1024, 4096):
    rows = wl.tile_rows(wl.real_text("prose"), ns)
    off, ln = wl.csr_for_fixed(ns, 4096)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        r = tamp_amd.compress_batch(rows.reshape(-1), off, ln, window=10, literal=8, extended=True)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[1:]))
    print(f"| `tamp_batch_compress`, host memory, {ns} x 4 KiB | {t * 1e6:.0f} us | {ns * 4096 / t / 1e6:.1f} MB/s | incl. copies both ways (Python wrapper ~20 us) |")
