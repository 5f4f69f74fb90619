"""tamp_compress_stream on a reference-named object with a 1 MiB host buffer: wall time per MiB (pieces vs the token-level
resume kernel).  Dev tool (GPU box)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ["TAMP_AMD_STREAM_BUFFER_MB"] = "1"
from tamp_amd import _lib, workloads as wl
import tamp_amd
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
lib.tamp_compress_stream.restype = C.c_int8; lib.tamp_compress_stream.argtypes = [C.c_void_p] * 5 + [sz, sz, CB, C.c_void_p]
mem_read = C.cast(lib.tamp_stream_mem_read, C.c_void_p); mem_write = C.cast(lib.tamp_stream_mem_write, C.c_void_p)
text = (wl.real_text("prose") * 2)[: 6 << 20]
conf = TampConf(window=10, literal=8, extended=1)
window, comp = (C.c_ubyte * 1024)(), (C.c_ubyte * 48)()
assert lib.tamp_compressor_init(comp, C.byref(conf), window) == 0
src = (C.c_ubyte * len(text)).from_buffer_copy(text); dst = (C.c_ubyte * (len(text) + 4096))()
rd, wr = MemReader(C.addressof(src), len(text), 0), MemWriter(C.addressof(dst), len(text) + 4096, 0)
cin, cout = C.c_size_t(0), C.c_size_t(0)
t0 = time.time()
rc = lib.tamp_compress_stream(comp, mem_read, C.byref(rd), mem_write, C.byref(wr), C.byref(cin), C.byref(cout), C.cast(None, CB), None)
dt = time.time() - t0
ok = bytes(dst[: wr.pos]) == tamp_amd.compress(text)
print(f"tamp_compress_stream: rc={rc} {len(text) / dt / 1e6:.1f} MB/s ({dt:.2f} s for {len(text) >> 20} MiB in 1 MiB buffers) same bytes as one-shot: {ok}")
