"""CPU tier: the C-ABI library loads and exports every symbol include/tamp_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = set()
    for header in ("tamp_amd.h", "tamp_compat.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(tamp_[a-z0-9_]+)\s*\(", text))
    names.discard("tamp_callback_t")
    return sorted(names)


def test_header_declares_the_boundary():
    names = _declared_functions()
    for must in ("tamp_batch_compress", "tamp_batch_decompress", "tamp_initialize_dictionary",
                 "tamp_compute_min_pattern_size", "tamp_amd_compress", "tamp_amd_decompress", "tamp_amd_read_header"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtamp_amd.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)  # loading needs libamdhip64 but no GPU
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/tamp_amd.h but not exported"
    assert set(_lib.SYMBOLS) == set(_declared_functions())


def test_host_helpers_need_no_device(oracle):
    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtamp_amd.so not built")
    lib = _lib.load()
    buf = (ctypes.c_ubyte * 1024)()
    for lit in (5, 6, 7, 8):
        lib.tamp_initialize_dictionary(buf, 1024, lit)
        assert bytes(buf) == oracle.initialize_dictionary(1024, lit)
    for w in range(8, 16):
        for lit in range(5, 9):
            assert lib.tamp_compute_min_pattern_size(w, lit) == oracle.min_pattern_size(w, lit)
    assert lib.tamp_amd_compress_bound(4096, 8, 0) == 4609
    # the compress launch plan (round 6: EIGHT workgroups per CU for the run-aware builds -- 64 VGPRs, 1,024 buckets at window 2^10 --
    # and the largest block that allows it)
    def plan(w, n, lazy=0):
        v = [ctypes.c_uint32(0) for _ in range(4)]
        assert lib.tamp_amd_compress_plan(w, n, lazy, *[ctypes.byref(x) for x in v]) == 0
        return tuple(x.value for x in v)
    blk, lds, threads, per_cu = plan(10, 4096)
    assert (blk, threads, per_cu) == (1024, 256, 8) and lds <= 20480
    assert plan(10, 0)[0] == 1024 and plan(10, 1 << 20)[0] == 1024
    assert plan(10, 256)[:1] == (256,) and plan(10, 256)[2] == 64          # short messages: one wavefront, the whole message
    assert plan(8, 4096)[0] % 256 == 0 and plan(8, 4096)[3] == 8            # smaller windows: the same
    assert plan(11, 4096)[3] == plan(11, 1024)[3] and plan(11, 4096)[0] >= 1024   # larger windows: the largest block at the occupancy of a 1,024-position one
    for w in range(8, 16):
        b, l, t, c = plan(w, 4096)
        assert 64 <= b <= 2048 and l <= 160 * 1024 and c >= 1, (w, b, l, c)
        bl, ll, tl, cl = plan(w, 4096, 1)
        assert cl <= 5 and ll <= 160 * 1024
    assert lib.tamp_amd_compress_plan(7, 4096, 0, None, None, None, None) != 0
    # a tuning override of the block size cannot outgrow the cursor region, which doubles as the sorted query list and the walk's
    # piece / step tables (1,024 buckets at window 2^10: 2 KB -> 1,024 positions at most; 2,048 buckets elsewhere: 2,048)
    os.environ["TAMP_AMD_BLK"] = "2048"
    try:
        assert plan(10, 4096)[0] == 1024 and plan(9, 4096)[0] == 2048
    finally:
        del os.environ["TAMP_AMD_BLK"]
    conf = _lib.TampAmdConf()
    consumed = ctypes.c_size_t(0)
    hdr = (ctypes.c_ubyte * 2)(0x5A, 0)
    assert lib.tamp_amd_read_header(ctypes.byref(conf), hdr, 1, ctypes.byref(consumed)) == 0
    assert (conf.window, conf.literal, conf.extended, conf.use_custom_dictionary, consumed.value) == (10, 8, 1, 0, 1)
    hdr = (ctypes.c_ubyte * 2)(0x59, 1)
    assert lib.tamp_amd_read_header(ctypes.byref(conf), hdr, 2, ctypes.byref(consumed)) == -3
    assert lib.tamp_amd_read_header(ctypes.byref(conf), hdr, 1, ctypes.byref(consumed)) == 2


def test_no_device_is_loud():
    """Without a GPU the codec entry points must fail, never fall back to a CPU path."""
    import tamp_amd
    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        with pytest.raises(tamp_amd.NativeLibraryError):
            tamp_amd.compress(b"abc")
        return
    if _lib.load().tamp_amd_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(tamp_amd.NativeLibraryError):
        tamp_amd.compress(b"abc")
    with pytest.raises(tamp_amd.NativeLibraryError):
        tamp_amd.decompress(bytes.fromhex("58b3041c8100030000"))
    with pytest.raises(tamp_amd.NativeLibraryError):
        tamp_amd.compress_batch([b"abc", b"def"])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tamp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".c", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libtamp_oracle" not in text and "libtamp_ref" not in text, f


def test_c_caller_compiles_and_links(tmp_path):
    """A plain C translation unit (the shape of tools/c-profiler/main.c:52-54 in the reference) builds against
    include/tamp_compat.h and links with libtamp_amd.so; host helpers run without a device."""
    import shutil
    import subprocess

    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH) or not shutil.which("gcc"):
        pytest.skip("library or gcc missing")
    src = tmp_path / "caller.c"
    src.write_text(
        '#include <stdio.h>\n#include "tamp_compat.h"\n'
        "int main(void) {\n"
        "  unsigned char window[1 << 10], out[64];\n"
        "  TampCompressor c; TampDecompressor d; TampConf conf = {.window = 10, .literal = 8, .extended = 1};\n"
        '  printf("%zu %zu %zu %d ", sizeof(TampConf), sizeof(TampCompressor), sizeof(TampDecompressor),\n'
        "         (int)tamp_compute_min_pattern_size(10, 8));\n"
        "  if (tamp_compressor_init(&c, &conf, window) != TAMP_OK) return 1;\n"
        "  if (tamp_decompressor_init(&d, NULL, window, 10) != TAMP_OK) return 2;\n"
        "  if (tamp_compressor_init(&c, &(TampConf){.window = 10, .literal = 4}, window) != TAMP_INVALID_CONF) return 3;\n"
        '  printf("%02x%02x%02x%02x\\n", window[0], window[1], window[2], window[3]);\n'
        "  (void)out; (void)tamp_compressor_compress_and_flush; (void)tamp_decompressor_decompress;\n"
        "  return 0;\n}\n"
    )
    exe = tmp_path / "caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(exe), "-L", libdir, "-ltamp_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert out[:4] == ["2", "48", "24", "2"]
    assert out[4] == "002e2f2f"  # tests/test_pseudorandom.py:22-24: dictionary starts 00 '.' '/' '/'


def test_window_copy_host_helper():
    """tamp_window_copy (common.h:424): destination wraps, source does not, every source byte is read before any is
    overwritten -- checked against a straight restatement for all distances around the overlap cases."""
    import ctypes as C
    import random

    from tamp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtamp_amd.so not built")
    lib = C.CDLL(_lib.LIB_PATH)
    lib.tamp_window_copy.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.c_uint16, C.c_uint8, C.c_uint16]
    lib.tamp_window_copy.restype = None
    rng = random.Random(1)
    W = 256
    for _ in range(2000):
        n = rng.randrange(0, 135)
        off = rng.randrange(0, W - n + 1)
        pos = (off + rng.randrange(-140, 141)) % W if rng.random() < 0.7 else rng.randrange(W)
        start = bytes(rng.randrange(256) for _ in range(W))
        buf = (C.c_ubyte * W).from_buffer_copy(start)
        p = C.c_uint16(pos)
        lib.tamp_window_copy(buf, C.byref(p), off, n, W - 1)
        want = bytearray(start)
        src = start[off : off + n]
        for i, b in enumerate(src):
            want[(pos + i) % W] = b
        assert bytes(buf) == bytes(want) and p.value == (pos + n) % W, (off, pos, n)
