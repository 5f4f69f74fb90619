"""ctypes binding of libtamp_amd.so (the C ABI of include/tamp_amd.h).

The product path has exactly one implementation: the HIP kernels behind this library.  If the
library is missing or no MI355X is visible, every codec entry point raises -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAMP_AMD_LIB") or os.path.join(_HERE, "libtamp_amd.so")  # (override: instrumented dev builds)

OK, OUTPUT_FULL, INPUT_EXHAUSTED = 0, 1, 2
ERROR, EXCESS_BITS, INVALID_CONF, OOB = -1, -2, -3, -4
NO_DEVICE, BAD_ARGUMENT = -20, -21
OP_POLL, OP_COMPRESS, OP_FLUSH, OP_COMPRESS_AND_FLUSH = 1, 2, 3, 4  # include/tamp_amd.h TAMP_AMD_OP_*
ALL_DEVICES = -1  # include/tamp_amd.h TAMP_AMD_ALL_DEVICES
WINDOW_BITS_EXACT = 0x80  # include/tamp_amd.h: TAMP_AMD_WINDOW_BITS_EXACT
MEM_HOST, MEM_DEVICE = 0, 1

# every symbol include/tamp_amd.h declares (tests/test_capi_symbols.py checks header <-> library <-> this list)
SYMBOLS = (
    "tamp_initialize_dictionary",
    "tamp_compute_min_pattern_size",
    "tamp_window_copy",
    "tamp_amd_compress_bound",
    "tamp_amd_device_count",
    "tamp_amd_version",
    "tamp_amd_last_error",
    "tamp_amd_compress_plan",
    "tamp_batch_compress",
    "tamp_batch_decompress",
    "tamp_amd_decoder_state_size",
    "tamp_amd_decoder_state_init",
    "tamp_batch_decompress_resume",
    "tamp_amd_encoder_state_size",
    "tamp_amd_encoder_state_init",
    "tamp_batch_compress_resume",
    "tamp_amd_compress",
    "tamp_amd_decompress",
    "tamp_amd_compress_segment",
    "tamp_amd_compress_piece",
    "tamp_amd_read_header",
    "tamp_amd_set_timing",
    "tamp_amd_last_kernel_ms",
    "tamp_amd_trim",
    "tamp_amd_host_alloc",
    "tamp_amd_host_free",
    # include/tamp_compat.h: the reference's own symbol names
    "tamp_compressor_init",
    "tamp_compressor_sink",
    "tamp_compressor_full",
    "tamp_compressor_poll",
    "tamp_compressor_compress_cb",
    "tamp_compressor_compress",
    "tamp_compressor_compress_and_flush_cb",
    "tamp_compressor_compress_and_flush",
    "tamp_compressor_flush",
    "tamp_compressor_reset_dictionary",
    "tamp_compress_stream",
    "tamp_decompress_stream",
    "tamp_stream_mem_read",
    "tamp_stream_mem_write",
    "tamp_stream_stdio_read",
    "tamp_stream_stdio_write",
    "tamp_decompressor_read_header",
    "tamp_decompressor_init",
    "tamp_decompressor_decompress_cb",
    "tamp_decompressor_decompress",
)


class TampAmdConf(C.Structure):
    _fields_ = [
        ("window", C.c_uint8),
        ("literal", C.c_uint8),
        ("use_custom_dictionary", C.c_uint8),
        ("extended", C.c_uint8),
        ("dictionary_reset", C.c_uint8),
        ("lazy_matching", C.c_uint8),
        ("input_hint", C.c_uint8),  # 0 auto, 1 plain, 2 run-aware build (include/tamp_amd.h TAMP_AMD_HINT_*)
        ("reserved", C.c_uint8),
    ]


class TampAmdCarry(C.Structure):
    """include/tamp_amd.h TampAmdCarry: what a compressor object holds between two pieces that no flush separates."""
    _fields_ = [
        ("rle_count", C.c_uint8),
        ("ext_count", C.c_uint8),
        ("ext_pos", C.c_uint16),
        ("bit_count", C.c_uint8),
        ("tail_len", C.c_uint8),
        ("reserved", C.c_uint16),
        ("bits", C.c_uint32),
        ("tail", C.c_uint8 * 16),
    ]


class NativeLibraryError(RuntimeError):
    """libtamp_amd.so is missing / unloadable, or no HIP device is available."""


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tamp_amd/csrc`.  tamp_amd has no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname).  If torch is
    # importable, load it FIRST so that this library binds to the runtime torch uses for device memory and
    # streams; loading /opt/rocm's copy first and torch's afterwards leaves the process with two runtimes.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, sz, u8, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, C.c_int
    lib.tamp_initialize_dictionary.argtypes = [vp, sz, u8]
    lib.tamp_initialize_dictionary.restype = None
    lib.tamp_compute_min_pattern_size.argtypes = [u8, u8]
    lib.tamp_compute_min_pattern_size.restype = C.c_int8
    lib.tamp_amd_compress_bound.argtypes = [sz, u8, i32]
    lib.tamp_amd_compress_bound.restype = sz
    lib.tamp_amd_device_count.restype = i32
    lib.tamp_amd_version.restype = C.c_char_p
    lib.tamp_amd_last_error.restype = C.c_char_p
    lib.tamp_batch_compress.argtypes = [C.POINTER(TampAmdConf), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, u32, i32, i32, vp]
    lib.tamp_batch_compress.restype = i32
    lib.tamp_batch_decompress.argtypes = [vp, sz, u8, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, vp]
    lib.tamp_batch_decompress.restype = i32
    lib.tamp_amd_decoder_state_size.argtypes = [u8]
    lib.tamp_amd_decoder_state_size.restype = sz
    lib.tamp_amd_decoder_state_init.argtypes = [vp, C.POINTER(TampAmdConf), u8]
    lib.tamp_amd_decoder_state_init.restype = C.c_int8
    lib.tamp_batch_decompress_resume.argtypes = [vp, sz, u8, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, vp]
    lib.tamp_batch_decompress_resume.restype = i32
    lib.tamp_amd_encoder_state_size.argtypes = [u8]
    lib.tamp_amd_encoder_state_size.restype = sz
    lib.tamp_amd_encoder_state_init.argtypes = [vp, C.POINTER(TampAmdConf), i32, u8]
    lib.tamp_amd_encoder_state_init.restype = C.c_int8
    lib.tamp_batch_compress_resume.argtypes = [vp, sz, u8, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, vp]
    lib.tamp_batch_compress_resume.restype = i32
    lib.tamp_amd_compress.argtypes = [C.POINTER(TampAmdConf), vp, vp, sz, C.POINTER(sz), vp, sz, i32]
    lib.tamp_amd_compress.restype = C.c_int8
    lib.tamp_amd_decompress.argtypes = [vp, sz, vp, sz, C.POINTER(sz), vp, sz, C.POINTER(sz), i32]
    lib.tamp_amd_decompress.restype = C.c_int8
    lib.tamp_amd_compress_segment.argtypes = [C.POINTER(TampAmdConf), i32, i32, i32, i32, vp, C.POINTER(C.c_uint16), vp, sz,
                                              C.POINTER(sz), vp, sz, C.POINTER(i32), i32]
    lib.tamp_amd_compress_segment.restype = C.c_int8
    lib.tamp_amd_compress_piece.argtypes = [C.POINTER(TampAmdConf), i32, i32, i32, i32, i32, vp, C.POINTER(C.c_uint16),
                                            C.POINTER(TampAmdCarry), vp, sz, C.POINTER(sz), vp, sz, C.POINTER(i32), i32]
    lib.tamp_amd_compress_piece.restype = C.c_int8
    lib.tamp_amd_read_header.argtypes = [C.POINTER(TampAmdConf), vp, sz, C.POINTER(sz)]
    lib.tamp_amd_read_header.restype = C.c_int8
    lib.tamp_amd_set_timing.argtypes = [i32]
    lib.tamp_amd_set_timing.restype = None
    lib.tamp_amd_last_kernel_ms.restype = C.c_float
    lib.tamp_amd_trim.argtypes = [i32]
    lib.tamp_amd_trim.restype = C.c_longlong
    lib.tamp_amd_host_alloc.argtypes = [sz]
    lib.tamp_amd_host_alloc.restype = vp
    lib.tamp_amd_host_free.argtypes = [vp]
    lib.tamp_amd_host_free.restype = None
    _lib = lib
    return lib


def check_launch(rc: int) -> None:
    """Library-level return code of a batch call -> exception (per-stream codes are in status[])."""
    if rc == OK:
        return
    if rc == NO_DEVICE:
        detail = load().tamp_amd_last_error().decode(errors="replace")
        raise NativeLibraryError("tamp_amd: no HIP device available or a HIP runtime call failed "
                                 f"[{detail}]; the codec only runs on the GPU")
    if rc == BAD_ARGUMENT:
        raise ValueError("tamp_amd: bad argument (invalid pointers / sizes / configuration)")
    raise RuntimeError(f"tamp_amd: unexpected return code {rc}")
