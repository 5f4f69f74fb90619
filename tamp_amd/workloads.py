"""Deterministic synthetic inputs for BASELINE.json's configs (host-side data only, no codec logic).

Thin ctypes front-end over ``libtamp_workloads.so`` (tamp_amd/csrc/workloads.c); see SURVEY.md
section 8(d) for the shapes.  Every generator returns a C-contiguous ``uint8`` array of shape
``(n_streams, stream_len)``: stream ``i`` (global index ``first_index + i``) is row ``i``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtamp_workloads.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise FileNotFoundError(f"{_SO} missing: run __graft_entry__.build() or `make -C tamp_amd/csrc`")
        _lib = C.CDLL(_SO)
        for name in ("wl_synth_text", "wl_telemetry", "wl_lcg_runs", "wl_stress"):
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_int]
            getattr(_lib, name).restype = None
        _lib.wl_telemetry_dictionary.argtypes = [C.c_void_p]
    return _lib


def _gen(name: str, n_streams: int, stream_len: int, first_index: int, threads: int | None) -> np.ndarray:
    out = np.empty((n_streams, stream_len), dtype=np.uint8)
    if n_streams and stream_len:
        getattr(_load(), name)(out.ctypes.data_as(C.c_void_p), n_streams, stream_len, first_index,
                               threads or min(32, os.cpu_count() or 1))
    return out


def synth_text(n_streams: int, stream_len: int = 4096, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """Config 2: Zipf(1.0) word sampler over a 512-word vocabulary, one xorshift32 stream per row."""
    return _gen("wl_synth_text", n_streams, stream_len, first_index, threads)


def telemetry(n_streams: int, stream_len: int = 256, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """Config 5: 7-bit JSON-ish telemetry messages, space padded."""
    return _gen("wl_telemetry", n_streams, stream_len, first_index, threads)


def lcg_runs(n_streams: int, stream_len: int = 512, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """Run-heavy two-letter data (the reference's LCG fuzz corpus recurrence)."""
    return _gen("wl_lcg_runs", n_streams, stream_len, first_index, threads)


def stress(n_streams: int, stream_len: int = 8192, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """xorshift stress shapes: incompressible / long runs / long repeats (row index mod 3)."""
    return _gen("wl_stress", n_streams, stream_len, first_index, threads)


def telemetry_dictionary(seeded_default_256: bytes) -> bytes:
    """Config 5's shared custom dictionary: the seeded 256-byte default with the field skeleton at its tail."""
    buf = np.frombuffer(bytes(seeded_default_256), dtype=np.uint8).copy()
    assert buf.size == 256
    _load().wl_telemetry_dictionary(buf.ctypes.data_as(C.c_void_p))
    return buf.tobytes()


def csr_for_fixed(n_streams: int, stream_len: int):
    """(in_off uint64[n], in_len uint32[n]) for equally sized, densely packed streams."""
    in_off = np.arange(n_streams, dtype=np.uint64) * np.uint64(stream_len)
    in_len = np.full(n_streams, stream_len, dtype=np.uint32)
    return in_off, in_len
