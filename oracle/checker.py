"""ctypes front-ends for the TEST-ONLY checkers.

``Oracle``  -> oracle/libtamp_oracle.so   (this repo's CPU restatement, tamp_oracle.c)
``Ref``     -> oracle/_ref/libtamp_ref.so (the reference C library itself, built in place by
               oracle/Makefile from /root/reference; absent if it was never built)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package ``tamp_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libtamp_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libtamp_ref.so")

OK, OUTPUT_FULL, INPUT_EXHAUSTED = 0, 1, 2
ERROR, EXCESS_BITS, INVALID_CONF, OOB = -1, -2, -3, -4


class _OracleConf(C.Structure):
    _fields_ = [
        ("window", C.c_uint8),
        ("literal", C.c_uint8),
        ("use_custom_dictionary", C.c_uint8),
        ("extended", C.c_uint8),
        ("dictionary_reset", C.c_uint8),
        ("lazy_matching", C.c_uint8),
    ]


def worst_case_compressed_size(n: int, literal: int = 8, dictionary_reset: bool = False) -> int:
    """1 header byte (+1) + ceil(n*(literal+1)/8): every byte a literal."""
    return 1 + int(dictionary_reset) + (n * (literal + 1) + 7) // 8


def _u8(buf) -> np.ndarray:
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf, dtype=np.uint8)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class BatchResult:
    out: np.ndarray
    out_off: np.ndarray
    out_len: np.ndarray
    status: np.ndarray
    seconds: float

    def stream(self, i: int) -> bytes:
        o = int(self.out_off[i])
        return self.out[o : o + int(self.out_len[i])].tobytes()


class _Base:
    prefix = ""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        self.lib = C.CDLL(path)
        self.path = path

    # ---- batch (CSR contract identical to include/tamp_amd.h) ----
    def _batch(self, decompress, in_buf, in_off, in_len, out_cap, *, window=10, literal=8, extended=True,
               dictionary=None, lazy=False, threads=1) -> BatchResult:
        fn = getattr(self.lib, self.prefix + "_batch")
        fn.restype = C.c_double
        fn.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        in_buf = _u8(in_buf)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
        out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
        n = len(in_len)
        out_off = np.zeros(n, dtype=np.uint64)
        if n:
            out_off[1:] = np.cumsum(out_cap.astype(np.uint64))[:-1]
        out = np.zeros(int(out_cap.astype(np.uint64).sum()) + 1, dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.int8)
        d = _u8(dictionary) if dictionary is not None else None
        secs = fn(int(decompress), window, literal, int(dictionary is not None), int(extended), int(lazy),
                  _p(d) if d is not None else None, len(d) if d is not None else 0, _p(in_buf), _p(in_off),
                  _p(in_len), _p(out), _p(out_off), _p(out_cap), _p(out_len), _p(status), n, threads)
        return BatchResult(out, out_off, out_len, status, secs)

    def compress_batch(self, in_buf, in_off, in_len, out_cap=None, **kw) -> BatchResult:
        if out_cap is None:
            lit = kw.get("literal", 8)
            out_cap = np.array([worst_case_compressed_size(int(x), lit) for x in np.asarray(in_len)], dtype=np.uint32)
        return self._batch(0, in_buf, in_off, in_len, out_cap, **kw)

    def decompress_batch(self, in_buf, in_off, in_len, out_cap, *, dictionary=None, threads=1) -> BatchResult:
        return self._batch(1, in_buf, in_off, in_len, out_cap, dictionary=dictionary, threads=threads)


class Oracle(_Base):
    prefix = "oracle"

    def __init__(self, path: str = ORACLE_SO):
        super().__init__(path)
        L = self.lib
        L.oracle_compress.restype = C.c_int
        L.oracle_compress.argtypes = [C.POINTER(_OracleConf), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.POINTER(C.c_size_t)]
        L.oracle_decompress.restype = C.c_int
        L.oracle_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint8, C.c_void_p,
                                        C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.oracle_initialize_dictionary.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8]
        L.oracle_min_pattern_size.restype = C.c_int
        L.oracle_min_pattern_size.argtypes = [C.c_uint8, C.c_uint8]

    def initialize_dictionary(self, size: int, literal: int = 8) -> bytes:
        b = np.zeros(size, dtype=np.uint8)
        self.lib.oracle_initialize_dictionary(_p(b), size, literal)
        return b.tobytes()

    def min_pattern_size(self, window: int, literal: int) -> int:
        return self.lib.oracle_min_pattern_size(window, literal)

    def compress(self, data, *, window=10, literal=8, extended=True, dictionary=None, dictionary_reset=False,
                 lazy_matching=False, cap=None):
        """-> (status, bytes)"""
        a = _u8(data)
        conf = _OracleConf(window, literal, int(dictionary is not None), int(extended), int(dictionary_reset),
                           int(lazy_matching))
        if cap is None:
            cap = worst_case_compressed_size(len(a), literal, dictionary_reset) + 8
        out = np.zeros(cap + 1, dtype=np.uint8)
        n = C.c_size_t(0)
        d = _u8(dictionary) if dictionary is not None else None
        r = self.lib.oracle_compress(C.byref(conf), _p(d) if d is not None else None, _p(a), len(a), _p(out), cap,
                                     C.byref(n))
        return r, out[: n.value].tobytes()

    def decompress(self, data, *, dictionary=None, cap=None, max_window_bits=15):
        """-> (status, bytes, consumed)"""
        a = _u8(data)
        if cap is None:
            cap = max(64, len(a) * 256)
        out = np.zeros(cap + 1, dtype=np.uint8)
        n, c = C.c_size_t(0), C.c_size_t(0)
        d = _u8(dictionary) if dictionary is not None else None
        r = self.lib.oracle_decompress(_p(a), len(a), _p(d) if d is not None else None, len(d) if d is not None else 0,
                                       max_window_bits, _p(out), cap, C.byref(n), C.byref(c))
        return r, out[: n.value].tobytes(), c.value


class Ref(_Base):
    prefix = "ref"

    def __init__(self, path: str = REF_SO):
        super().__init__(path)
        L = self.lib
        L.ref_compress.restype = C.c_int
        L.ref_compress.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
        L.ref_decompress.restype = C.c_int
        L.ref_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ref_initialize_dictionary.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.ref_min_pattern_size.restype = C.c_int

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def initialize_dictionary(self, size: int, literal: int = 8) -> bytes:
        b = np.zeros(size, dtype=np.uint8)
        self.lib.ref_initialize_dictionary(_p(b), size, literal)
        return b.tobytes()

    def min_pattern_size(self, window: int, literal: int) -> int:
        return self.lib.ref_min_pattern_size(window, literal)

    def sizes(self):
        return (self.lib.ref_sizeof_conf(), self.lib.ref_sizeof_compressor(), self.lib.ref_sizeof_decompressor())

    def compress(self, data, *, window=10, literal=8, extended=True, dictionary=None, dictionary_reset=False,
                 lazy_matching=False, cap=None):
        a = _u8(data)
        if cap is None:
            cap = worst_case_compressed_size(len(a), literal, dictionary_reset) + 8
        out = np.zeros(cap + 1, dtype=np.uint8)
        n = C.c_size_t(0)
        d = _u8(dictionary) if dictionary is not None else None
        r = self.lib.ref_compress(window, literal, int(dictionary is not None), int(extended), int(dictionary_reset),
                                  int(lazy_matching), _p(d) if d is not None else None, _p(a), len(a), _p(out), cap,
                                  C.byref(n))
        return r, out[: n.value].tobytes()

    def decompress(self, data, *, dictionary=None, cap=None, max_window_bits=15):
        a = _u8(data)
        if cap is None:
            cap = max(64, len(a) * 256)
        out = np.zeros(cap + 1, dtype=np.uint8)
        n, c = C.c_size_t(0), C.c_size_t(0)
        d = _u8(dictionary) if dictionary is not None else None
        r = self.lib.ref_decompress(_p(a), len(a), _p(d) if d is not None else None, len(d) if d is not None else 0,
                                    max_window_bits, _p(out), cap, C.byref(n), C.byref(c))
        return r, out[: n.value].tobytes(), c.value
