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

    def stream_script(self, ops, *, window=10, literal=8, extended=True, dictionary=None, dictionary_reset=False,
                      append=False, lazy_matching=False):
        """Same op script as ``Ref.stream_script`` replayed on the restatement: bytes between flush points form one
        ``oracle_compress_segment`` call; last_was_flush is tracked here (compressor.c:234,548,784,864)."""
        L = self.lib
        L.oracle_compress_segment.restype = C.c_int
        L.oracle_compress_segment.argtypes = [C.POINTER(_OracleConf)] + [C.c_int] * 4 + [
            C.c_void_p, C.POINTER(C.c_uint16), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
            C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        if not (8 <= window <= 15 and 5 <= literal <= 8) or (append and (not dictionary_reset or dictionary is not None)):
            return INVALID_CONF, b""
        conf = _OracleConf(window, literal, int(dictionary is not None), int(extended), int(dictionary_reset),
                           int(lazy_matching))
        win = np.zeros(1 << window, dtype=np.uint8)
        if dictionary is not None:
            win[:] = _u8(dictionary)
        wp = C.c_uint16(0)
        st = dict(opened=False, resume=False, lwf=bool(append))
        pending, emitted = bytearray(), bytearray()

        def segment(flush_token):
            n = len(pending)
            if n:
                st["lwf"] = False
            want = bool(flush_token) and not st["lwf"]
            a = _u8(bytes(pending))
            out = np.zeros(worst_case_compressed_size(n, literal, True) + 8, dtype=np.uint8)
            k, tok = C.c_size_t(0), C.c_int(0)
            r = L.oracle_compress_segment(C.byref(conf), int(not st["opened"] and not append),
                                          int(not st["opened"] and append), int(st["resume"]), int(want), _p(win),
                                          C.byref(wp), _p(a) if n else None, n, _p(out), len(out), C.byref(k),
                                          C.byref(tok))
            emitted.extend(out[: k.value].tobytes())
            if r == OK:
                st["opened"] = st["resume"] = True
                pending.clear()
                if tok.value:
                    st["lwf"] = True
            return r

        for op in ops:
            r = OK
            if op[0] == "write":
                pending.extend(bytes(op[1]))
            elif op[0] == "flush":
                r = segment(bool(op[1]))
            elif op[0] == "close":
                r = segment(dictionary_reset)
            elif op[0] == "reset":
                if not dictionary_reset:
                    return INVALID_CONF, bytes(emitted)
                for _ in range(2):
                    st["lwf"] = False
                    r = segment(True)
                    if r != OK:
                        break
                st["resume"] = False
                conf.use_custom_dictionary = 0
                wp = C.c_uint16(0)
                st["lwf"] = bool(append)
            if r != OK:
                return r, bytes(emitted)
        return OK, bytes(emitted)

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

    def decode_script(self, data, script, *, conf=None, window_bits=15, dictionary=None):
        """One resumable decoder object fed call by call.  ``script`` = [(take, cap), ...]: each call sees the next
        ``take`` unconsumed bytes of ``data`` and ``cap`` bytes of output room; what it does not consume is offered
        again.  ``conf`` = None (header from the stream) or (window, literal, custom, extended, dictionary_reset) with
        ``data`` starting after the header.  -> (init status, [(status, bytes, consumed), ...])"""
        L = self.lib

        class _Dec(C.Structure):
            _fields_ = [("bit_buffer", C.c_uint32), ("window_pos", C.c_uint16), ("bit_buffer_pos", C.c_uint8),
                        ("token_state", C.c_uint8), ("pending_window_offset", C.c_uint16),
                        ("pending_match_size", C.c_uint16), ("conf", C.c_uint8), ("skip_bytes", C.c_uint8),
                        ("flags", C.c_uint8), ("window_bits_max", C.c_uint8)]

        L.oracle_decoder_init.restype = C.c_int
        L.oracle_decoder_init.argtypes = [C.POINTER(_Dec), C.c_void_p, C.c_void_p, C.c_uint8]
        L.oracle_decoder_call.restype = C.c_int
        L.oracle_decoder_call.argtypes = [C.POINTER(_Dec), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        d = _Dec()
        win = np.zeros(1 << 15, dtype=np.uint8)
        if dictionary is not None:
            dd = _u8(dictionary)[: 1 << 15]
            win[: len(dd)] = dd
        oc = _OracleConf(conf[0], conf[1], int(conf[2]), int(conf[3]), int(conf[4]), 0) if conf is not None else None
        r0 = L.oracle_decoder_init(C.byref(d), _p(win), C.byref(oc) if oc is not None else None, window_bits)
        calls = []
        if r0 != OK:
            return r0, calls
        a = _u8(data)
        pos = 0
        for take, cap in script:
            chunk = np.ascontiguousarray(a[pos : pos + take])
            out = np.zeros(cap + 1, dtype=np.uint8)
            w, k = C.c_size_t(0), C.c_size_t(0)
            r = L.oracle_decoder_call(C.byref(d), _p(win), _p(chunk) if len(chunk) else None, len(chunk), _p(out), cap,
                                      C.byref(w), C.byref(k))
            calls.append((r, out[: w.value].tobytes(), k.value))
            pos += k.value
        return r0, calls


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
        L.ref_stream_new.restype = C.c_void_p
        L.ref_stream_new.argtypes = [C.c_int] * 7 + [C.c_void_p, C.POINTER(C.c_int)]
        L.ref_stream_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
        L.ref_stream_flush.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ref_stream_reset_dictionary.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ref_stream_free.argtypes = [C.c_void_p]

    def stream_script(self, ops, *, window=10, literal=8, extended=True, dictionary=None, dictionary_reset=False,
                      append=False, lazy_matching=False, counts=None):
        """Replay ``ops`` -- ("write", bytes) | ("flush", write_token) | ("reset",) | ("close",) -- on ONE reference
        compressor object, the way tamp/_c_compressor.pyx drives it.  Returns (res, bytes emitted so far); ``counts`` (a
        list) receives the bytes each op wrote."""
        d = _u8(dictionary) if dictionary is not None else None
        res = C.c_int(0)
        h = self.lib.ref_stream_new(window, literal, int(dictionary is not None), int(extended), int(dictionary_reset),
                                    int(append), int(lazy_matching), _p(d) if d is not None else None, C.byref(res))
        emitted = bytearray()
        try:
            if res.value < 0:
                return res.value, bytes(emitted)
            for op in ops:
                n = C.c_size_t(0)
                if op[0] == "write":
                    a = _u8(op[1])
                    out = np.zeros(worst_case_compressed_size(len(a), literal, True) + 64, dtype=np.uint8)
                    r = self.lib.ref_stream_write(h, _p(a), len(a), _p(out), len(out), C.byref(n)) if len(a) else 0
                elif op[0] in ("flush", "close"):
                    out = np.zeros(64, dtype=np.uint8)
                    tok = bool(op[1]) if op[0] == "flush" else bool(dictionary_reset)
                    r = self.lib.ref_stream_flush(h, int(tok), _p(out), 32, C.byref(n))
                elif op[0] == "reset":
                    out = np.zeros(64, dtype=np.uint8)
                    r = self.lib.ref_stream_reset_dictionary(h, _p(out), 32, C.byref(n))
                else:
                    raise ValueError(op[0])
                emitted += out[: n.value].tobytes()
                if counts is not None:
                    counts.append(int(n.value))
                if r < 0:
                    return r, bytes(emitted)
            return 0, bytes(emitted)
        finally:
            self.lib.ref_stream_free(h)

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

    def decode_script(self, data, script, *, conf=None, window_bits=15, dictionary=None):
        """``Oracle.decode_script`` on a real reference TampDecompressor object."""
        L = self.lib
        L.ref_decoder_new.restype = C.c_void_p
        L.ref_decoder_new.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.ref_decoder_call.restype = C.c_int
        L.ref_decoder_call.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ref_decoder_free.argtypes = [C.c_void_p]
        dd = _u8(dictionary) if dictionary is not None else None
        r0 = C.c_int(0)
        cf = conf if conf is not None else (-1, 0, 0, 0, 0)
        h = L.ref_decoder_new(int(cf[0]), int(cf[1]), int(cf[2]), int(cf[3]), int(cf[4]), window_bits,
                              _p(dd) if dd is not None else None, len(dd) if dd is not None else 0, C.byref(r0))
        calls = []
        try:
            if r0.value != OK:
                return r0.value, calls
            a = _u8(data)
            pos = 0
            for take, cap in script:
                chunk = np.ascontiguousarray(a[pos : pos + take])
                out = np.zeros(cap + 1, dtype=np.uint8)
                w, k = C.c_size_t(0), C.c_size_t(0)
                r = L.ref_decoder_call(h, _p(chunk) if len(chunk) else None, len(chunk), _p(out), cap, C.byref(w),
                                       C.byref(k))
                calls.append((r, out[: w.value].tobytes(), k.value))
                pos += k.value
            return r0.value, calls
        finally:
            L.ref_decoder_free(h)

    def encode_script(self, ops, *, window=10, literal=8, extended=True, dictionary=None, dictionary_reset=False,
                      append=False, lazy_matching=False):
        """One reference TampCompressor object driven below flush granularity.  ops:
        ("compress", data, cap) / ("poll", cap) / ("flush", write_token, cap) / ("compress_and_flush", data, write_token, cap)
        / ("sink", data).  Input that a call does not consume is dropped (the script decides what to offer next).
        -> (init status, [(status, bytes written, consumed), ...])"""
        L = self.lib
        L.ref_stream_new.restype = C.c_void_p
        L.ref_stream_new.argtypes = [C.c_int] * 7 + [C.c_void_p, C.POINTER(C.c_int)]
        L.ref_stream_free.argtypes = [C.c_void_p]
        sz = C.POINTER(C.c_size_t)
        L.ref_stream_sink.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz]
        L.ref_stream_full.argtypes = [C.c_void_p]
        L.ref_stream_poll.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, sz]
        L.ref_stream_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, sz, sz]
        L.ref_stream_compress_and_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, sz, sz,
                                                    C.c_int]
        L.ref_stream_flush.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, sz]
        for fn in ("ref_stream_sink", "ref_stream_full", "ref_stream_poll", "ref_stream_compress",
                   "ref_stream_compress_and_flush", "ref_stream_flush"):
            getattr(L, fn).restype = C.c_int
        dd = _u8(dictionary) if dictionary is not None else None
        r0 = C.c_int(0)
        h = L.ref_stream_new(window, literal, int(dictionary is not None), int(extended), int(dictionary_reset),
                             int(append), int(lazy_matching), _p(dd) if dd is not None else None, C.byref(r0))
        calls = []
        try:
            if r0.value != OK:
                return r0.value, calls
            for op in ops:
                w, k = C.c_size_t(0), C.c_size_t(0)
                if op[0] == "sink":
                    a = _u8(op[1])
                    L.ref_stream_sink(h, _p(a) if len(a) else None, len(a), C.byref(k))
                    calls.append((OK, b"", k.value))
                    continue
                cap = op[-1]
                out = np.zeros(cap + 1, dtype=np.uint8)
                if op[0] == "poll":
                    r = L.ref_stream_poll(h, _p(out), cap, C.byref(w))
                elif op[0] == "flush":
                    r = L.ref_stream_flush(h, int(op[1]), _p(out), cap, C.byref(w))
                elif op[0] == "compress":
                    a = _u8(op[1])
                    r = L.ref_stream_compress(h, _p(a) if len(a) else None, len(a), _p(out), cap, C.byref(w), C.byref(k))
                else:
                    a = _u8(op[1])
                    r = L.ref_stream_compress_and_flush(h, _p(a) if len(a) else None, len(a), _p(out), cap, C.byref(w),
                                                        C.byref(k), int(op[2]))
                calls.append((r, out[: w.value].tobytes(), k.value))
            return r0.value, calls
        finally:
            L.ref_stream_free(h)
