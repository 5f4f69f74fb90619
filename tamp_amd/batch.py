"""Batch entry points: many independent streams per launch (the hot path of this package).

``compress_batch`` / ``decompress_batch`` take either host data (bytes-likes / numpy) or data already
resident in HBM (torch CUDA tensors); torch is used only as a device-memory container -- pointers go
straight into the C ABI (include/tamp_amd.h).  CSR contract: stream ``i`` is
``data[in_off[i] : in_off[i] + in_len[i]]``; results land in ``out[out_off[i] : out_off[i] + out_len[i]]``.
"""
from __future__ import annotations

import ctypes as C
import numbers
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import TampAmdConf


def compress_bound(n: int, literal: int = 8, dictionary_reset: bool = False) -> int:
    """Worst-case ``.tamp`` size of an ``n``-byte stream (every byte a literal)."""
    return 1 + int(bool(dictionary_reset)) + (n * (literal + 1) + 7) // 8


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


@dataclass
class BatchResult:
    out: object          # uint8 buffer (numpy array or torch tensor) holding every stream's slab
    out_off: object      # uint64/int64 [n]
    out_len: object      # uint32/int32 [n]
    status: object       # int8 [n] -- tamp_res per stream
    in_consumed: object = None
    kernel_ms: float = -1.0
    _keep: object = None  # device tensors the asynchronous launch still reads (converted tables, the dictionary)
    _ws_key: object = None  # (streams, capacity, device) when the slab may serve as the next call's workspace (``reuse=``)

    def stream(self, i: int) -> bytes:
        o, n = int(self.out_off[i]), int(self.out_len[i])
        chunk = self.out[o : o + n]
        return bytes(chunk.cpu().numpy()) if _is_torch(chunk) else chunk.tobytes()

    def streams(self):
        return [self.stream(i) for i in range(len(self.out_len))]


def _conf(window, literal, extended, dictionary, dictionary_reset=False, lazy_matching=False, run_aware=None) -> TampAmdConf:
    # run_aware: None = TAMP_AMD_HINT_AUTO (by stream length), False = PLAIN, True = RUNS
    hint = 0 if run_aware is None else (2 if run_aware else 1)
    return TampAmdConf(window, literal, int(dictionary is not None), int(bool(extended)), int(bool(dictionary_reset)),
                       int(bool(lazy_matching)), hint)


def _np_u8(x) -> np.ndarray:
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x.reshape(-1), dtype=np.uint8)
    return np.frombuffer(bytes(x), dtype=np.uint8)


def _is_int(x) -> bool:
    """A plain integer capacity: Python or numpy integer scalars, not bool (numpy scalars are not ``int`` instances)."""
    return isinstance(x, numbers.Integral) and not isinstance(x, bool)


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def pack_streams(streams: Sequence[bytes]):
    """list of bytes -> (flat uint8 array, in_off uint64[n], in_len uint32[n])."""
    lens = np.fromiter((len(s) for s in streams), dtype=np.uint32, count=len(streams))
    offs = np.zeros(len(streams), dtype=np.uint64)
    if len(streams):
        offs[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
    flat = np.frombuffer(b"".join(bytes(s) for s in streams), dtype=np.uint8) if len(streams) else np.zeros(0, np.uint8)
    return flat, offs, lens


def _slab_offsets(caps: np.ndarray):
    offs = np.zeros(len(caps), dtype=np.uint64)
    if len(caps):
        offs[1:] = np.cumsum(caps.astype(np.uint64))[:-1]
    return offs, int(caps.astype(np.uint64).sum())


def trim(device: int = 0) -> int:
    """Release the device scratch the library keeps between calls on ``device`` (decoder window slabs, the split decoder's
    record slab); returns the bytes released.  ``tamp_amd_trim`` of the C ABI."""
    freed = int(_lib.load().tamp_amd_trim(device))
    if freed < 0:
        _lib.check_launch(freed)
    return freed


def compress_batch(data, in_off=None, in_len=None, *, window: int = 10, literal: int = 8, extended: bool = True,
                   dictionary=None, dictionary_reset: bool = False, lazy_matching: bool = False, out_cap=None,
                   max_in_len: int = 0, device: int = 0, stream=None, timing: bool = False,
                   run_aware=None, reuse=None) -> BatchResult:
    """Compress many independent streams in one launch.

    ``data`` is a list of bytes-likes (host), a flat numpy uint8 array + ``in_off``/``in_len`` (host), or a flat
    torch CUDA uint8 tensor + CUDA ``in_off`` (int64) / ``in_len`` (int32) tensors (device, zero-copy).
    Stream ``i``'s output equals ``tamp.compress(stream_i, window=..., literal=..., dictionary=..., extended=...)``
    of the reference.  ``status[i]`` holds the reference's ``tamp_res`` code for that stream.
    ``run_aware`` picks the kernel build for batches of SHORT messages (same bytes either way): True = the run-aware
    build (long runs listed once, most extended matches settled without a search), False = the lean one-wavefront build
    (faster for short messages), None = the lean build.  Streams of 1 KiB and more (``max_in_len`` >= 1024, or unknown)
    always take the run-aware build since round 3 -- it was the faster one on every text measured -- and the argument
    (like ``$TAMP_AMD_RUNS``) is ignored for them.
    ``reuse`` (device batches with an integer ``out_cap``): the result of an earlier call with the same stream count and
    capacity on the same device; its output slab and tables are overwritten instead of allocating new ones.
    """
    if stream is not None and _is_torch(data):
        # The call makes its tables (capacities, offsets) and its output slab with torch: they have to come into being on the
        # stream the kernels are enqueued on, or the launch races the fill kernels of torch's current stream (and the caching
        # allocator hands the slab's memory on while the launch still writes it).
        import torch

        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=data.device)):
            return compress_batch(data, in_off, in_len, window=window, literal=literal, extended=extended, dictionary=dictionary,
                                  dictionary_reset=dictionary_reset, lazy_matching=lazy_matching, out_cap=out_cap,
                                  max_in_len=max_in_len, device=device, stream=None, timing=timing, run_aware=run_aware, reuse=reuse)
    lib = _lib.load()
    conf = _conf(window, literal, extended, dictionary, dictionary_reset, lazy_matching, run_aware)
    if dictionary is not None and not _is_torch(dictionary) and len(dictionary) != (1 << window):
        raise ValueError("Dictionary-window size mismatch.")  # tamp/_c_compressor.pyx:43-46
    lib.tamp_amd_set_timing(1 if timing else 0)

    if _is_torch(data):
        import torch

        n = int(in_len.numel())
        dev = data.device
        ws = None
        if reuse is not None and out_cap is not None and _is_int(out_cap) and reuse._ws_key == (n, int(out_cap), str(dev)):
            # steady-state callers (one launch per step on the same shapes): the output slab and its tables of the
            # previous call are written again -- no allocation, no fill kernels, nothing on the host but the launch
            ws = reuse
            out, out_off_t, out_len_t, status_t, out_cap_t = ws.out, ws.out_off, ws.out_len, ws.status, ws._keep[3]
        elif out_cap is None or _is_int(out_cap):
            if _is_int(out_cap):
                cap1 = int(out_cap)  # one capacity for every stream: no device round trip to lay the slabs out
                # (max_in_len stays as given: 0 = unknown, which AUTO reads as "long streams" -> the run-aware build;
                # pass max_in_len for batches of short messages)
            else:
                if not max_in_len:
                    max_in_len = int(in_len.max().item()) if n else 0
                cap1 = compress_bound(max_in_len, literal, dictionary_reset)
            out_cap_t = torch.full((n,), cap1, dtype=torch.int32, device=dev)
            out_off_t = torch.arange(n, dtype=torch.int64, device=dev) * cap1
            total = n * cap1
        else:
            out_cap_t = out_cap.to(device=dev, dtype=torch.int32)
            out_off_t = torch.cumsum(out_cap_t.to(torch.int64), 0) - out_cap_t.to(torch.int64)
            total = int(out_cap_t.to(torch.int64).sum().item())
        if ws is None:
            out = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
            out_len_t = torch.empty(n, dtype=torch.int32, device=dev)
            status_t = torch.empty(n, dtype=torch.int8, device=dev)
        dict_t = None
        if dictionary is not None:
            dict_t = dictionary if _is_torch(dictionary) else torch.frombuffer(bytearray(dictionary), dtype=torch.uint8).to(dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        # the launch is asynchronous: converted copies of the tables must outlive it (they ride on the result)
        in_off_t, in_len_t = in_off.to(torch.int64), in_len.to(torch.int32)
        rc = lib.tamp_batch_compress(C.byref(conf), _ptr(dict_t), _ptr(data), _ptr(in_off_t),
                                     _ptr(in_len_t), _ptr(out), _ptr(out_off_t), _ptr(out_cap_t),
                                     _ptr(out_len_t), _ptr(status_t), n, int(max_in_len), _lib.MEM_DEVICE,
                                     dev.index or 0, C.c_void_p(st))
        _lib.check_launch(rc)
        ms = lib.tamp_amd_last_kernel_ms() if timing else -1.0
        res = BatchResult(out, out_off_t, out_len_t, status_t, None, ms, (data, in_off_t, in_len_t, out_cap_t, dict_t))
        res._ws_key = (n, int(out_cap), str(dev)) if (out_cap is not None and _is_int(out_cap)) else None
        return res

    if in_off is None:
        flat, in_off, in_len = pack_streams(data)
    else:
        flat = _np_u8(data)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    n = len(in_len)
    if out_cap is None:
        out_cap = np.array([compress_bound(int(x), literal, dictionary_reset) for x in in_len], dtype=np.uint32)
    elif _is_int(out_cap):
        out_cap = np.full(n, int(out_cap), dtype=np.uint32)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    out_off, total = _slab_offsets(out_cap)
    out = np.zeros(total + 1, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.int8)
    d = _np_u8(dictionary) if dictionary is not None else None
    rc = lib.tamp_batch_compress(C.byref(conf), _ptr(d), _ptr(flat if flat.size else np.zeros(1, np.uint8)),
                                 _ptr(in_off), _ptr(in_len), _ptr(out), _ptr(out_off), _ptr(out_cap), _ptr(out_len),
                                 _ptr(status), n, int(max_in_len), _lib.MEM_HOST, device,
                                 C.c_void_p(stream) if stream else None)
    _lib.check_launch(rc)
    ms = lib.tamp_amd_last_kernel_ms() if timing else -1.0
    return BatchResult(out, out_off, out_len, status, None, ms)


def decompress_batch(data, in_off=None, in_len=None, *, out_cap, dictionary=None, max_window_bits: int = 15,
                     scan_headers: bool = True, device: int = 0, stream=None, timing: bool = False) -> BatchResult:
    """Decompress many independent ``.tamp`` streams in one launch (configuration read from each header).

    ``max_window_bits`` is the reference's limit (larger headers -> TAMP_INVALID_CONF).  With ``scan_headers`` the
    library first looks at the batch's headers to size its on-chip windows for the largest one present (one tiny
    kernel + a 4-byte copy that waits for the stream); pass ``scan_headers=False`` to stay fully asynchronous.

    ``out_cap`` (int or per-stream array) bounds each stream's output.  ``status[i]`` is the reference's code:
    2 (INPUT_EXHAUSTED) on normal completion, 1 (OUTPUT_FULL) if ``out_cap[i]`` was reached with work left,
    -4 / -3 for malformed input.
    """
    if stream is not None and _is_torch(data):
        import torch  # (tables and output slab on the launch stream: see compress_batch)

        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=data.device)):
            return decompress_batch(data, in_off, in_len, out_cap=out_cap, dictionary=dictionary, max_window_bits=max_window_bits,
                                    scan_headers=scan_headers, device=device, stream=None, timing=timing)
    lib = _lib.load()
    lib.tamp_amd_set_timing(1 if timing else 0)
    if not scan_headers:
        max_window_bits |= _lib.WINDOW_BITS_EXACT
    if _is_torch(data):
        import torch

        n = int(in_len.numel())
        dev = data.device
        if _is_int(out_cap):
            out_cap = int(out_cap)
            out_cap_t = torch.full((n,), out_cap, dtype=torch.int32, device=dev)
            out_off_t = torch.arange(n, dtype=torch.int64, device=dev) * out_cap
            total = n * out_cap
        else:
            out_cap_t = out_cap.to(device=dev, dtype=torch.int32)
            out_off_t = torch.cumsum(out_cap_t.to(torch.int64), 0) - out_cap_t.to(torch.int64)
            total = int(out_cap_t.to(torch.int64).sum().item())
        out = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        out_len_t = torch.empty(n, dtype=torch.int32, device=dev)
        status_t = torch.empty(n, dtype=torch.int8, device=dev)
        consumed_t = torch.empty(n, dtype=torch.int32, device=dev)
        dict_t, dict_len = None, 0
        if dictionary is not None:
            dict_t = dictionary if _is_torch(dictionary) else torch.frombuffer(bytearray(dictionary), dtype=torch.uint8).to(dev)
            dict_len = int(dict_t.numel())
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        in_off_t, in_len_t = in_off.to(torch.int64), in_len.to(torch.int32)  # kept alive on the result (async launch)
        rc = lib.tamp_batch_decompress(_ptr(dict_t), dict_len, max_window_bits, _ptr(data), _ptr(in_off_t),
                                       _ptr(in_len_t), _ptr(out), _ptr(out_off_t), _ptr(out_cap_t),
                                       _ptr(out_len_t), _ptr(status_t), _ptr(consumed_t), n, _lib.MEM_DEVICE,
                                       dev.index or 0, C.c_void_p(st))
        _lib.check_launch(rc)
        ms = lib.tamp_amd_last_kernel_ms() if timing else -1.0
        return BatchResult(out, out_off_t, out_len_t, status_t, consumed_t, ms, (data, in_off_t, in_len_t, out_cap_t, dict_t))

    if in_off is None:
        flat, in_off, in_len = pack_streams(data)
    else:
        flat = _np_u8(data)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    n = len(in_len)
    if _is_int(out_cap):
        out_cap = np.full(n, int(out_cap), dtype=np.uint32)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    out_off, total = _slab_offsets(out_cap)
    out = np.zeros(total + 1, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.int8)
    consumed = np.zeros(n, dtype=np.uint32)
    d = _np_u8(dictionary) if dictionary is not None else None
    rc = lib.tamp_batch_decompress(_ptr(d), len(d) if d is not None else 0, max_window_bits,
                                   _ptr(flat if flat.size else np.zeros(1, np.uint8)), _ptr(in_off), _ptr(in_len),
                                   _ptr(out), _ptr(out_off), _ptr(out_cap), _ptr(out_len), _ptr(status),
                                   _ptr(consumed), n, _lib.MEM_HOST, device, C.c_void_p(stream) if stream else None)
    _lib.check_launch(rc)
    ms = lib.tamp_amd_last_kernel_ms() if timing else -1.0
    return BatchResult(out, out_off, out_len, status, consumed, ms)


class DecoderBatch:
    """``n`` resumable decoder objects advanced together, one launch per step (``tamp_batch_decompress_resume``).

    Each object is what the reference's ``TampDecompressor`` is (tamp/_c_src/tamp/decompressor.h:13-57): its state and
    window survive between steps, so a stream may arrive in pieces of any size and be decoded into output buffers of
    any size -- e.g. many network connections decoded as their packets come in.  ``step(chunks, out_caps)`` offers
    object ``i`` the bytes ``chunks[i]`` and ``out_caps[i]`` bytes of room and returns per object the reference's
    status code (1 output full, 2 input exhausted, negative = error), the bytes produced and the number of input bytes
    consumed; what was not consumed must be offered again.  ``conf`` = None reads the header from each stream;
    a ``dictionary`` is placed in every object's window (used by streams whose header has the custom bit).
    """

    def __init__(self, n: int, *, window_bits: int = 15, conf: Optional[TampAmdConf] = None, dictionary=None,
                 device: int = 0):
        lib = _lib.load()
        self.n, self.window_bits, self.device = n, window_bits, device
        self.stride = (lib.tamp_amd_decoder_state_size(window_bits) + 15) & ~15
        if conf is not None and conf.use_custom_dictionary and dictionary is None:
            raise ValueError("custom dictionary expected")
        one = np.zeros(self.stride, dtype=np.uint8)
        if dictionary is not None:  # the window buffer holds the dictionary, as with the reference's object
            d = _np_u8(dictionary)[: 1 << window_bits]
            one[16 : 16 + len(d)] = d
        res = lib.tamp_amd_decoder_state_init(_ptr(one), C.byref(conf) if conf is not None else None, window_bits)
        if res != _lib.OK:
            raise ValueError(f"tamp_amd_decoder_state_init -> {res}")
        self.states = np.tile(one, (n, 1))

    def step(self, chunks: Sequence, out_caps):
        lib = _lib.load()
        flat, in_off, in_len = pack_streams(chunks)
        # one call looks at 2^28 bytes per object at most (32-bit bit counters) and reports what it consumed
        in_len = np.minimum(in_len, np.uint32(1 << 28))
        out_cap = np.ascontiguousarray(np.broadcast_to(np.asarray(out_caps, dtype=np.uint32), (self.n,)))
        out_off, total = _slab_offsets(out_cap)
        out = np.zeros(total + 1, dtype=np.uint8)
        out_len = np.zeros(self.n, dtype=np.uint32)
        status = np.zeros(self.n, dtype=np.int8)
        consumed = np.zeros(self.n, dtype=np.uint32)
        rc = lib.tamp_batch_decompress_resume(_ptr(self.states), self.stride, self.window_bits,
                                              _ptr(flat if flat.size else np.zeros(1, np.uint8)), _ptr(in_off),
                                              _ptr(in_len), _ptr(out), _ptr(out_off), _ptr(out_cap), _ptr(out_len),
                                              _ptr(status), _ptr(consumed), self.n, _lib.MEM_HOST, self.device, None)
        _lib.check_launch(rc)
        outs = [out[int(out_off[i]) : int(out_off[i]) + int(out_len[i])].tobytes() for i in range(self.n)]
        return status, outs, consumed


class EncoderBatch:
    """``n`` compressor objects below flush granularity, advanced together (``tamp_batch_compress_resume``).

    Each object is the reference's ``TampCompressor`` (tamp/_c_src/tamp/compressor.h:13-66): the 16-byte input ring,
    a growing RLE run / extended match, pending output bits and the window all survive between calls, so input may
    arrive in pieces of any size and output buffers may be as small as the reference allows.  Every method runs the
    reference's call of the same name on all objects in one launch and returns per object
    ``(status, bytes, consumed)``.  Whole segments belong to ``compress_batch`` / ``Compressor``.
    """

    def __init__(self, n: int, *, window: int = 10, literal: int = 8, extended: bool = True, dictionary=None,
                 dictionary_reset: bool = False, append: bool = False, lazy_matching: bool = False, device: int = 0):
        lib = _lib.load()
        conf = _conf(window, literal, extended, dictionary, dictionary_reset, lazy_matching)
        self.n, self.window_bits, self.device = n, window, device
        self.stride = (lib.tamp_amd_encoder_state_size(window) + 15) & ~15
        one = np.zeros(self.stride, dtype=np.uint8)
        if dictionary is not None:
            if len(dictionary) != (1 << window):
                raise ValueError("Dictionary-window size mismatch.")
            one[40 : 40 + (1 << window)] = _np_u8(dictionary)
        res = lib.tamp_amd_encoder_state_init(_ptr(one), C.byref(conf), int(append), window)
        if res != _lib.OK:
            raise ValueError(f"tamp_amd_encoder_state_init -> {res}")
        self.states = np.tile(one, (n, 1))

    def _run(self, op: int, chunks, out_caps, write_token: bool = False):
        lib = _lib.load()
        if chunks is None:
            chunks = [b""] * self.n
        flat, in_off, in_len = pack_streams(chunks)
        out_cap = np.ascontiguousarray(np.broadcast_to(np.asarray(out_caps, dtype=np.uint32), (self.n,)))
        out_off, total = _slab_offsets(out_cap)
        out = np.zeros(total + 1, dtype=np.uint8)
        out_len = np.zeros(self.n, dtype=np.uint32)
        status = np.zeros(self.n, dtype=np.int8)
        consumed = np.zeros(self.n, dtype=np.uint32)
        rc = lib.tamp_batch_compress_resume(_ptr(self.states), self.stride, self.window_bits, op, int(write_token),
                                            _ptr(flat if flat.size else np.zeros(1, np.uint8)), _ptr(in_off), _ptr(in_len),
                                            _ptr(out), _ptr(out_off), _ptr(out_cap), _ptr(out_len), _ptr(status),
                                            _ptr(consumed), self.n, _lib.MEM_HOST, self.device, None)
        _lib.check_launch(rc)
        outs = [out[int(out_off[i]) : int(out_off[i]) + int(out_len[i])].tobytes() for i in range(self.n)]
        return status, outs, consumed

    def poll(self, out_caps):
        return self._run(_lib.OP_POLL, None, out_caps)

    def compress(self, chunks: Sequence, out_caps):
        return self._run(_lib.OP_COMPRESS, chunks, out_caps)

    def flush(self, out_caps, write_token: bool = True):
        return self._run(_lib.OP_FLUSH, None, out_caps, write_token)

    def compress_and_flush(self, chunks: Sequence, out_caps, write_token: bool = False):
        return self._run(_lib.OP_COMPRESS_AND_FLUSH, chunks, out_caps, write_token)

    def sink(self, chunks: Sequence):
        """tamp_compressor_sink (compressor.c:665-679): bytes into the rings, host side (a buffer copy, no codec work)."""
        taken = np.zeros(self.n, dtype=np.uint32)
        for i, ch in enumerate(chunks):
            st = self.states[i]
            size, pos = int(st[7]), int(st[8])
            k = min(16 - size, len(ch))
            for j in range(k):
                st[12 + ((pos + size + j) & 15)] = ch[j]
            st[7] = size + k
            taken[i] = k
        return taken

    def full(self):
        return self.states[:, 7] == 16
