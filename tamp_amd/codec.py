"""File-object surface of the reference (``tamp/_c_compressor.pyx``, ``tamp/_c_decompressor.pyx``) over the engine.

Same names, keyword arguments and exception mapping (``tamp/_c_common.pyx:6-16``).  The codec work of every
call goes through the C ABI as a batch of one stream on the GPU; nothing is computed on the host.

The streaming ``Compressor`` encodes one segment per flush point with the window
carried across calls on the host side of the C ABI; FLUSH tokens, ``dictionary_reset``, ``append`` and
``reset_dictionary()`` are produced by the kernel, not emulated on the CPU.
"""
from __future__ import annotations

import builtins
import ctypes as C
from io import BytesIO
from typing import Union

import numpy as np

from . import _lib
from ._lib import TampAmdConf

CHUNK_SIZE = 1 << 20  # tamp/_c_common.pyx:4
MAX_STEP_INPUT = 1 << 28  # compressed bytes offered to one decoder call (the kernels' bit counters are 32 bits wide)


def _error_lookup():
    from . import ExcessBitsError

    return {  # tamp/_c_common.pyx:6-16
        _lib.OUTPUT_FULL: IndexError,
        _lib.INPUT_EXHAUSTED: IndexError,
        _lib.ERROR: Exception,
        _lib.EXCESS_BITS: ExcessBitsError,
        _lib.INVALID_CONF: ValueError,
        _lib.OOB: ValueError,
    }


def _raise_for(res: int):
    if res == _lib.NO_DEVICE:
        detail = _lib.load().tamp_amd_last_error().decode(errors="replace")
        raise _lib.NativeLibraryError(f"tamp_amd: no HIP device available [{detail}]; the codec only runs on the GPU")
    if res == _lib.BAD_ARGUMENT:
        raise ValueError("tamp_amd: bad argument")
    raise _error_lookup().get(res, NotImplementedError)


class Compressor:
    """``tamp.Compressor`` (tamp/_c_compressor.pyx:13-186) over the piece / segment calls of the engine.

    ``write()`` gathers bytes and, whenever ``PIECE_MIN`` of them are waiting, hands them to the GPU as one PIECE
    (``tamp_amd_compress_piece``, finish = 0): the piece ends as a ``tamp_compressor_compress`` call ends -- no FLUSH token,
    no drain -- and the object keeps what the reference's object keeps (window, window position, a run / extended match
    that is still growing, < 8 output bits, < 16 unparsed input bytes).  ``write()`` returns the bytes that reached ``f``
    during the call, like the reference (tamp/_c_compressor.pyx:74-118), and memory stays bounded by ``PIECE_MAX``.
    ``flush()`` / ``reset_dictionary()`` / ``close()`` end the segment on the GPU (FLUSH tokens, ``dictionary_reset``,
    ``append`` are produced by the kernel, not emulated on the host).  With ``lazy_matching`` the cached match of
    compressor.c:576-619 is not carried between pieces: such a stream is gathered until the next flush point.
    """

    PIECE_MIN = 64 << 10  # bytes gathered before a piece is worth a launch
    PIECE_MAX = 8 << 20   # most bytes handed over at once

    def __init__(self, f, *, window: int = 10, literal: int = 8, dictionary=None, lazy_matching: bool = False,
                 extended: bool = True, dictionary_reset: bool = False, append: bool = False, device: int = 0):
        if dictionary is not None and len(dictionary) != (1 << window):
            raise ValueError("Dictionary-window size mismatch.")
        if not (8 <= window <= 15 and 5 <= literal <= 8):
            raise ValueError  # tamp_compressor_init -> TAMP_INVALID_CONF -> ValueError
        if append and (not dictionary_reset or dictionary is not None):
            raise ValueError  # compressor.c:209
        if not hasattr(f, "write"):
            f = builtins.open(str(f), "wb")
            self._close_f_on_close = True
        else:
            self._close_f_on_close = False
        self.f = f
        self._conf = TampAmdConf(window, literal, int(dictionary is not None), int(bool(extended)),
                                 int(bool(dictionary_reset)), int(bool(lazy_matching)))
        self._dictionary_reset = bool(dictionary_reset)
        self._append = bool(append)
        self._window = (C.c_ubyte * (1 << window))()
        if dictionary is not None:
            self._window[:] = bytes(dictionary)
        self._window_pos = C.c_uint16(0)
        self._carry = _lib.TampAmdCarry()
        self._pending = bytearray()
        self._opened = False    # header / append marker already sent
        self._resume = False    # _window holds a carried window (else: fresh stream)
        self._dirty = False     # data was handed over since the last flush point (compressor.c:548)
        self._last_was_flush = self._append  # compressor.c:234
        self._device = device
        self._streaming = not lazy_matching
        _lib.load()  # fail loudly now if the native library is missing

    def _call(self, data: bytes, finish: bool, want_token: bool):
        """One piece on the GPU; returns (bytes written to f, FLUSH token written)."""
        lib = _lib.load()
        n = len(data)
        cap = lib.tamp_amd_compress_bound(n + 272, self._conf.literal, 0) + 8
        out = (C.c_ubyte * cap)()
        written = C.c_size_t(0)
        token = C.c_int(0)
        src = (C.c_ubyte * max(n, 1)).from_buffer_copy(data if n else b"\0")
        first = not self._opened
        res = lib.tamp_amd_compress_piece(
            C.byref(self._conf), int(first and not self._append), int(first and self._append), int(self._resume),
            int(finish), int(want_token), self._window, C.byref(self._window_pos), C.byref(self._carry), out, cap,
            C.byref(written), src, n, C.byref(token), self._device)
        if res < 0:
            _raise_for(res)
        if res == _lib.OUTPUT_FULL:
            raise RuntimeError("tamp_amd: piece output did not fit its bound")  # (cannot happen: cap is the worst case)
        self._opened = self._resume = True
        if written.value:
            self.f.write(bytes(out[: written.value]))
        return written.value, bool(token.value)

    def write(self, data) -> int:
        self._pending += bytes(data)
        if not self._streaming or len(self._pending) < self.PIECE_MIN:
            return 0
        total = 0
        while len(self._pending) >= self.PIECE_MIN:  # bounded pieces, no flush token in between
            piece = bytes(self._pending[: self.PIECE_MAX])
            del self._pending[: len(piece)]
            total += self._call(piece, finish=False, want_token=False)[0]
            self._dirty = True
        return total

    def _segment(self, flush_token: bool) -> int:
        n = len(self._pending)
        if n or self._dirty:
            self._last_was_flush = False  # compressor.c:548
        want_token = bool(flush_token) and not self._last_was_flush  # compressor.c:784
        if n == 0 and not self._dirty and self._opened and not (want_token and self._dictionary_reset):
            return 0  # byte aligned, nothing buffered: the reference's flush writes nothing either
        if self._streaming:
            total = 0
            while len(self._pending) > self.PIECE_MAX:  # (a flush after a very large gathered write: still bounded pieces)
                piece = bytes(self._pending[: self.PIECE_MAX])
                del self._pending[: len(piece)]
                total += self._call(piece, finish=False, want_token=False)[0]
            written, token = self._call(bytes(self._pending), finish=True, want_token=want_token)
            total += written
        else:
            total, token = self._segment_gathered(want_token)
        self._pending.clear()
        self._dirty = False
        if token:
            self._last_was_flush = True
        return total

    def _segment_gathered(self, want_token: bool):
        """lazy_matching: everything since the last flush point as ONE segment (tamp_amd_compress_segment)."""
        lib = _lib.load()
        n = len(self._pending)
        cap = lib.tamp_amd_compress_bound(n, self._conf.literal, 0) + 4
        out = (C.c_ubyte * cap)()
        written = C.c_size_t(0)
        token = C.c_int(0)
        src = (C.c_ubyte * max(n, 1)).from_buffer_copy(bytes(self._pending) if n else b"\0")
        res = lib.tamp_amd_compress_segment(
            C.byref(self._conf), int(not self._opened and not self._append), int(not self._opened and self._append),
            int(self._resume), int(want_token), self._window, C.byref(self._window_pos), out, cap,
            C.byref(written), src, n, C.byref(token), self._device)
        if res < 0:
            _raise_for(res)
        self._opened = self._resume = True
        if written.value:
            self.f.write(bytes(out[: written.value]))
        return written.value, bool(token.value)

    def flush(self, write_token: bool = True) -> int:
        n = self._segment(write_token)
        self.f.flush()
        return n

    def reset_dictionary(self) -> int:
        """``tamp_compressor_reset_dictionary`` (compressor.c:845-881): double FLUSH, then a fresh seeded window."""
        if not self._dictionary_reset:
            raise ValueError  # TAMP_INVALID_CONF
        total = 0
        for _ in range(2):
            self._last_was_flush = False
            total += self._segment(True)
        self._resume = False  # next segment starts from the seed dictionary again (never the custom one)
        self._carry = _lib.TampAmdCarry()
        self._conf.use_custom_dictionary = 0
        self._window_pos = C.c_uint16(0)
        self._last_was_flush = self._append
        self.f.flush()
        return total

    def close(self) -> int:
        bytes_written = self.flush(write_token=self._dictionary_reset)
        if self._close_f_on_close:
            self.f.close()
        return bytes_written

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()


class TextCompressor(Compressor):
    def write(self, data: str) -> int:
        return super().write(data.encode())


#: one-shot ``compress()`` calls of at least this many bytes in the v1 format go to the batch call as ONE stream, which
#: spreads its blocks over all workgroups (tamp_compress_kernel<.., BLOCKM>; csrc/tamp_capi.hip launch_compress_blocks)
ONE_SHOT_BLOCK_MIN = 256 << 10
# The batch tables hold 32-bit lengths: a one-shot call beyond that stays with the streaming object, which cuts the data into
# pieces.  Decoding: 2^29 - 1 compressed bytes per call (kMaxDecodeIn, tamp_common.hpp) and a capacity that fits 32 bits.
ONE_SHOT_MAX_IN = 0xFFFFFFFF
ONE_SHOT_MAX_DECODE_IN = (1 << 29) - 1


def compress(data: Union[bytes, str], *args, **kwargs) -> bytes:
    """``tamp.compress`` (tamp/_c_compressor.pyx:189-199)."""
    if (not args and isinstance(data, (bytes, bytearray, memoryview)) and ONE_SHOT_BLOCK_MIN <= len(data) <= ONE_SHOT_MAX_IN
            and kwargs.get("extended", True) is False and kwargs.get("literal", 8) == 8
            and not kwargs.get("lazy_matching") and not kwargs.get("dictionary_reset") and not kwargs.get("append")
            and set(kwargs) <= {"window", "literal", "dictionary", "extended", "lazy_matching", "dictionary_reset", "append", "device"}):
        # the same bytes as Compressor(...).write(data) + flush(write_token=False): the batch contract
        from .batch import compress_batch

        window, dictionary = kwargs.get("window", 10), kwargs.get("dictionary")
        if dictionary is not None and len(dictionary) != (1 << window):
            raise ValueError("Dictionary-window size mismatch.")
        if not 8 <= window <= 15:
            raise ValueError
        r = compress_batch([bytes(data)], window=window, literal=8, extended=False, dictionary=dictionary,
                           device=kwargs.get("device", 0))
        if int(r.status[0]) != _lib.OK:
            _raise_for(int(r.status[0]))
        return r.stream(0)
    with BytesIO() as f:
        c = TextCompressor(f, *args, **kwargs) if isinstance(data, str) else Compressor(f, *args, **kwargs)
        c.write(data)
        c.flush(write_token=False)
        f.seek(0)
        return f.read()


class Decompressor:
    """``tamp.Decompressor`` (tamp/_c_decompressor.pyx:12-176).

    One resumable decoder object on the device (``tamp_batch_decompress_resume``): ``readinto`` /
    ``read(n)`` pull input from ``f`` a chunk at a time and decode exactly as much as the caller's buffer takes, the
    object's state and window carrying over between calls -- the reference's loop, so its restricted-read and
    stream-break behaviours hold and memory stays bounded.  ``read()`` reads the rest of ``f`` first and decodes it in
    a few large steps.
    """

    def __init__(self, f, *, dictionary=None, device: int = 0):
        if not hasattr(f, "read"):
            f = builtins.open(str(f), "rb")
            self._close_f_on_close = True
        else:
            self._close_f_on_close = False
        self.f = f
        self._device = device
        lib = _lib.load()
        conf = TampAmdConf()
        consumed = C.c_size_t(0)
        header = bytearray()
        while True:  # tamp/_c_decompressor.pyx:50-61
            b = f.read(1)
            if not b:
                _raise_for(_lib.INPUT_EXHAUSTED)
            header += b
            hb = (C.c_ubyte * len(header)).from_buffer_copy(bytes(header))
            res = lib.tamp_amd_read_header(C.byref(conf), hb, len(header), C.byref(consumed))
            if res == _lib.OK:
                break
            if res != _lib.INPUT_EXHAUSTED:
                _raise_for(res)
        if conf.use_custom_dictionary and dictionary is None:
            raise ValueError
        if dictionary is not None and len(dictionary) < (1 << conf.window):
            raise ValueError("Dictionary-window size mismatch.")
        # the decoder object: 16 bytes of state + its window (tamp_decompressor_init with the conf, :72-75)
        self._window_bits = conf.window
        self._stride = (lib.tamp_amd_decoder_state_size(conf.window) + 15) & ~15
        self._slot = np.zeros(self._stride, dtype=np.uint8)
        if dictionary is not None:
            self._slot[16 : 16 + (1 << conf.window)] = np.frombuffer(bytes(dictionary[: 1 << conf.window]), dtype=np.uint8)
        res = lib.tamp_amd_decoder_state_init(self._slot.ctypes.data_as(C.c_void_p), C.byref(conf), conf.window)
        if res < 0:
            _raise_for(res)
        self._pending = b""  # input read from f and not consumed yet

    def _step(self, room: int):
        """One tamp_decompressor_decompress call: offer the pending input and `room` bytes of output."""
        lib = _lib.load()
        n = min(len(self._pending), MAX_STEP_INPUT)  # one call counts bits in 32-bit registers: offer a bounded piece
        src = np.frombuffer(self._pending, dtype=np.uint8, count=n) if n else np.zeros(1, np.uint8)
        out = np.empty(max(room, 1), dtype=np.uint8)
        zero = np.zeros(1, dtype=np.uint64)
        in_len, out_cap = np.array([n], dtype=np.uint32), np.array([room], dtype=np.uint32)
        out_len, consumed = np.zeros(1, dtype=np.uint32), np.zeros(1, dtype=np.uint32)
        status = np.zeros(1, dtype=np.int8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = lib.tamp_batch_decompress_resume(p(self._slot), self._stride, self._window_bits, p(src), p(zero), p(in_len),
                                              p(out), p(zero), p(out_cap), p(out_len), p(status), p(consumed), 1,
                                              _lib.MEM_HOST, self._device, None)
        _lib.check_launch(rc)
        self._pending = self._pending[int(consumed[0]) :]
        self._progress = int(consumed[0]) > 0
        return int(status[0]), out[: int(out_len[0])]

    def readinto(self, buf: bytearray) -> int:  # tamp/_c_decompressor.pyx:77-129
        size, pos = len(buf), 0
        while size:
            res, out = self._step(min(size, 1 << 30))
            buf[pos : pos + len(out)] = out.tobytes()
            pos += len(out)
            size -= len(out)
            if res == _lib.INPUT_EXHAUSTED:
                if self._pending and self._progress:
                    continue  # a bounded piece was consumed, the rest of what is buffered comes next
                chunk = self.f.read(CHUNK_SIZE)
                if not chunk:
                    break
                self._pending += bytes(chunk)
            elif res < 0:
                _raise_for(res)
        return pos

    def read(self, size: int = -1) -> bytearray:  # tamp/_c_decompressor.pyx:131-164
        if size == 0:
            return bytearray()
        if size > 0:
            buf = bytearray(size)
            n = self.readinto(buf)
            del buf[n:]
            return buf
        self._pending += bytes(self.f.read())  # to the end of the stream: everything at once, a few large steps
        out = []
        room = max(CHUNK_SIZE, 8 * len(self._pending))
        while True:
            buf = bytearray(room)
            n = self.readinto(buf)
            if n < room:
                del buf[n:]
                if n:
                    out.append(buf)
                break
            out.append(buf)
            room <<= 1
        return out[0] if len(out) == 1 else bytearray(b"".join(out))

    def close(self):
        if self._close_f_on_close:
            self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()


class TextDecompressor(Decompressor):
    def read(self, *args, **kwargs) -> str:
        return super().read(*args, **kwargs).decode()


def decompress(data: bytes, *args, **kwargs) -> bytearray:
    """``tamp.decompress`` (tamp/_c_decompressor.pyx:184-187).

    A whole v1 stream of ``ONE_SHOT_BLOCK_MIN`` bytes and more goes to the batch decoder as a batch of one, where the
    library decodes ONE long stream with the whole device; everything else -- and whatever that call does
    not finish with the normal end-of-input status -- takes the decoder object below."""
    blob = bytes(data)
    if len(blob) >= ONE_SHOT_BLOCK_MIN and not args and set(kwargs) <= {"dictionary"} and (blob[0] & 1) == 0:
        from .batch import decompress_batch
        # Room for the worst case at once (a call that runs out of room would take the one-wavefront decoder to find that
        # out): the densest v1 token is a 15-byte match in 7 + 8 bits (window 2^8, compressor.c:33-36) -- 8 bytes per byte.
        # The extended format has no useful bound (an RLE token is 241 bytes in 14 bits): 8 x is tried, then 64 x.
        # A blob whose room does not fit the 32-bit capacity, or that the device cannot hold that much of, takes the
        # streaming object below, as every input did before this shortcut existed.
        for factor in ((8,) if not blob[0] & 2 else (8, 64)):
            cap = factor * len(blob) + 64
            if len(blob) > ONE_SHOT_MAX_DECODE_IN or cap > 0xFFFFFFFF:
                break
            try:
                r = decompress_batch([blob], out_cap=cap, dictionary=kwargs.get("dictionary"))
            except (MemoryError, OverflowError, _lib.NativeLibraryError):  # (the object below reports what is really wrong)
                break
            if int(r.status[0]) == _lib.INPUT_EXHAUSTED:
                o, n = int(r.out_off[0]), int(r.out_len[0])
                return bytearray(memoryview(r.out)[o : o + n])  # (one copy of the bytes, not two)
            if int(r.status[0]) != _lib.OUTPUT_FULL:
                break
    with BytesIO(blob) as f:
        d = Decompressor(f, *args, **kwargs)
        return d.read()


def open(f, mode: str = "rb", **kwargs):  # noqa: A001  (tamp/__init__.py:96-105)
    if "r" in mode and "w" in mode:
        raise ValueError
    if "r" in mode:
        return Decompressor(f, **kwargs) if "b" in mode else TextDecompressor(f, **kwargs)
    elif "w" in mode:
        return Compressor(f, **kwargs) if "b" in mode else TextCompressor(f, **kwargs)
    raise ValueError
