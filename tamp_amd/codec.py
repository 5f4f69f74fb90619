"""File-object surface of the reference (``tamp/_c_compressor.pyx``, ``tamp/_c_decompressor.pyx``) over the engine.

Same names, keyword arguments and exception mapping (``tamp/_c_common.pyx:6-16``).  The codec work of every
call goes through the C ABI as a batch of one stream on the GPU; nothing is computed on the host.

Scope in this release (SURVEY.md section 8f row 2 is "next"): a ``Compressor`` produces ONE segment -- data is
gathered by ``write()`` and encoded by ``flush(write_token=False)`` / ``close()``; mid-stream FLUSH tokens,
``dictionary_reset`` / ``append`` are rejected with ``NotImplementedError`` rather than emulated on the CPU.
"""
from __future__ import annotations

import builtins
import ctypes as C
from io import BytesIO
from typing import Union

from . import _lib
from ._lib import TampAmdConf

CHUNK_SIZE = 1 << 20  # tamp/_c_common.pyx:4


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
    """``tamp.Compressor`` (tamp/_c_compressor.pyx:13-186)."""

    def __init__(self, f, *, window: int = 10, literal: int = 8, dictionary=None, lazy_matching: bool = False,
                 extended: bool = True, dictionary_reset: bool = False, append: bool = False, device: int = 0):
        if dictionary is not None and len(dictionary) != (1 << window):
            raise ValueError("Dictionary-window size mismatch.")
        if not (8 <= window <= 15 and 5 <= literal <= 8):
            raise ValueError  # tamp_compressor_init -> TAMP_INVALID_CONF -> ValueError
        if dictionary_reset or append:
            raise NotImplementedError("dictionary_reset / append: SURVEY.md section 8f row 2 (next)")
        if not hasattr(f, "write"):
            f = builtins.open(str(f), "wb")
            self._close_f_on_close = True
        else:
            self._close_f_on_close = False
        self.f = f
        self._conf = TampAmdConf(window, literal, int(dictionary is not None), int(bool(extended)), 0,
                                 int(bool(lazy_matching)))
        self._dictionary = bytes(dictionary) if dictionary is not None else None
        self._dictionary_reset = dictionary_reset
        self._pending = bytearray()
        self._emitted = False
        self._device = device
        _lib.load()  # fail loudly now if the native library is missing

    def write(self, data) -> int:
        if self._emitted:
            raise NotImplementedError("writing after a flush needs window carry-over: SURVEY.md section 8f row 2")
        self._pending += bytes(data)
        return 0

    def flush(self, write_token: bool = True) -> int:
        if write_token:
            raise NotImplementedError("mid-stream FLUSH token: SURVEY.md section 8f row 2 (next)")
        if self._emitted:
            return 0
        lib = _lib.load()
        n = len(self._pending)
        cap = lib.tamp_amd_compress_bound(n, self._conf.literal, 0)
        out = (C.c_ubyte * cap)()
        written = C.c_size_t(0)
        src = (C.c_ubyte * max(n, 1)).from_buffer_copy(bytes(self._pending) if n else b"\0")
        d = (C.c_ubyte * len(self._dictionary)).from_buffer_copy(self._dictionary) if self._dictionary else None
        res = lib.tamp_amd_compress(C.byref(self._conf), d, out, cap, C.byref(written), src, n, self._device)
        if res < 0:
            _raise_for(res)
        self._emitted = True
        self._pending.clear()
        self.f.write(bytes(out[: written.value]))
        self.f.flush()
        return written.value

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


def compress(data: Union[bytes, str], *args, **kwargs) -> bytes:
    """``tamp.compress`` (tamp/_c_compressor.pyx:189-199)."""
    with BytesIO() as f:
        c = TextCompressor(f, *args, **kwargs) if isinstance(data, str) else Compressor(f, *args, **kwargs)
        c.write(data)
        c.flush(write_token=False)
        f.seek(0)
        return f.read()


class Decompressor:
    """``tamp.Decompressor`` (tamp/_c_decompressor.pyx:12-176).

    The whole stream read so far is decoded on the device; ``read(n)`` / ``readinto`` hand out slices, so the
    restricted-read and stream-break behaviours of the reference's tests hold.
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
        self._dictionary = bytes(dictionary[: 1 << conf.window]) if dictionary is not None else None
        self._compressed = bytearray(header)
        self._out = bytearray()
        self._pos = 0

    def _decode_all(self):
        more = self.f.read()
        if not more and self._out_valid:
            return
        self._compressed += more
        lib = _lib.load()
        n = len(self._compressed)
        src = (C.c_ubyte * n).from_buffer_copy(bytes(self._compressed))
        d = (C.c_ubyte * len(self._dictionary)).from_buffer_copy(self._dictionary) if self._dictionary else None
        cap = max(4096, 8 * n)
        while True:
            out = (C.c_ubyte * cap)()
            written, consumed = C.c_size_t(0), C.c_size_t(0)
            res = lib.tamp_amd_decompress(d, len(self._dictionary) if self._dictionary else 0, out, cap,
                                          C.byref(written), src, n, C.byref(consumed), self._device)
            if res == _lib.OUTPUT_FULL:
                cap *= 4
                continue
            if res < 0:
                _raise_for(res)
            break
        self._out = bytearray(out[: written.value])
        self._out_valid = True

    _out_valid = False

    def readinto(self, buf: bytearray) -> int:
        self._decode_all()
        chunk = self._out[self._pos : self._pos + len(buf)]
        buf[: len(chunk)] = chunk
        self._pos += len(chunk)
        return len(chunk)

    def read(self, size: int = -1) -> bytearray:
        if size == 0:
            return bytearray()
        self._decode_all()
        end = len(self._out) if size < 0 else min(len(self._out), self._pos + size)
        chunk = bytearray(self._out[self._pos : end])
        self._pos = end
        return chunk

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
    """``tamp.decompress`` (tamp/_c_decompressor.pyx:184-187)."""
    with BytesIO(bytes(data)) as f:
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
