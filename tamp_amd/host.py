"""Host-side helpers of the reference's Python surface (no device needed).

Mirrors ``tamp/__init__.py:18-70``: ``bit_size``, ``initialize_dictionary``, ``compute_min_pattern_size``.
The dictionary bytes come from the C ABI's ``tamp_initialize_dictionary`` (include/tamp_amd.h) when the
default seed is used, so the Python surface and the kernels' default windows are the same bytes.
"""
from __future__ import annotations

import ctypes as C

from . import _lib

_DEFAULT_SEED = 3758097560  # tamp/__init__.py:37, common.c:38


def bit_size(value: int) -> int:
    """Number of bits needed to represent ``value`` (tamp/__init__.py:18-23); -1 if it exceeds 32 bits."""
    for i in range(32):
        if not value:
            return i
        value >>= 1
    return -1


def _xorshift32(seed: int):
    while True:
        seed ^= (seed << 13) & 0xFFFFFFFF
        seed ^= (seed >> 17) & 0xFFFFFFFF
        seed ^= (seed << 5) & 0xFFFFFFFF
        yield seed


def initialize_dictionary(source, seed=None, literal: int = 8) -> bytearray:
    """Seed a window buffer (tamp/__init__.py:33-63).  ``source`` is a size or a bytearray filled in place."""
    if not (5 <= literal <= 8):
        raise ValueError("literal must be between 5 and 8")
    if seed == 0:
        return bytearray(source)
    out = source if isinstance(source, bytearray) else bytearray(source)
    size = len(out)
    if seed is None or seed == _DEFAULT_SEED:
        lib = _lib.load()
        buf = (C.c_ubyte * max(size, 1)).from_buffer(out) if size else None
        if size:
            n8 = size & ~7  # the reference's Python generator emits whole 8-byte groups only
            lib.tamp_initialize_dictionary(buf, n8, literal)
        return out
    if literal <= 5:
        chars = bytes(c & 0x1F for c in b" etaoinshrdlcumw")
    elif literal <= 6:
        chars = bytes(c & 0x3F for c in b" etaoinshrdlcumw")
    else:
        chars = b" \x000ei>to<ans\nr/."
    gen = _xorshift32(seed)
    i = 0
    for _ in range(size >> 3):
        value = next(gen)
        for _ in range(8):
            out[i] = chars[value & 0x0F]
            value >>= 4
            i += 1
    return out


def compute_min_pattern_size(window: int, literal: int) -> int:
    """tamp/__init__.py:66-70 / common.c:54-56."""
    if not (7 < window < 16 and 4 < literal < 9):
        raise ValueError
    return 2 + (window > (10 + ((literal - 5) << 1)))
