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
    """Bits needed to hold ``value``; -1 from 2**31 upwards (the contract of tamp/__init__.py:18-23, whose probe stops
    after 31 shifts)."""
    width = int(value).bit_length()
    return width if width < 32 else -1


_SEED_ALPHABETS = {  # common.c:18-25: sixteen seed characters per literal width; wider literals get the mark-up set
    5: bytes(c & 0x1F for c in b" etaoinshrdlcumw"),
    6: bytes(c & 0x3F for c in b" etaoinshrdlcumw"),
}
_SEED_ALPHABET_WIDE = b" \x000ei>to<ans\nr/."


def _seed_stream(state: int):
    """Marsaglia xorshift32 (13, 17, 5) -- the generator behind the seeded dictionary (common.c:27-35)."""
    m = 0xFFFFFFFF
    while True:
        state = (state ^ (state << 13)) & m
        state ^= state >> 17
        state = (state ^ (state << 5)) & m
        yield state


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
    alphabet = _SEED_ALPHABETS.get(literal, _SEED_ALPHABET_WIDE)
    draws = _seed_stream(seed)
    for group in range(size // 8):  # one 32-bit draw seeds eight bytes, low nibble first
        word = next(draws)
        out[8 * group : 8 * group + 8] = bytes(alphabet[(word >> (4 * k)) & 15] for k in range(8))
    return out


def compute_min_pattern_size(window: int, literal: int) -> int:
    """Shortest match worth a token: 3 once the window outgrows ``10 + 2 * (literal - 5)`` bits, else 2 (common.c:54-56);
    ``ValueError`` outside window 8..15 / literal 5..8, as the reference's Python twin raises."""
    if window not in range(8, 16) or literal not in range(5, 9):
        raise ValueError
    return 3 if window > 10 + 2 * (literal - 5) else 2
