"""``import tamp`` resolves to the MI355X engine: the drop-in name for the reference's Python package.

BASELINE.json's north star keeps "the tamp.compress/decompress Python surface"; this alias makes code written for the
reference (``tamp/__init__.py``: ``compress``, ``decompress``, ``Compressor``, ``Decompressor``, ``TextCompressor``,
``TextDecompressor``, ``open``, ``initialize_dictionary``, ``compute_min_pattern_size``, ``bit_size``,
``ExcessBitsError``) run on ``tamp_amd`` unchanged.  Everything is re-exported from ``tamp_amd``; nothing is implemented here.
"""
import tamp_amd as _engine
from tamp_amd import *  # noqa: F401,F403
from tamp_amd import open  # noqa: F401,A004  (``*`` leaves builtins' names alone)

__all__ = list(_engine.__all__)
__version__ = _engine.__version__
