"""tamp_amd -- MI355X-native batch codec for the Tamp ``.tamp`` format.

Python surface of the reference kept name-for-name (``tamp/__init__.py``, ``tamp/_c_compressor.pyx``,
``tamp/_c_decompressor.pyx``): ``compress``, ``decompress``, ``Compressor``, ``Decompressor``,
``TextCompressor``, ``TextDecompressor``, ``open``, ``initialize_dictionary``,
``compute_min_pattern_size``, ``bit_size``, ``ExcessBitsError`` -- plus the batch entry points this
package exists for, ``compress_batch`` / ``decompress_batch``.

Every codec call runs the HIP kernels in ``libtamp_amd.so`` on an MI355X.  There is no CPU fallback:
without the library or a device the calls raise ``NativeLibraryError``.
"""
from __future__ import annotations

__version__ = "0.1.0"


class ExcessBitsError(Exception):
    """Provided data has more bits than expected ``literal`` bits."""  # tamp/__init__.py:14-15


from ._lib import NativeLibraryError  # noqa: E402
from .batch import BatchResult, DecoderBatch, EncoderBatch, compress_batch, compress_bound, decompress_batch, pack_streams, trim  # noqa: E402
from .codec import (  # noqa: E402
    Compressor,
    Decompressor,
    TextCompressor,
    TextDecompressor,
    compress,
    decompress,
)
from .codec import open as open  # noqa: E402,A001
from .host import bit_size, compute_min_pattern_size, initialize_dictionary  # noqa: E402
from .sharding import partition_streams, shard_for_rank  # noqa: E402

__all__ = [
    "ExcessBitsError", "NativeLibraryError", "BatchResult", "compress", "decompress", "compress_batch",
    "decompress_batch", "compress_bound", "pack_streams", "trim", "DecoderBatch", "EncoderBatch", "Compressor", "Decompressor", "TextCompressor",
    "TextDecompressor", "open", "initialize_dictionary", "compute_min_pattern_size", "bit_size",
    "partition_streams", "shard_for_rank",
]
