"""decompress_amd — MI355X-native many-stream DEFLATE engine.

Host-side mirror of the reference's hot-path interface (mirage/decompress
`De.Inf.Ns`, `Zl.Inf.Ns`, `De.Higher`, `Zl.Higher`) over the C ABI of
include/mdeflate.h.  All compute runs in hand-written HIP kernels
(decompress_amd/csrc); there is no CPU path in this package.
"""
from . import _lib  # noqa: F401
from .engine import (Engine, Error, STATUS_NAMES, FORMAT_DEFLATE, FORMAT_ZLIB, FORMAT_GZIP,  # noqa: F401
                     DRIVER_ZL, DRIVER_HIGHER, DRIVER_CLI)
from . import de, zl, gz, lz, lzo  # noqa: F401

__all__ = ["Engine", "Error", "de", "zl", "gz", "STATUS_NAMES", "FORMAT_DEFLATE", "FORMAT_ZLIB", "FORMAT_GZIP"]
