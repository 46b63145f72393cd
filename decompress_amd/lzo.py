"""Mirror of the reference's `Lzo` module (decompress.lzo, lib/lzo.ml): `Lzo.compress` /
`Lzo.uncompress`, run on the GPU through the C ABI (csrc/lzo_kernels.hip)."""
from . import engine as _engine


def max_compressed_length(n):
    """room Lzo.compress never exceeds (liblzo's bound; the reference's tests use (len + 1) * 2)"""
    return n + n // 16 + 64 + 3


def compress(src, device=0):
    """`Lzo.compress in_data out_data wrkmem` (lib/lzo.ml:656-660) -> bytes"""
    st, out = _engine.default_engine(device).lzo_many(True, [src], [max_compressed_length(len(src))])[0]
    if st != 0:
        raise _engine.Error(_engine.STATUS_NAMES[st])
    return out


def uncompress(src, dst_len, device=0):
    """`Lzo.uncompress input output` (lib/lzo.ml:395-403) -> ("Ok", bytes) | ("Error", message)"""
    st, out = _engine.default_engine(device).lzo_many(False, [src], [dst_len])[0]
    return ("Ok", out) if st == 0 else ("Error", _engine.STATUS_NAMES[st])
