"""Mirror of the reference's `Gz` module surface for the hot path (lib/gz.ml: `Gz.Inf`,
`Gz.Def`, `Gz.Higher`), executed on the GPU through the C ABI: header / trailer handling in
`csrc/gz_kernels.hip`, the DEFLATE body in the batched inflate / deflate kernels."""
import ctypes

from . import engine as _engine
from ._lib import GzMeta

# Gz.os, lib/gz.ml:158-246 (RFC1952 numbering)
OS = {"FAT": 0, "Amiga": 1, "VMS": 2, "Unix": 3, "VM": 4, "Atari": 5, "HPFS": 6, "Macintosh": 7, "Z": 8, "CPM": 9,
      "TOPS20": 10, "NTFS": 11, "QDOS": 12, "Acorn": 13, "Unknown": 255}


def extra(payload, key):
    """Gz.Inf.extra ~key (lib/gz.ml:617-633): 2-character subfield id, uint16_be length, value."""
    if len(key) != 2:
        raise ValueError("Subfield ID must be 2 characters.")
    idx = 0
    while payload is not None and idx + 4 <= len(payload):
        k, ln = payload[idx:idx + 2], int.from_bytes(payload[idx + 2:idx + 4], "big")
        if idx + 4 + ln > len(payload):
            break
        if k == key:
            return payload[idx + 4:idx + 4 + ln]
        idx += 4 + ln
    return None


class Higher:
    """Gz.Higher (lib/gz.ml:921-982)."""

    @staticmethod
    def compress(src, level=0, filename=None, comment=None, mtime=0, os="Unix", ascii=False, hcrc=False,
                 queue=4096, device=0):
        """`Gz.Higher.compress ?level ?filename ?comment ~w ~q ~refill ~flush time cfg i o`
        (`?level` defaults to 0 upstream, lib/gz.ml:928; `cfg` = ascii / hcrc / os / mtime)."""
        eng = _engine.default_engine(device)
        hdr = dict(mtime=int(mtime) & 0xffffffff, os=OS[os] if isinstance(os, str) else int(os), hcrc=int(bool(hcrc)),
                   ascii=int(bool(ascii)), filename=filename, comment=comment)
        st, out, _ = eng.deflate_many([src], _engine.FORMAT_GZIP, level=level, queue=queue, header=hdr)[0]
        if st != 0:
            raise _engine.Error(_engine.STATUS_NAMES[st])
        return out

    @staticmethod
    def uncompress(src, dst_len, device=0):
        """`Gz.Higher.uncompress ~refill ~flush i o` -> ("Ok", metadata, bytes) | ("Error", msg)."""
        eng = _engine.default_engine(device)
        src = bytes(src)
        dst = ctypes.create_string_buffer(max(dst_len, 1))
        used, wrote, m = ctypes.c_size_t(), ctypes.c_size_t(), GzMeta()
        st = eng.lib.md_gz_higher_uncompress(eng.ctx, src, len(src), dst, dst_len, ctypes.byref(used),
                                             ctypes.byref(wrote), ctypes.byref(m))
        if st < 0:
            eng._check(st)
        if st != 0:
            return "Error", _engine.STATUS_NAMES[st]
        meta = {
            "filename": src[m.name_off:m.name_off + m.name_len] if m.has_name else None,
            "comment": src[m.comment_off:m.comment_off + m.comment_len] if m.has_comment else None,
            "os": m.os, "mtime": m.mtime,
            "extra": src[m.extra_off:m.extra_off + m.extra_len] if m.has_extra else None,
        }
        return "Ok", meta, dst.raw[:wrote.value]


class Inf:
    @staticmethod
    def inflate_batch(srcs, dst_lens, device=0):
        """n GZip members at once -> [(status, consumed, bytes, crc32)]"""
        return _engine.default_engine(device).inflate_many(srcs, dst_lens, _engine.FORMAT_GZIP)


class Def:
    @staticmethod
    def deflate_batch(bufs, level=4, queue=4096, device=0, **header):
        """n buffers at once -> [(status, gzip bytes, crc32 of the input)]"""
        hdr = dict(mtime=0, os=3, hcrc=0, ascii=0, filename=None, comment=None)
        hdr.update(header)
        hdr["os"] = OS[hdr["os"]] if isinstance(hdr["os"], str) else int(hdr["os"])
        return _engine.default_engine(device).deflate_many(bufs, _engine.FORMAT_GZIP, level=level, queue=queue, header=hdr)
