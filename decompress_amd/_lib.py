"""ctypes binding of libmdeflate.so (the C ABI of include/mdeflate.h).

There is no fallback: if the HIP library is missing this module raises, and
every compute call goes through the .so.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MD_LIBMDEFLATE: another build of the same library (measurement builds of tools/dbg; there is still no fallback)
SO = os.environ.get("MD_LIBMDEFLATE") or os.path.join(HERE, "libmdeflate.so")

c_sz = ctypes.c_size_t
c_u8p = ctypes.c_void_p
c_vp = ctypes.c_void_p
MD_STREAM_NULL = ctypes.c_void_p(-1).value  # md_create: enqueue on the legacy default stream

class GzHeader(ctypes.Structure):
    """md_gz_header of include/mdeflate.h"""
    _fields_ = [("mtime", ctypes.c_uint32), ("os", ctypes.c_int), ("hcrc", ctypes.c_int), ("ascii", ctypes.c_int),
                ("filename", ctypes.c_char_p), ("comment", ctypes.c_char_p)]


class InfResume(ctypes.Structure):
    """md_inf_resume (mdeflate.h): the last block boundary inside a piece of a stream"""
    _fields_ = [("bits", ctypes.c_uint64), ("out", ctypes.c_uint64), ("adler", ctypes.c_uint32), ("last", ctypes.c_uint32),
                ("consumed", ctypes.c_uint64), ("checksum", ctypes.c_uint32), ("crc_out", ctypes.c_uint32), ("crc_end", ctypes.c_uint32)]


class DeflateParams(ctypes.Structure):
    """md_deflate_params of include/mdeflate.h"""
    _fields_ = [("level", ctypes.c_int), ("queue_len", ctypes.c_int), ("driver", ctypes.c_int), ("dynamic", ctypes.c_int),
                ("matcher", ctypes.c_int), ("gz_header", ctypes.POINTER(GzHeader)), ("wbits", ctypes.c_int),
                ("total_in_bytes", ctypes.c_size_t)]


c_pp = ctypes.POINTER(DeflateParams)
c_szp = ctypes.POINTER(c_sz)

# every symbol include/mdeflate.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("md_version", ctypes.c_int, []),
    ("md_status_string", ctypes.c_char_p, [ctypes.c_int]),
    ("md_shard_plan", ctypes.c_int, [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    ("md_last_error_string", ctypes.c_char_p, [c_vp]),
    ("md_device_count", ctypes.c_int, []),
    ("md_create", c_vp, [ctypes.c_int, c_vp]),
    ("md_destroy", None, [c_vp]),
    ("md_synchronize", ctypes.c_int, [c_vp]),
    ("md_set_option", ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    ("md_def_batch_open", c_vp, [c_vp, ctypes.c_int, c_pp, c_sz]),
    ("md_def_batch_src", ctypes.c_int, [c_vp, c_sz, c_vp, c_sz]),
    ("md_def_batch_encode", ctypes.c_int, [c_vp]),
    ("md_def_batch_pending", c_sz, [c_vp, c_sz]),
    ("md_def_batch_out", c_sz, [c_vp, c_sz, c_vp, c_sz]),
    ("md_def_batch_status", ctypes.c_int, [c_vp, c_sz]),
    ("md_def_batch_error", ctypes.c_int, [c_vp, c_sz]),
    ("md_def_batch_checksum", ctypes.c_uint32, [c_vp, c_sz]),
    ("md_def_batch_close", None, [c_vp]),
    ("md_host_alloc", c_vp, [c_vp, c_sz]),
    ("md_host_free", None, [c_vp, c_vp]),
    ("md_timing_begin", ctypes.c_int, [c_vp]),
    ("md_timing_end", ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_float)]),
    ("md_inflate_batch_device", ctypes.c_int,
     [c_vp, ctypes.c_int, c_sz] + [c_vp] * 10),
    ("md_inflate_batch_host", ctypes.c_int,
     [c_vp, ctypes.c_int, c_sz, c_vp, c_sz, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    ("md_deflate_batch_device", ctypes.c_int, [c_vp, ctypes.c_int, c_pp, c_sz] + [c_vp] * 9),
    ("md_deflate_batch_host", ctypes.c_int,
     [c_vp, ctypes.c_int, c_pp, c_sz, c_vp, c_sz, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp]),
    ("md_de_higher_uncompress", ctypes.c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_szp]),
    ("md_zl_higher_uncompress", ctypes.c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_szp]),
    ("md_de_lz77_compress", ctypes.c_int,
     [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, c_szp, c_vp, c_vp]),
    ("md_de_def_encode", ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, c_szp]),
    ("md_de_def_run", ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, c_szp, c_vp, c_sz, c_szp]),
    ("md_de_def_ns_deflate", ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, c_szp]),
    ("md_zl_def_ns_deflate", ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, c_szp]),
    ("md_de_def_ns_compress_bound", c_sz, [c_sz]),
    ("md_zl_def_ns_compress_bound", c_sz, [c_sz]),
    ("md_def_ns_batch_device", ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_sz, c_sz] + [c_vp] * 9),
    ("md_inf_decoder", c_vp, [c_vp, ctypes.c_int, c_vp, c_sz]),
    ("md_inf_src", ctypes.c_int, [c_vp, c_vp, c_sz, c_sz]),
    ("md_inf_decode", ctypes.c_int, [c_vp]),
    ("md_inf_flush", None, [c_vp]),
    ("md_inf_dst_rem", c_sz, [c_vp]),
    ("md_inf_src_rem", c_sz, [c_vp]),
    ("md_inf_status", ctypes.c_int, [c_vp]),
    ("md_inf_checksum", ctypes.c_uint32, [c_vp]),
    ("md_inf_free", None, [c_vp]),
    ("md_inf_reset", None, [c_vp]),
    ("md_inf_chunk_bytes", None, [c_vp, c_sz]),
    ("md_inflate_continue_batch_device", ctypes.c_int, [c_vp, c_sz] + [c_vp] * 17),
    ("md_de_inf_continue_host", ctypes.c_int, [c_vp, c_vp, c_sz, ctypes.c_uint, c_vp, c_sz, c_sz, ctypes.c_uint32, ctypes.c_uint, c_vp, c_vp, c_vp]),
    ("md_inf_message", ctypes.c_char_p, [c_vp]),
    ("md_def_encoder", c_vp, [c_vp, ctypes.c_int, c_pp, c_vp, c_sz]),
    ("md_def_src", ctypes.c_int, [c_vp, c_vp, c_sz, c_sz]),
    ("md_def_encode", ctypes.c_int, [c_vp]),
    ("md_def_dst", None, [c_vp, c_vp, c_sz]),
    ("md_def_dst_rem", c_sz, [c_vp]),
    ("md_def_status", ctypes.c_int, [c_vp]),
    ("md_def_checksum", ctypes.c_uint32, [c_vp]),
    ("md_def_free", None, [c_vp]),
    ("md_de_higher_compress", ctypes.c_int,
     [c_vp, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz)]),
    ("md_zl_higher_compress", ctypes.c_int,
     [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz)]),
    ("md_de_inf_ns_inflate", ctypes.c_int,
     [c_vp, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz)]),
    ("md_zl_inf_ns_inflate", ctypes.c_int,
     [c_vp, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz)]),
    ("md_gz_higher_compress", ctypes.c_int,
     [c_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(GzHeader), c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz)]),
    ("md_gz_higher_uncompress", ctypes.c_int,
     [c_vp, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz), c_vp]),
    ("md_crc32_batch_device", ctypes.c_int, [c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    ("md_lzo_uncompress_batch_device", ctypes.c_int, [c_vp, c_sz] + [c_vp] * 8),
    ("md_lzo_compress_batch_device", ctypes.c_int, [c_vp, c_sz] + [c_vp] * 8),
    ("md_lzo_uncompress", ctypes.c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz)]),
    ("md_lzo_compress", ctypes.c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, ctypes.POINTER(c_sz)]),
]


class GzMeta(ctypes.Structure):
    """md_gz_meta of include/mdeflate.h"""
    _fields_ = [(k, ctypes.c_uint32) for k in ("flg", "mtime", "xfl", "os")] + \
               [(k, ctypes.c_int) for k in ("has_extra", "has_name", "has_comment")] + \
               [(k, c_sz) for k in ("extra_off", "extra_len", "name_off", "name_len", "comment_off", "comment_len")]
# exported but not part of the public header (tuning knobs)
EXTRA = [("md_get_profile", ctypes.c_int, [c_vp, c_vp])]

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError(
                "decompress_amd: %s is missing — build it with "
                "`python -m decompress_amd.build` (there is no CPU fallback)" % SO)
        lib = ctypes.CDLL(SO)
        for name, res, args in SYMBOLS + EXTRA:
            fn = getattr(lib, name)  # AttributeError = ABI drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
