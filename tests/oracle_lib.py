"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (the checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import contextlib
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

STATUS = {
    0: "Ok", 1: "Unexpected_end_of_input", 2: "Unexpected_end_of_output",
    3: "Invalid_kind_of_block", 4: "Invalid_dictionary",
    5: "Invalid_complement_of_length", 6: "Invalid_distance",
    7: "Invalid_distance_code", 8: "Invalid_header", 9: "Invalid_checksum",
}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
        _LIB = Oracle(ctypes.CDLL(so))
    return _LIB


class MiniLzo:
    """oracle/_ref/libminilzo.so — the reference's own LZO oracle (test/minilzo-2.10), built from the
    reference tree by oracle/Makefile.  TEST INFRASTRUCTURE ONLY."""

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        sz = ctypes.c_size_t
        for fn in ("lzo1x_1_compress", "lzo1x_decompress_safe"):
            f = getattr(self.lib, fn)
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, ctypes.POINTER(sz), ctypes.c_void_p]

    def compress(self, data):
        dst = ctypes.create_string_buffer(len(data) + len(data) // 16 + 64 + 3)
        n = ctypes.c_size_t(len(dst))
        wrk = ctypes.create_string_buffer(16384 * 8)
        assert self.lib.lzo1x_1_compress(bytes(data), len(data), dst, ctypes.byref(n), wrk) == 0
        return dst.raw[: n.value]

    def decompress(self, data, cap):
        dst = ctypes.create_string_buffer(max(cap, 1))
        n = ctypes.c_size_t(cap)
        rc = self.lib.lzo1x_decompress_safe(bytes(data), len(data), dst, ctypes.byref(n), None)
        return rc, dst.raw[: n.value]


def load_minilzo():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libminilzo.so")
    return MiniLzo(path) if os.path.exists(path) else None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        sz = ctypes.c_size_t
        for fn in ("orc_de_inf_ns_inflate", "orc_zl_inf_ns_inflate"):
            f = getattr(lib, fn)
            f.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, sz, ctypes.POINTER(sz), ctypes.POINTER(sz)]
            f.restype = ctypes.c_int
        lib.orc_adler32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, sz]
        lib.orc_adler32.restype = ctypes.c_uint32
        lib.orc_crc32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, sz]
        lib.orc_crc32.restype = ctypes.c_uint32
        ci = ctypes.c_int
        lib.orc_deflate_raw.restype = ctypes.c_void_p
        lib.orc_deflate_raw.argtypes = [ctypes.c_char_p, sz, ci, ci, ci, ci, ctypes.POINTER(sz),
                                        ctypes.POINTER(ctypes.c_uint32)]
        lib.orc_deflate_raw_m.restype = ctypes.c_void_p
        lib.orc_deflate_raw_m.argtypes = [ctypes.c_char_p, sz, ci, ci, ci, ci, ci, ctypes.POINTER(sz),
                                          ctypes.POINTER(ctypes.c_uint32)]
        lib.orc_zl_deflate.restype = ctypes.c_void_p
        lib.orc_zl_deflate.argtypes = [ctypes.c_char_p, sz, ci, ci, ci, ctypes.POINTER(sz)]
        lib.orc_free.argtypes = [ctypes.c_void_p]
        lib.orc_tree_make.argtypes = [ci, ci, ctypes.POINTER(ci), ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        lib.orc_tree_make.restype = ci
        lib.orc_encode_cmds.restype = ctypes.c_void_p
        lib.orc_encode_cmds.argtypes = [ctypes.POINTER(ci), ci, ci, ctypes.POINTER(sz)]
        lib.orc_lz77_cmds.restype = ci
        lib.orc_lz77_cmds.argtypes = [ctypes.c_char_p, sz, ci, ci, ctypes.POINTER(ci), ci]

        class GzMeta(ctypes.Structure):
            _fields_ = [(k, ctypes.c_uint32) for k in ("cm", "flg", "mtime", "xfl", "os")] + \
                       [(k, ctypes.c_int) for k in ("has_extra", "has_name", "has_comment")] + \
                       [(k, ctypes.c_size_t) for k in ("extra_off", "extra_len", "name_off", "name_len",
                                                       "comment_off", "comment_len")]
        self.GzMeta = GzMeta
        lib.orc_gz_inflate.restype = ci
        lib.orc_gz_inflate.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, sz, ctypes.POINTER(sz),
                                       ctypes.POINTER(sz), ctypes.POINTER(GzMeta)]
        lib.orc_gz_deflate.restype = ctypes.c_void_p
        lib.orc_gz_deflate.argtypes = [ctypes.c_char_p, sz, ci, ci, ctypes.c_uint32, ci, ci, ci, ctypes.c_char_p,
                                       ctypes.c_char_p, ctypes.POINTER(sz)]
        lib.orc_status_string.restype = ctypes.c_char_p
        lib.orc_status_string.argtypes = [ci]

    def lzo_uncompress(self, src, cap):
        """Lzo.uncompress -> (status, bytes)"""
        self.lib.orc_lzo_uncompress.restype = ctypes.c_int
        self.lib.orc_lzo_uncompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                                ctypes.POINTER(ctypes.c_size_t)]
        dst = ctypes.create_string_buffer(max(cap, 1))
        w = ctypes.c_size_t()
        rc = self.lib.orc_lzo_uncompress(bytes(src), len(src), dst, cap, ctypes.byref(w))
        return rc, dst.raw[: w.value]

    def lzo_compress(self, src, cap=None):
        """Lzo.compress -> (status, bytes)"""
        self.lib.orc_lzo_compress.restype = ctypes.c_int
        self.lib.orc_lzo_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                              ctypes.POINTER(ctypes.c_size_t)]
        cap = len(src) + len(src) // 16 + 64 + 3 if cap is None else cap
        dst = ctypes.create_string_buffer(max(cap, 1))
        w = ctypes.c_size_t()
        rc = self.lib.orc_lzo_compress(bytes(src), len(src), dst, cap, ctypes.byref(w))
        return rc, dst.raw[: w.value]

    def gz_inflate(self, src, cap):
        """Gz.Inf over a whole buffer -> (status, consumed, bytes, meta dict)."""
        src = bytes(src)
        dst = ctypes.create_string_buffer(max(cap, 1))
        c, w = ctypes.c_size_t(), ctypes.c_size_t()
        m = self.GzMeta()
        rc = self.lib.orc_gz_inflate(src, len(src), dst, cap, ctypes.byref(c), ctypes.byref(w), ctypes.byref(m))
        meta = {"flg": m.flg, "os": m.os, "mtime": m.mtime, "xfl": m.xfl,
                "name": src[m.name_off:m.name_off + m.name_len] if m.has_name else None,
                "comment": src[m.comment_off:m.comment_off + m.comment_len] if m.has_comment else None,
                "extra": src[m.extra_off:m.extra_off + m.extra_len] if m.has_extra else None}
        return rc, c.value, dst.raw[: w.value], meta

    def gz_deflate(self, src, level=4, queue=4096, mtime=0, os=3, hcrc=False, ascii=False, name=None, comment=None):
        n = ctypes.c_size_t()
        p = self.lib.orc_gz_deflate(bytes(src), len(src), level, queue, mtime, os, int(hcrc), int(ascii),
                                    name, comment, ctypes.byref(n))
        out = ctypes.string_at(p, n.value)
        self.lib.orc_free(p)
        return out

    def status_string(self, st):
        return self.lib.orc_status_string(st).decode()

    def _inflate(self, fn, src, cap):
        dst = ctypes.create_string_buffer(max(cap, 1))
        c, w = ctypes.c_size_t(), ctypes.c_size_t()
        rc = fn(bytes(src), len(src), dst, cap, ctypes.byref(c), ctypes.byref(w))
        return rc, c.value, dst.raw[: w.value]

    def de_inflate(self, src, cap=65536):
        """De.Inf.Ns.inflate -> (status, consumed, output bytes)"""
        return self._inflate(self.lib.orc_de_inf_ns_inflate, src, cap)

    def zl_inflate(self, src, cap=65536):
        """Zl.Inf.Ns.inflate -> (status, consumed, output bytes)"""
        return self._inflate(self.lib.orc_zl_inf_ns_inflate, src, cap)

    def adler32(self, data, init=1):
        return self.lib.orc_adler32(init, bytes(data), len(data))

    def crc32(self, data, init=0):
        return self.lib.orc_crc32(init, bytes(data), len(data))

    # ------------------------------------------------------------------ deflate
    DRV_ZL, DRV_HIGHER, DRV_CLI = 0, 1, 2

    def _take(self, p, n):
        out = ctypes.string_at(p, n)
        self.lib.orc_free(p)
        return out

    @contextlib.contextmanager
    def src_piece(self, piece):
        """Inside the block the deflate drivers hand the input to De.Lz77 `piece` bytes per `Await (oracle.h)."""
        self.lib.orc_set_src_piece.argtypes = [ctypes.c_size_t]
        self.lib.orc_set_src_piece.restype = None
        self.lib.orc_set_src_piece(piece)
        try:
            yield
        finally:
            self.lib.orc_set_src_piece(0)

    @contextlib.contextmanager
    def src_pieces(self, first, piece):
        """as src_piece, with a first piece of `first` bytes"""
        self.lib.orc_set_src_first_piece.argtypes = [ctypes.c_size_t]
        self.lib.orc_set_src_first_piece.restype = None
        self.lib.orc_set_src_first_piece(first)
        try:
            with self.src_piece(piece):
                yield
        finally:
            self.lib.orc_set_src_first_piece(0)

    def deflate_raw(self, data, level=6, queue=4096, driver=0, dynamic=True, matcher=0):
        """De.Lz77 (matcher 0) or lib/lz.ml's Lz (matcher 1) + De.Def under one of the reference's
        drivers -> (raw DEFLATE, adler32 of input)"""
        n, a = ctypes.c_size_t(), ctypes.c_uint32()
        p = self.lib.orc_deflate_raw_m(bytes(data), len(data), level, queue, driver, int(dynamic), matcher,
                                       ctypes.byref(n), ctypes.byref(a))
        if not p:
            return None, a.value  # the reference would raise De.Queue.Full
        return self._take(p, n.value), a.value

    def zl_deflate(self, data, level=6, queue=4096, dynamic=True):
        """Zl.Higher.compress ~level ~dynamic (lib/zl.ml:634-648)"""
        n = ctypes.c_size_t()
        p = self.lib.orc_zl_deflate(bytes(data), len(data), level, queue, int(dynamic), ctypes.byref(n))
        assert p
        return self._take(p, n.value)

    def tree_make(self, length, freqs, max_length=15):
        f = (ctypes.c_int * len(freqs))(*freqs)
        lens = (ctypes.c_int * length)()
        codes = (ctypes.c_int * length)()
        mc = self.lib.orc_tree_make(length, max_length, f, len(freqs), lens, codes)
        return mc, list(lens), list(codes), list(f)

    def encode_cmds(self, cmds, kind):
        arr = (ctypes.c_int * len(cmds))(*cmds)
        n = ctypes.c_size_t()
        p = self.lib.orc_encode_cmds(arr, len(cmds), {"flat": 0, "fixed": 1, "dynamic": 2}[kind], ctypes.byref(n))
        return self._take(p, n.value)

    def def_ns(self, data, level=4, cap=None, zl=False):
        """De.Def.Ns.deflate (Zl.Def.Ns.deflate with zl=True) -> (status, bytes); status -1 = Invalid_compression_level"""
        data = bytes(data)
        bound = self.lib.orc_de_def_ns_compress_bound
        bound.restype = ctypes.c_size_t
        bound.argtypes = [ctypes.c_size_t]
        if cap is None:
            cap = bound(len(data)) + (6 if zl else 0)
        dst, n = ctypes.create_string_buffer(max(1, cap)), ctypes.c_size_t()
        fn = self.lib.orc_zl_def_ns_deflate if zl else self.lib.orc_de_def_ns_deflate
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
        st = fn(data, len(data), dst, cap, level, ctypes.byref(n))
        return st, dst.raw[:n.value]

    def def_script(self, ops, queue=4096):
        """De.Def.encode driven by a list of operations -> (bytes, [0 `Ok | 1 `Block, ...]) or None (Queue.Full / bad list)"""
        arr = (ctypes.c_int * max(1, len(ops)))(*[o if o < 2**31 else o - 2**32 for o in ops])
        rcs, nrc, n = (ctypes.c_int * 64)(), ctypes.c_int(), ctypes.c_size_t()
        self.lib.orc_def_script.restype = ctypes.c_void_p
        self.lib.orc_def_script.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                            ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t)]
        p = self.lib.orc_def_script(arr, len(ops), queue, rcs, 64, ctypes.byref(nrc), ctypes.byref(n))
        if not p:
            return None
        return self._take(p, n.value), list(rcs[:nrc.value])

    def lz77_all(self, data, level=4, queue=4096, matcher=0):
        """every queue fill of De.Lz77.compress (or Lz.compress) in order + literals[286] + distances[30]"""
        cap = len(data) + len(data) // max(1, queue - 1) + 8
        out, lits, dsts = (ctypes.c_int * cap)(), (ctypes.c_int * 286)(), (ctypes.c_int * 30)()
        self.lib.orc_lz77_cmds_ex.restype = ctypes.c_int
        n = self.lib.orc_lz77_cmds_ex(bytes(data), ctypes.c_size_t(len(data)), level, queue, matcher, out, cap, lits, dsts)
        assert n <= cap
        return [c & 0xffffffff for c in out[:n]], list(lits), list(dsts)

    def lz77_cmds(self, data, level=4, queue=4096):
        out = (ctypes.c_int * queue)()
        n = self.lib.orc_lz77_cmds(bytes(data), len(data), level, queue, out, queue)
        return None if n < 0 else list(out[:n])
