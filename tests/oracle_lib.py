"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (the checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
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
