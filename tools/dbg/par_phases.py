"""Run ON THE GPU BOX with MD_DEBUG_HOSTPATH=1: the phases of the whole-chip path for one long stream (capi.cpp par_decode
prints them): body on the device, candidates found, pieces decoded, windows + resolve."""
import ctypes, sys, time, zlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
data = workloads.text(77, mib << 20)
for name, z in (("level0", zlib.compress(data, 0)), ("level6", zlib.compress(data, 6))):
    dst = ctypes.create_string_buffer(len(data))
    w = ctypes.c_size_t()
    for _ in range(2):
        print(name, file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        st = eng.lib.md_zl_higher_uncompress(eng.ctx, z, len(z), dst, len(data), ctypes.byref(w))
        print("  call: %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    assert st == 0 and dst.raw == data
