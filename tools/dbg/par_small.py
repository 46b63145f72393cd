"""Run ON THE GPU BOX: where does the whole-chip path start to pay?  One zlib stream of n KiB of text, with the path
(inflate_parallel_min = 16 KiB) and without."""
import ctypes, sys, time, zlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
for kib in (128, 256, 512, 1024, 2048, 4096):
    data = workloads.text(kib, kib << 10)
    z = zlib.compress(data, 6)
    dst = ctypes.create_string_buffer(len(data))
    w = ctypes.c_size_t()
    row = []
    for par in (16, 0):
        eng.set_option("inflate_parallel_min", par)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            st = eng.lib.md_zl_higher_uncompress(eng.ctx, z, len(z), dst, len(data), ctypes.byref(w))
            best = min(best, time.perf_counter() - t0)
        assert st == 0 and dst.raw == data
        v = eng.lib.md_set_option(eng.ctx, b"inflate_parallel_last", 0)
        row.append("%.2f ms (%d pieces)" % (best * 1e3, v & 0xffffff))
    print("%5d KiB of text, %7d compressed: pieces %s | serial %s" % (kib, len(z), row[0], row[1]), flush=True)
