import ctypes, sys, zlib
sys.path.insert(0, "/root/repo")
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
data = workloads.text(77, 8 << 20)
for name, z in (("level0", zlib.compress(data, 0)), ("level6", zlib.compress(data, 6))):
    dst = ctypes.create_string_buffer(len(data))
    w = ctypes.c_size_t()
    for _ in range(2):
        print(name, file=sys.stderr, flush=True)
        st = eng.lib.md_zl_higher_uncompress(eng.ctx, z, len(z), dst, len(data), ctypes.byref(w))
    assert st == 0 and dst.raw == data
