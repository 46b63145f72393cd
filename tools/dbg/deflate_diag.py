import sys, os, zlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
from tests import oracle_lib
n, nb = int(sys.argv[1]), int(sys.argv[2]) * 1024
eng = decompress_amd.Engine(0)
bufs = [workloads.ascii_uniform(0xC3 + i, nb) for i in range(n)]
outs = eng.deflate_many(bufs, level=6, fmt=decompress_amd.FORMAT_ZLIB) if hasattr(eng, "deflate_many") else None
if len(sys.argv) > 3:
    for _ in range(int(sys.argv[3])):
        outs = eng.deflate_many(bufs, level=6, fmt=decompress_amd.FORMAT_ZLIB)
orc = oracle_lib.load()
bad = 0
for k, (o, b) in enumerate(zip(outs, bufs)):
    st, o = o[0], o[1]
    try:
        rt = zlib.decompress(o) == b
    except Exception as e:
        rt = repr(e)
    if rt is not True:
        bad += 1
        if bad <= 5:
            z = orc.zl_deflate(b, 6)
            m = next((i for i in range(min(len(o), len(z))) if o[i] != z[i]), None)
            print("status", st, "stream", k, "roundtrip", rt, "len", len(o), len(z), "first mismatch", m)
print("bad", bad, "of", n)
