import sys, os, zlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
from tests import oracle_lib
ids = [int(x) for x in sys.argv[2:]]
nb = int(sys.argv[1]) * 1024
eng = decompress_amd.Engine(0)
orc = oracle_lib.load()
bufs = [workloads.ascii_uniform(0xC3 + i, nb) for i in ids]
outs = eng.deflate_many(bufs, level=6, fmt=decompress_amd.FORMAT_DEFLATE)
for i, (st, o, _), b in zip(ids, outs, bufs):
    z = orc.deflate_raw(b, 6)[0] if hasattr(orc, "deflate_raw") else None
    m = next((j for j in range(min(len(o), len(z))) if o[j] != z[j]), None)
    print("stream", i, "status", st, "len", len(o), len(z), "first mismatch", m, "rt", zlib.decompress(o, -15) == b)
    if m is not None:
        # token-level diff: parse both with a tiny inflate that logs (pos, kind, len, dist)
        from tools.dbg.tok import tokens
        ta, tb = tokens(o), tokens(z)
        for k, (x, y) in enumerate(zip(ta, tb)):
            if x != y:
                print(" first differing token", k, "gpu", x, "oracle", y, "prev", ta[max(0,k-3):k])
                break
