"""Per corpus file: the deflate kernels' time (run under rocprofv3 --kernel-trace; see corpus_kinds.sh) on 512 copies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
eng.set_option("deflate_workspace_cap_mib", 0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, data in workloads.corpus().items():
    bufs = [data] * 512
    res = eng.deflate_many(bufs, decompress_amd.FORMAT_GZIP, level=level)
    print("FILE %s %d bytes ratio %.3f" % (name, len(data), len(res[0][1]) / max(1, len(data))), flush=True)
