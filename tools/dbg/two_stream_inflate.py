import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
n, parts = 4096, int(sys.argv[1])
dev = torch.device("cuda", 0)
streams = workloads.c2_streams(256, nbytes=256 * 1024, workers=0)
streams = [streams[i % 256] for i in range(n)]
per = n // parts
engs, args = [], []
for k in range(parts):
    e = decompress_amd.Engine(0)
    e.set_option('overlap', int(sys.argv[2]) if len(sys.argv) > 2 else 2)
    sub = streams[k * per:(k + 1) * per]
    blob, off, ln = workloads.pack(sub)
    cap = np.full(per, 256 * 1024, dtype=np.int64)
    ooff = np.arange(per, dtype=np.int64) * (256 * 1024)
    t = lambda a: torch.from_numpy(a).to(dev)
    a = (t(blob), t(off), t(ln), torch.empty(per * 256 * 1024, dtype=torch.uint8, device=dev), t(ooff), t(cap))
    engs.append(e); args.append(a)
res = [None] * parts
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(parts):
        res[k] = engs[k].inflate_batch(decompress_amd.FORMAT_ZLIB, *args[k], results=res[k])
    for e in engs:
        e.synchronize()
    dt = time.perf_counter() - t0
print("parts", parts, "ms %.3f" % (dt * 1e3), "ok", all(bool((r[2] == 0).all().item()) for r in res))
