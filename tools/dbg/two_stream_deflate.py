import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
n, parts, nb = int(sys.argv[2]), int(sys.argv[1]), int(sys.argv[3]) * 1024
dev = torch.device("cuda", 0)
uniq = [workloads.ascii_uniform(0xC3 + i, nb) for i in range(64)]
per = n // parts
engs, args = [], []
for k in range(parts):
    e = decompress_amd.Engine(0)
    sub = [uniq[i % 64] for i in range(per)]
    blob, off, ln = workloads.pack(sub)
    cap = np.full(per, 2 * nb + 8192, dtype=np.int64)
    ooff = np.arange(per, dtype=np.int64) * (2 * nb + 8192)
    t = lambda a: torch.from_numpy(a).to(dev)
    a = (t(blob), t(off), t(ln), torch.empty(int(cap.sum()), dtype=torch.uint8, device=dev), t(ooff), t(cap))
    engs.append(e); args.append(a)
res = [None] * parts
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(parts):
        res[k] = engs[k].deflate_batch(decompress_amd.FORMAT_ZLIB, *args[k], level=6, results=res[k])
    for e in engs:
        e.synchronize()
    dt = time.perf_counter() - t0
print("parts", parts, "n", n, "ms %.2f" % (dt * 1e3), "ok", all(bool((r[1] == 0).all().item()) for r in res))
