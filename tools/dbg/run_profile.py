import sys, os, zlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
dev = eng.device
nb, n = 1 << 20, 64
for name, plain in (("zeros", bytes(nb)), ("ab-run", (b"ab" * 70 + b"c") * (nb // 141 + 1))):
    plain = plain[:nb]
    z = zlib.compress(plain, 6)
    blob, in_off, in_len = workloads.pack([z] * n)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_out = torch.empty(n * nb, dtype=torch.uint8, device=dev)
    args = (decompress_amd.FORMAT_ZLIB, t(blob), t(in_off), t(in_len), d_out, t(np.arange(n, dtype=np.int64) * nb), t(np.full(n, nb, dtype=np.int64)))
    res = eng.inflate_batch(*args); torch.cuda.synchronize()
    eng.set_option("profile", 1)
    res = eng.inflate_batch(*args, res)
    p = eng.get_profile(); eng.set_option("profile", 0)
    cyc = sum(v for k, v in p.items() if k.startswith("cyc_")); r = max(1, p["rounds"])
    print(name, "ratio %.4f total %.1f Mcyc rounds %d cyc/round %.0f B/round %.0f passes/round %.1f lanes/round %.1f near_it/round %.1f" % (len(z)/nb, cyc/1e6, r, cyc/r, nb/r, p["passes"]/r, p["lanes"]/r, p["near_iters"]/r))
    print("   ", "  ".join("%s %.0f" % (k[4:], v / r) for k, v in p.items() if k.startswith("cyc_")))
    print("    ends: chain %d fit %d records %d stage %d eob %d" % (p["end_chain"], p["end_fit"], p["end_records"], p["end_stage"], p["end_eob"]))
