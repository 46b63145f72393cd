"""time a batch of 2048 copies of one stream (one resident generation) for representative C2 streams"""
import sys, os, zlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
dev = eng.device
cat = list(workloads.corpus().items())
offs, o = {}, 0
for k, v in cat:
    offs[k] = o; o += len(v)
total = o
nb = 262144
n = 2048
def run(name, plain):
    z = zlib.compress(plain, 6)
    blob, in_off, in_len = workloads.pack([z] * n)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_out = torch.empty(n * nb, dtype=torch.uint8, device=dev)
    args = (decompress_amd.FORMAT_ZLIB, t(blob), t(in_off), t(in_len), d_out, t(np.arange(n, dtype=np.int64) * nb), t(np.full(n, nb, dtype=np.int64)))
    res = eng.inflate_batch(*args)
    torch.cuda.synchronize()
    eng.timing_begin()
    for _ in range(3): res = eng.inflate_batch(*args, res)
    ms = eng.timing_end() / 3
    ok = bool((res[2] == 0).all().item())
    print("%-10s ratio %.3f  %.2f ms  ok=%s" % (name, len(z) / nb, ms, ok), flush=True)
catb = b"".join(v for _, v in cat)
for k, v in cat:
    p = (catb + catb)[offs[k]:offs[k] + nb]
    run(k, p)
run("zipf", workloads.text(0xC3, nb))
run("markov", workloads.markov_text(0xC3, nb))
