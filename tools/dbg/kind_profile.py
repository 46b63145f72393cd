"""in-kernel phase profile (stream 0 of 2048 copies) for representative C2 streams: where the slow kinds lose time"""
import sys, os, zlib
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
nb, n = 262144, 2048
catb = b"".join(v for _, v in cat)
def blocks(z):
    """number of deflate blocks, by inflating with zlib block by block"""
    d = zlib.decompressobj()
    return None
def run(name, plain):
    z = zlib.compress(plain, 6)
    blob, in_off, in_len = workloads.pack([z] * n)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_out = torch.empty(n * nb, dtype=torch.uint8, device=dev)
    args = (decompress_amd.FORMAT_ZLIB, t(blob), t(in_off), t(in_len), d_out, t(np.arange(n, dtype=np.int64) * nb), t(np.full(n, nb, dtype=np.int64)))
    res = eng.inflate_batch(*args)
    torch.cuda.synchronize()
    eng.set_option("profile", 1)
    res = eng.inflate_batch(*args, res)
    p = eng.get_profile()
    eng.set_option("profile", 0)
    cyc = sum(v for k, v in p.items() if k.startswith("cyc_"))
    r = max(1, p["rounds"])
    print("%-8s ratio %.3f total %.2f Mcyc rounds %d (%.0f cyc/round, %.0f B/round) passes/round %.2f lanes/round %.1f slots/pass %.1f near_it/round %.1f" % (
        name, len(z) / nb, cyc / 1e6, r, cyc / r, nb / r, p["passes"] / r, p["lanes"] / r, p["slots"] / max(1, p["passes"]), p["near_iters"] / r))
    print("         rounds cut short by: chain %d  staging-fit %d  records %d  stage-stop %d  eob %d; long near matches %d" % (
        p["end_chain"], p["end_fit"], p["end_records"], p["end_stage"], p["end_eob"], p["long_near"]))
    print("         " + "  ".join("%s %.0f" % (k[4:], v / r) for k, v in p.items() if k.startswith("cyc_")) + "   (cycles per round)", flush=True)
names = sys.argv[1:] or ["book1", "geo", "obj2", "paper2", "pic", "progl"]
for k in names:
    run(k, (catb + catb)[offs[k]:offs[k] + nb])
run("zipf", workloads.text(0xC3, nb))
run("markov", workloads.markov_text(0xC3, nb))
