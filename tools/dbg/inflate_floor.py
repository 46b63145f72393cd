"""Run ON THE GPU BOX with MD_LIBMDEFLATE=decompress_amd/libmdeflate_kb.so (tools/dbg/build_known_bounds.sh):
where the wall of the one-stream-per-workgroup inflate design is.

  normal        the kernel as shipped (measurement build, mode 0)
  record        mode 1: sync results written to HBM (cost of the record)
  replay        mode 2: sync passes SKIPPED, their results read back - a perfect sync, everything else as it is
  replay+nomatch   mode 2|4: and the copier only flushes (no far, no near matches)
  replay+nocopy    mode 2|8: and the copier does nothing: emit + headers alone
  nomatch / nocopy  modes 4 / 8 with the real sync: the decoder alone

on full C2 (4 096 streams), 2 048 streams (one generation) and 256 streams (one per CU).
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import decompress_amd
from decompress_amd import workloads

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "inflate_floor.json")
eng = decompress_amd.Engine(0)
dev = eng.device
NB = 262144
N = 4096
streams = workloads.c2_streams(N, nbytes=NB)
plains = None


def setup(sub):
    n = len(sub)
    blob, in_off, in_len = workloads.pack(sub)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_out = torch.empty(n * NB, dtype=torch.uint8, device=dev)
    return (decompress_amd.FORMAT_ZLIB, t(blob), t(in_off), t(in_len), d_out, t(np.arange(n, dtype=np.int64) * NB), t(np.full(n, NB, dtype=np.int64)))


def timed(args, reps=5):
    res = eng.inflate_batch(*args)
    torch.cuda.synchronize()
    eng.timing_begin()
    for _ in range(reps):
        res = eng.inflate_batch(*args, res)
    ms = eng.timing_end() / reps
    return ms, int((res[2] != 0).sum().item())


rows = []
for waves in (2, 1):
    eng.set_option("inflate_waves", waves)
    for n in (4096, 2048, 256):
        sub = streams[:: N // n]
        args = setup(sub)
        ref = None
        for name, mode in (("normal", 0), ("record", 1), ("replay", 2), ("replay+nomatch", 2 | 4), ("replay+nocopy", 2 | 8), ("nomatch", 4), ("nocopy", 8), ("normal again", 0)):
            eng.set_option("debug_known_bounds", mode | (n << 5))
            ms, bad = timed(args)
            same = None
            if mode in (0, 1, 2):
                torch.cuda.synchronize()
                cur = args[4].clone()
                if ref is None:
                    ref = cur
                same = bool(torch.equal(cur, ref))
            rows.append({"waves": waves, "streams": n, "mode": name, "ms": round(ms, 4), "bad_status": bad, "same_bytes": same})
            print(rows[-1], flush=True)
        del args, ref
eng.set_option("debug_known_bounds", 0 | (1 << 5))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(rows, f, indent=1)
