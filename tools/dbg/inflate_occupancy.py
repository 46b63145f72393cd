"""Run ON THE GPU BOX: the occupancy curve of the inflate kernel.

  (A) kernel time against the number of C2 streams in the batch (256 .. 4096: one stream per CU up to two generations);
  (B) 2 048 C2 streams (one generation at 8 per CU) with the LDS of a stream padded so that a CU holds 1, 2, 4, 6, 8 at
      once (md_set_option "debug_inflate_lds_pad"), both kernel forms.

Writes a table (JSON + text) to the path given as argv[1] (default gpurun_out/inflate_occupancy.json).
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import decompress_amd
from decompress_amd import workloads

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "inflate_occupancy.json")
eng = decompress_amd.Engine(0)
dev = eng.device
NB = 262144
N = 4096
streams = workloads.c2_streams(N, nbytes=NB)
SMEM = int(os.environ.get("MD_SMEM_BYTES", "20432"))
LDS = 160 * 1024


def run(sub, waves, pad, reps=5):
    eng.set_option("inflate_waves", waves)
    eng.set_option("debug_inflate_lds_pad", pad)
    n = len(sub)
    blob, in_off, in_len = workloads.pack(sub)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_out = torch.empty(n * NB, dtype=torch.uint8, device=dev)
    args = (decompress_amd.FORMAT_ZLIB, t(blob), t(in_off), t(in_len), d_out, t(np.arange(n, dtype=np.int64) * NB), t(np.full(n, NB, dtype=np.int64)))
    res = eng.inflate_batch(*args)
    torch.cuda.synchronize()
    eng.timing_begin()
    for _ in range(reps):
        res = eng.inflate_batch(*args, res)
    ms = eng.timing_end() / reps
    ok = bool((res[2] == 0).all().item()) and bool((res[0] == NB).all().item())
    return ms, ok


rows = []
for waves in (2, 1):
    for n in (256, 512, 1024, 2048, 3072, 4096):
        # every 4096 / n-th stream: the same mix of kinds at every size
        sub = streams[:: N // n] if N % n == 0 else streams[:n]
        ms, ok = run(sub, waves, 0)
        rows.append({"exp": "streams", "waves": waves, "streams": n, "per_cu_limit": 8 * (2 // waves) if False else 8, "ms": round(ms, 4), "ok": ok})
        print(rows[-1], flush=True)
for waves in (2, 1):
    for k in (1, 2, 3, 4, 5, 6, 7, 8):
        # the largest LDS footprint that still lets k streams share a CU, minus a little
        per = LDS // k
        pad = max(0, per - SMEM - 256) if k < 8 else 0
        if pad + SMEM > 160 * 1024 - 1024:
            pad = 160 * 1024 - 1024 - SMEM
        sub = streams[::2]
        ms, ok = run(sub, waves, pad)
        rows.append({"exp": "per_cu", "waves": waves, "streams": 2048, "streams_per_cu": k, "pad": pad, "ms": round(ms, 4),
                     "generations": 8.0 / k, "ms_per_generation": round(ms * k / 8.0, 4), "ok": ok})
        print(rows[-1], flush=True)
eng.set_option("debug_inflate_lds_pad", 0)
eng.set_option("inflate_waves", 2)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(rows, f, indent=1)
