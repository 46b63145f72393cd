#!/usr/bin/env python3
"""Where a stream's time goes in the LZO decoder (a -DMD_LZO_PROF build leaves cycle counts in out_len):
    MD_LIBMDEFLATE=tools/dbg/variants/lib_lzoprof.so python tools/dbg/lzo_phases.py [streams]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import decompress_amd
from decompress_amd import workloads, lzo
from tests import oracle_lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = 128 * 1024
orc = oracle_lib.load()
uniq = [workloads.text(0xC5 + 2 * i, nb) for i in range(32)]
zs = [orc.lzo_compress(u)[1] for u in uniq]
eng = decompress_amd.Engine(0)
bufs = [zs[i % 32] for i in range(n)]
blob, off, ln = workloads.pack(bufs, align=32)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(a).to(dev)
d_out = torch.zeros(n * nb + 64, dtype=torch.uint8, device=dev)
ooff = np.arange(n, dtype=np.int64) * nb
ocap = np.full(n, nb, dtype=np.int64)
r = eng.lzo_batch(False, t(blob), t(off), t(ln), d_out, t(ooff), t(ocap))
torch.cuda.synchronize()
v = r[0].cpu().numpy().astype(np.uint64)
tot = r[1].cpu().numpy().astype(np.float64) * 1024
names = ("literals into staging, records (rest of the windows)", "far matches (fence, loads)", "near matches", "write-out, slow path",
         "windows: ring, decode of both states", "windows: the walk", "windows: places, cuts", "-")
print("cycles per stream: %.0f K" % (tot.mean() / 1e3))
for k, nm in enumerate(names):
    c = ((v >> np.uint64(8 * k)) & np.uint64(0xff)).astype(np.float64) / 255
    print("%-56s %5.1f %%" % (nm, 100 * c.mean()))
