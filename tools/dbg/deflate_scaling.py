"""How the three deflate kernels' time depends on the number of 1 MiB printable-ASCII streams in the launch
(run under rocprofv3 --kernel-trace --stats, or read the HIP-event total).  Usage: deflate_scaling.py n [n ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decompress_amd

eng = decompress_amd.Engine(0)
eng.set_option("deflate_workspace_cap_mib", 0)
dev = eng.device
nb = 1 << 20
for n in [int(a) for a in sys.argv[1:]]:
    g = torch.Generator(device=dev)
    g.manual_seed(0xC3)
    d_in = torch.randint(0x20, 0x7f, (n * nb,), dtype=torch.uint8, device=dev, generator=g)
    cap = nb + nb // 4 + 8192
    off = torch.arange(n, dtype=torch.int64, device=dev)
    d_off, d_len = off * nb, torch.full((n,), nb, dtype=torch.int64, device=dev)
    d_ooff, d_cap = off * cap, torch.full((n,), cap, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=6, queue=4096, total_in=n * nb)
    eng.synchronize()
    eng.timing_begin()
    res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=6, queue=4096, results=res, total_in=n * nb)
    ms = eng.timing_end()
    print("n %5d: %8.2f ms all three kernels" % (n, ms), flush=True)
    del d_in, d_out
