"""C3 (4096 x 1 MiB printable ASCII, zlib level 6) whole and in slices of positions: time per batch, device memory the
context's scratch takes, bytes equal.  Usage: python tools/dbg/deflate_slices.py [cap_mib ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import decompress_amd

caps = [int(a) for a in sys.argv[1:]] or [-1, 0, 29491, 16384, 8192]  # -1: the library default
eng = decompress_amd.Engine(0)
dev = eng.device
n, nb = 4096, 1 << 20
g = torch.Generator(device=dev)
g.manual_seed(0xC3)
d_in = torch.randint(0x20, 0x7f, (n * nb,), dtype=torch.uint8, device=dev, generator=g)
cap = nb + nb // 4 + 8192
off = torch.arange(n, dtype=torch.int64, device=dev)
d_off, d_len = off * nb, torch.full((n,), nb, dtype=torch.int64, device=dev)
d_ooff, d_cap = off * cap, torch.full((n,), cap, dtype=torch.int64, device=dev)
ref = None
for c in caps:
    eng.set_option("release_workspace", 1)
    if c >= 0:
        eng.set_option("deflate_workspace_cap_mib", c)
    d_out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=6, queue=4096, total_in=n * nb)
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=6, queue=4096, results=res, total_in=n * nb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    out_len, status, adler = res
    sig = (int(out_len.sum().item()), int(status.abs().sum().item()), int(adler.to(torch.int64).sum().item()),
           int(d_out[:: 4099].to(torch.int64).sum().item()))
    if ref is None:
        ref = sig
    print("cap %6d MiB: %8.1f ms, scratch %7.2f GiB (%.2f B per input byte), same=%s" % (c, ms, used / 2**30, used / (n * nb), sig == ref), flush=True)
    del d_out
