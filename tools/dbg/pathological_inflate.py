import sys, os, zlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
def raw(data, level=6, strat=0):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
    return co.compress(data) + co.flush()
cases = {}
cases["zeros 4 MiB L9"] = (raw(bytes(4 << 20), 9), 4 << 20)
cases["50k empty stored blocks"] = (b"\x00\x00\x00\xff\xff" * 50000 + b"\x01\x00\x00\xff\xff", 16)
cases["50k empty fixed blocks"] = None
# empty fixed block = bits: BFINAL=0, BTYPE=01, EOB(7 bits 0000000) = 10 bits; build a bitstream
bits = []
for _ in range(50000): bits += [0, 1, 0] + [0] * 7
bits += [1, 1, 0] + [0] * 7
b = bytearray((len(bits) + 7) // 8)
for i, v in enumerate(bits):
    if v: b[i >> 3] |= 1 << (i & 7)
cases["50k empty fixed blocks"] = (bytes(b), 16)
import random
rng = random.Random(1)
cases["random 2 MiB (stored)"] = (raw(bytes(rng.getrandbits(8) for _ in range(2 << 20))), 2 << 20)
cases["text 2 MiB L1"] = (raw(workloads.text(5, 2 << 20), 1), 2 << 20)
cases["text 2 MiB fixed huffman"] = (raw(workloads.text(6, 2 << 20), 6, zlib.Z_FIXED), 2 << 20)
cases["many 1-byte dynamic blocks"] = None
co = zlib.compressobj(6, zlib.DEFLATED, -15)
out = b""
for i in range(20000): out += co.compress(bytes([65 + i % 26]) * 40) + co.flush(zlib.Z_FULL_FLUSH)
out += co.flush()
cases["20k full-flush blocks"] = (out, 20000 * 40)
del cases["many 1-byte dynamic blocks"]
import numpy as np
import torch
for name, (z, n) in cases.items():
    st, used, o, _ = eng.inflate_many([z], [n])[0]
    want = zlib.decompressobj(-15).decompress(z)
    # the kernel alone (HIP events), input and output resident: the call above also pays for allocation and copies
    dev = eng.device
    d_in = torch.from_numpy(np.frombuffer(z + bytes(16), dtype=np.uint8).copy()).to(dev)
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    t = lambda v: torch.tensor([v], dtype=torch.int64, device=dev)
    args = (decompress_amd.FORMAT_DEFLATE, d_in, t(0), t(len(z)), d_out, t(0), t(n))
    eng.inflate_batch(*args)
    eng.synchronize()
    eng.timing_begin()
    eng.inflate_batch(*args)
    ms = eng.timing_end()
    print("%-28s in %8d out %8d status %d ok=%s  kernel %.2f ms" % (name, len(z), len(o), st, o == want[:n] and used == len(z), ms), flush=True)
