#!/usr/bin/env python3
"""Run ON THE GPU BOX: one long stream through the Higher.uncompress entry points (round 6: decoded in pieces by the
whole chip, capi.cpp inflate_parallel) - bytes against libz, host-to-host rate, what the path did."""
import ctypes, sys, time, zlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads

eng = decompress_amd.Engine(0)
lib, ctx = eng.lib, eng.ctx
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def last():
    v = lib.md_set_option(ctx, b"inflate_parallel_last", 0)
    return v & 0xffffff, v >> 24


def run(name, fn, z, want, cap=None):
    cap = len(want) if cap is None else cap
    dst = ctypes.create_string_buffer(max(cap, 1))
    used, wrote = ctypes.c_size_t(), ctypes.c_size_t()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        if fn == "zl":
            st = lib.md_zl_higher_uncompress(ctx, z, len(z), dst, cap, ctypes.byref(wrote))
        elif fn == "de":
            st = lib.md_de_higher_uncompress(ctx, z, len(z), dst, cap, ctypes.byref(wrote))
        elif fn == "zlns":
            st = lib.md_zl_inf_ns_inflate(ctx, z, len(z), dst, cap, ctypes.byref(used), ctypes.byref(wrote))
        else:
            st = lib.md_gz_higher_uncompress(ctx, z, len(z), dst, cap, ctypes.byref(used), ctypes.byref(wrote), None)
        best = min(best, time.perf_counter() - t0)
    ok = st == 0 and wrote.value == len(want) and dst.raw[:wrote.value] == want
    print("%-34s st %d ok %s  %7.1f ms  %8.1f MiB/s  pieces, rounds = %s  used %d of %d" % (
        name, st, ok, best * 1e3, len(want) / 2**20 / best, last(), used.value, len(z)), flush=True)
    return st, ok


data = workloads.text(77, mib << 20)
z = zlib.compress(data, 6)
print("text %d MiB -> %d bytes" % (mib, len(z)))
eng.set_option("inflate_parallel_min", 0)
run("zl serial (one pair of waves)", "zl", z[:len(z)], data) if mib <= 16 else None
eng.set_option("inflate_parallel_min", 96)
run("zl parallel", "zl", z, data)
run("zl.ns parallel", "zlns", z, data)
co = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = co.compress(data) + co.flush()
run("raw parallel", "de", raw, data)
co = zlib.compressobj(6, zlib.DEFLATED, 31)
gz = co.compress(data) + co.flush()
run("gzip parallel", "gz", gz, data)
# a wrong checksum, a cut stream, too little room: the serial path's answers
bad = bytearray(z); bad[-1] ^= 1
print("bad adler:", run("zl bad checksum", "zl", bytes(bad), data))
print("cut:", run("zl cut", "zl", z[:len(z) // 2], data))
print("short room:", run("zl short room", "zl", z, data, cap=len(data) - 5))
# other kinds of data
for name, d in (("markov", workloads.markov_text(5, min(mib, 8) << 20)), ("ascii noise", workloads.ascii_uniform(3, min(mib, 16) << 20)),
                ("zeros", bytes(min(mib, 16) << 20)), ("corpus-like mix", (workloads.text(1, 1 << 20) + workloads.ascii_uniform(2, 1 << 20)) * min(mib // 2, 8))):
    for lvl in (1, 6, 9):
        zz = zlib.compress(d, lvl)
        run("%s level %d" % (name, lvl), "zl", zz, d)
# full-flush units and stored blocks
co = zlib.compressobj(6)
parts = []
d8 = data[:8 << 20]
for i in range(0, len(d8), 40000):
    parts.append(co.compress(d8[i:i + 40000])); parts.append(co.flush(zlib.Z_FULL_FLUSH))
parts.append(co.flush())
run("Z_FULL_FLUSH every 40 KB", "zl", b"".join(parts), d8)
co = zlib.compressobj(6)
parts = []
d1 = data[:800000]
for i in range(0, len(d1), 40):
    parts.append(co.compress(d1[i:i + 40])); parts.append(co.flush(zlib.Z_FULL_FLUSH))
parts.append(co.flush())
run("20 000 Z_FULL_FLUSH units of 40 B", "zl", b"".join(parts), d1)
run("stored (level 0)", "zl", zlib.compress(data[:8 << 20], 0), data[:8 << 20])
