import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
n, nb, kind = int(sys.argv[1]), int(sys.argv[2]) * 1024, sys.argv[3]
gen = workloads.ascii_uniform if kind == "ascii" else workloads.text
eng = decompress_amd.Engine(0)
eng.set_option("profile", 1)
bufs = [gen(0xC3 + i, nb) for i in range(n)]
eng.timing_begin()
outs = eng.deflate_many(bufs, level=6, fmt=decompress_amd.FORMAT_ZLIB)
ms = eng.timing_end()
p = eng.get_profile_raw()
names = ["setup", "ring fill", "parse (literal runs + lazy steps)", "lane0", "pack", "trees"]
tot = sum(p[:6])
print("kernel ms %.2f  ticks of stream 0: %d (100 MHz -> %.2f ms)" % (ms, tot, tot / 1e5))
for k, v in zip(names, p[:6]):
    print("  %-36s %6.1f%%  %.2f ms" % (k, 100.0 * v / max(tot, 1), v / 1e5))
print("  prep batches %d  bulk steps %d  bulk literals %d  iterations %d" % tuple(p[8:12]))
print("  matcher steps %d  longest_match calls %d  chain links %d" % tuple(p[12:15]))
for i, nm in enumerate(["PREP", "WRITE", "DONE", "TREES"]):
    print("  lane-0 turns ending in %-5s: %5d turns, %.2f ms total, %.2f us avg" % (nm, p[20 + i], p[16 + i] / 1e5, p[16 + i] / 100.0 / max(p[20 + i], 1)))
print("  longest lane-0 turn %.1f us" % (p[24] / 100.0))
print("  lane-0 section: entry %.2f ms  stream_step %.2f ms  publish %.2f ms" % (p[25] / 1e5, p[26] / 1e5, p[27] / 1e5))
tp = []
for w in p[28:32]:
    tp += [w & 0xffffffff, w >> 32]
print("  trees: fill+heapify %.2f  merge %.2f  lengths %.2f  codes %.2f  scan %.2f  symbols+cost %.2f ms" % tuple(x / 1e5 for x in tp[:6]))
