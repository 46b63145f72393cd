"""one long stream through the De.Inf.decode protocol (pieces of 1 MiB of input): how fast is a single stream?"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import decompress_amd
from decompress_amd import de, workloads
plain = workloads.text(0x51, 64 << 20)
z = zlib.compress(plain, 6)
eng = decompress_amd.Engine(0)
for chunk in (1 << 20, 8 << 20, 32 << 20):
    t0 = time.perf_counter()
    verdict, out, sigs = de.Inf.decode_chunks((z[i:i + (256 << 10)] for i in range(0, len(z), 256 << 10)), o_len=1 << 20,
                                              fmt=decompress_amd.FORMAT_ZLIB, chunk_bytes=chunk)
    dt = time.perf_counter() - t0
    print("chunk %d MiB: %s, %d MiB out in %.2f s = %.0f MiB/s, %d Flush before the last input" % (
        chunk >> 20, verdict, len(out) >> 20, dt, len(out) / 2**20 / dt, sum(1 for s in sigs if s == de.FLUSH)), out == plain, flush=True)
# the tool's call shape (bin/decompress.ml:77-100): everything at once
from decompress_amd import cli
t0 = time.perf_counter()
st, out, msg = cli.run(False, "zlib", 4, z)
dt = time.perf_counter() - t0
print("cli -d -f zlib: exit %d, %.0f MiB/s" % (st, len(out) / 2**20 / dt), out == plain)
