"""randomized parity stress of the inflate path against the oracle (test infrastructure; run on the GPU box):
    python tests/stress_inflate.py [batches] [seed]"""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import decompress_amd
from decompress_amd import workloads
from tests import oracle_lib

def plain(rng, n):
    k = rng.randrange(9)
    if k == 0: return bytes(rng.getrandbits(8) for _ in range(n))
    if k == 1: return workloads.text(rng.randrange(1 << 30), n)
    if k == 2: return bytes([rng.randrange(4)]) * n
    if k == 3:
        unit = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 300)))
        return (unit * (n // len(unit) + 1))[:n]
    if k == 4: return workloads.corpus_slice(rng.randrange(1 << 20), n)
    if k == 5:
        grams = [bytes(rng.getrandbits(8) for _ in range(rng.randrange(3, 6))) for _ in range(rng.randrange(2, 60))]
        out = bytearray()
        while len(out) < n: out += rng.choice(grams)
        return bytes(out[:n])
    if k == 6: return bytes(rng.choice(b"ab") for _ in range(n))
    if k == 7:
        a = workloads.corpus_slice(rng.randrange(1 << 20), n // 2 + 1)
        return (a + bytes(rng.randrange(1, 5000)) + a)[:n]
    return workloads.ascii_uniform(rng.randrange(1 << 30), n)

def run(batches, seed, verbose=True):
    """-> number of streams whose (status, consumed, output bytes) differ from the oracle's"""
    rng = random.Random(seed)
    eng, orc = decompress_amd.Engine(0), oracle_lib.load()
    bad = 0
    for b in range(batches):
        srcs, caps = [], []
        for i in range(rng.choice((7, 300, 2300))):
            n = rng.choice((0, 1, 5, 100, 3000, 40000, 200000)) if rng.random() < 0.5 else rng.randrange(0, 70000)
            pl = plain(rng, n)
            co = zlib.compressobj(rng.randrange(0, 10), zlib.DEFLATED, 15, rng.randrange(1, 10), rng.choice((0, 0, 0, 1, 2, 3, 4)))
            z = co.compress(pl) + co.flush()
            r = rng.random()
            if r < 0.08 and len(z) > 2: z = z[:rng.randrange(len(z))]
            elif r < 0.16 and len(z) > 8:
                k = rng.randrange(len(z)); z = z[:k] + bytes([z[k] ^ (1 << rng.randrange(8))]) + z[k + 1:]
            cap = len(pl) + rng.choice((0, 0, 0, 1, 7, 40, 5000)) if rng.random() < 0.85 else max(0, len(pl) - rng.choice((1, 2, 30, 1000)))
            srcs.append(z); caps.append(cap)
        res = eng.inflate_many(srcs, caps, decompress_amd.FORMAT_ZLIB)
        for z, cap, (st, used, out, _) in zip(srcs, caps, res):
            ost, oused, oout = orc.zl_inflate(z, cap)
            # the oracle's output on a failing stream is everything before the failing token, like the kernel's
            if (st, used) != (ost, oused) or out != oout:
                bad += 1
                print("MISMATCH batch %d: len %d cap %d  gpu (%d,%d,%d)  oracle (%d,%d,%d)" % (b, len(z), cap, st, used, len(out), ost, oused, len(oout)), flush=True)
        if verbose:
            print("batch %d: %d streams, %d mismatches so far" % (b, len(srcs), bad), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("STRESS", "FAILED" if bad else "PASSED")
    sys.exit(1 if bad else 0)
