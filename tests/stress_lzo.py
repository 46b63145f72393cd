"""randomized parity stress of the LZO path against the oracle and minilzo (test infrastructure; run on the GPU box):
    python tests/stress_lzo.py [batches] [seed]
Streams of every kind and length, compressed by the GPU (bytes = the oracle's), by minilzo and by hand-damaged copies;
uncompressed by the GPU into exact, short and generous room: status and bytes = the oracle's (round 6: the decoder works
in batches of instructions with a slow path - this is what keeps the two honest)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import decompress_amd
from decompress_amd import lzo
from tests import oracle_lib
from tests.stress_inflate import plain


def run(batches, seed, verbose=True):
    rng = random.Random(seed)
    eng, orc, m = decompress_amd.Engine(0), oracle_lib.load(), oracle_lib.load_minilzo()
    bad = 0
    for b in range(batches):
        datas = []
        for i in range(rng.choice((9, 400, 1500))):
            n = rng.choice((0, 1, 5, 19, 20, 21, 300, 3000, 49152, 49153, 131072, 300000)) if rng.random() < 0.4 else rng.randrange(0, 90000)
            d = plain(rng, n)
            if rng.random() < 0.2:  # long runs and long matches: lengths that go on over zero bytes
                d = (d[:n // 3] + bytes([rng.getrandbits(8)]) * rng.randrange(0, 3000) + d[:n // 2] + d[:n // 3])[:max(n, 1)]
            datas.append(d)
        zs = eng.lzo_many(True, datas, [lzo.max_compressed_length(len(d)) for d in datas])
        srcs, caps = [], []
        for d, (st, z) in zip(datas, zs):
            ost, oz = orc.lzo_compress(d)
            if (st, z) != (ost, oz):
                bad += 1
                print("COMPRESS MISMATCH batch %d len %d" % (b, len(d)), flush=True)
            src = m.compress(d) if (m is not None and rng.random() < 0.3) else z
            r = rng.random()
            if r < 0.1 and len(src) > 2: src = src[:rng.randrange(len(src))]
            elif r < 0.25 and len(src) > 4:
                k = rng.randrange(len(src)); src = src[:k] + bytes([rng.getrandbits(8)]) + src[k + 1:]
            cap = len(d) + rng.choice((0, 0, 0, 1, 3, 70, 5000)) if rng.random() < 0.8 else max(0, len(d) - rng.choice((1, 2, 3, 30, 1000)))
            srcs.append(src); caps.append(cap)
        # the compressor again into room that is (mostly) too small: status and bytes = the oracle's
        small = [max(0, len(z) - rng.choice((0, 1, 2, 3, 4, 17, 300))) if rng.random() < 0.7 else rng.randrange(0, len(z) + 1) for _, z in zs]
        for d, cap, got in zip(datas, small, eng.lzo_many(True, datas, small)):
            if got != orc.lzo_compress(d, cap=cap):
                bad += 1
                print("COMPRESS (short room) MISMATCH batch %d len %d cap %d" % (b, len(d), cap), flush=True)
        res = eng.lzo_many(False, srcs, caps)
        for src, cap, (st, out) in zip(srcs, caps, res):
            ost, oout = orc.lzo_uncompress(src, cap)
            if (st, out) != (ost, oout):
                bad += 1
                print("MISMATCH batch %d: len %d cap %d  gpu (%d,%d)  oracle (%d,%d)" % (b, len(src), cap, st, len(out), ost, len(oout)), flush=True)
        if verbose:
            print("batch %d: %d streams, %d mismatches so far" % (b, len(srcs), bad), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("STRESS", "FAILED" if bad else "PASSED")
    sys.exit(1 if bad else 0)
