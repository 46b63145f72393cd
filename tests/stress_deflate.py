"""randomized parity stress of the deflate path against the oracle (test infrastructure; run on the GPU box):
    python tests/stress_deflate.py [batches] [seed]
Every output byte, status and checksum of Zl.Def / De.Higher / the CLI driver / Gz.Def / the Lz matcher, levels 0-9,
queues 256..16384, fixed or dynamic blocks, exact-fit and short capacities."""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import decompress_amd
from decompress_amd import engine
from tests import oracle_lib
from tests.stress_inflate import plain


def run(batches, seed, verbose=True):
    rng = random.Random(seed)
    eng, orc = decompress_amd.Engine(0), oracle_lib.load()
    bad = 0
    for b in range(batches):
        level, queue = rng.randrange(0, 10), rng.choice((256, 1024, 4096, 4096, 16384))
        driver, dynamic = rng.choice((0, 0, 1, 2)), rng.random() < 0.8
        fmt = rng.choice((decompress_amd.FORMAT_DEFLATE, decompress_amd.FORMAT_ZLIB)) if driver == 0 else decompress_amd.FORMAT_DEFLATE
        matcher = rng.choice((0, 0, 0, 1)) if fmt == decompress_amd.FORMAT_DEFLATE else 0
        if driver == 1:
            level = 4
        bufs = [plain(rng, rng.choice((0, 1, 2, 3, 40, 700, 5000, 33000, 70000, 140000)) if rng.random() < 0.6 else rng.randrange(0, 50000))
                for _ in range(rng.choice((3, 40, 130)))]
        want = []
        for d in bufs:
            if fmt == decompress_amd.FORMAT_ZLIB:
                want.append(orc.zl_deflate(d, level=level, queue=queue, dynamic=dynamic))
            else:
                want.append(orc.deflate_raw(d, level=level, queue=queue, driver=driver, dynamic=dynamic, matcher=matcher)[0])
        caps = None
        if rng.random() < 0.3 and all(w is not None for w in want):
            caps = [len(w) - (1 if (i % 3 == 0 and len(w) > 0) else 0) for i, w in enumerate(want)]
        res = eng.deflate_many(bufs, fmt, level=level, queue=queue, driver=driver, dynamic=dynamic, matcher=matcher, caps=caps)
        for i, (d, w, (st, out, _)) in enumerate(zip(bufs, want, res)):
            short = caps is not None and caps[i] < len(w)
            ok = (st == 13) if w is None else (st == 2) if short else (st == 0 and out == w)  # 13 = Queue.Full (the CLI driver)
            if not ok:
                bad += 1
                print("MISMATCH batch %d stream %d: len %d level %d queue %d driver %d dynamic %s matcher %d fmt %d short %s -> status %d, %d vs %s bytes" % (
                    b, i, len(d), level, queue, driver, dynamic, matcher, fmt, short, st, len(out), None if w is None else len(w)), flush=True)
        if verbose:
            print("batch %d: %d buffers level %d queue %d driver %d, %d mismatches so far" % (b, len(bufs), level, queue, driver, bad), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("STRESS", "FAILED" if bad else "PASSED")
    sys.exit(1 if bad else 0)
