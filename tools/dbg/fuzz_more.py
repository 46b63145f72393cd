"""One-off soak: the differential fuzz tests of tests/test_gpu_fuzz.py and the mixed-block test of test_gpu_inflate.py with many
more seeds, on both forms of the inflate kernel.  python tools/dbg/fuzz_more.py [first_seed] [count]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import decompress_amd
from tests import oracle_lib
import tests.test_gpu_fuzz as F

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng, orc = decompress_amd.Engine(0), oracle_lib.load()
t0, fails = time.time(), 0
for k in range(count):
    seed = first + k
    for waves in (2, 1):
        eng.set_option("inflate_waves", waves)
        try:
            F.test_inflate_garbage_and_corruption(eng, orc, seed)
        except AssertionError as e:
            fails += 1
            print("FAIL inflate seed", seed, "waves", waves, str(e)[:200], flush=True)
    eng.set_option("inflate_waves", 2)
    try:
        F.test_deflate_random_parameters(orc, seed)
    except AssertionError as e:
        fails += 1
        print("FAIL deflate seed", seed, str(e)[:200], flush=True)
    print("seed", seed, "ok so far, %.0f s" % (time.time() - t0), flush=True)
print("done: %d seeds, %d failures" % (count, fails))
