"""randomized stress of the whole-chip path for ONE long stream (capi.cpp inflate_parallel) against libz and the serial
path (test infrastructure; run on the GPU box):
    python tests/stress_long_stream.py [streams] [seed]
Long streams made of stretches of every kind (text, noise in stored blocks, runs, compressed data inside the plaintext),
any level / strategy / memLevel, flush markers of every sort, random piece sizes for the path; sometimes cut, damaged or
given too little room: bytes, counts and statuses must be those of the serial path (and libz's bytes when the stream is good)."""
import ctypes, os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import decompress_amd
from tests.stress_inflate import plain


def one(eng, fmt, z, cap):
    dst = ctypes.create_string_buffer(max(cap, 1))
    used, wrote = ctypes.c_size_t(), ctypes.c_size_t()
    f = {"zl": eng.lib.md_zl_inf_ns_inflate, "de": eng.lib.md_de_inf_ns_inflate}[fmt]
    st = f(eng.ctx, z, len(z), dst, cap, ctypes.byref(used), ctypes.byref(wrote))
    return st, used.value, dst.raw[:wrote.value]


def run(streams, seed, verbose=True):
    rng = random.Random(seed)
    eng = decompress_amd.Engine(0)
    bad = took = 0
    for k in range(streams):
        parts = []
        for _ in range(rng.randrange(2, 9)):
            p = plain(rng, rng.randrange(1000, 900000))
            if rng.random() < 0.25:
                p = zlib.compress(p, 6)  # compressed data inside the plaintext: real block headers that are not this stream's
            parts.append(p)
        data = b"".join(parts)
        wb = 15 if rng.random() < 0.6 else -15
        co = zlib.compressobj(rng.randrange(0, 10), zlib.DEFLATED, wb, rng.randrange(1, 10), rng.choice((0, 0, 0, 1, 2, 3, 4)))
        z = bytearray()
        pos = 0
        while pos < len(data):
            step = rng.choice((len(data), 100000, 30000, 5000, 700))
            z += co.compress(data[pos:pos + step])
            pos += step
            if rng.random() < 0.3:
                z += co.flush(rng.choice((zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH) if hasattr(zlib, "Z_PARTIAL_FLUSH") else (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH)))
        z += co.flush()
        z = bytes(z)
        cap = len(data)
        r = rng.random()
        if r < 0.1: z = z[:rng.randrange(len(z))]
        elif r < 0.2:
            i = rng.randrange(len(z)); z = z[:i] + bytes([z[i] ^ (1 << rng.randrange(8))]) + z[i + 1:]
        elif r < 0.3: cap = max(0, cap - rng.choice((1, 100, 70000)))
        elif r < 0.4: z += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 2000)))
        fmt = "zl" if wb > 0 else "de"
        eng.set_option("inflate_parallel_chunk", rng.choice((4, 16, 64, 64, 256)))
        eng.set_option("inflate_parallel_min", rng.choice((16, 64, 512)))
        got = one(eng, fmt, z, cap)
        v = eng.lib.md_set_option(eng.ctx, b"inflate_parallel_last", 0)
        took += (v & 0xffffff) > 0
        eng.set_option("inflate_parallel_min", 0)
        want = one(eng, fmt, z, cap)
        if got != want or (r >= 0.4 and got != (0, len(z), data)):
            bad += 1
            print("MISMATCH stream %d: %d -> %d bytes, case %.2f: parallel (%d,%d,%d) serial (%d,%d,%d)" % (
                k, len(z), len(data), r, got[0], got[1], len(got[2]), want[0], want[1], len(want[2])), flush=True)
        if verbose and k % 10 == 9:
            print("%d streams, %d by the pieces, %d mismatches so far" % (k + 1, took, bad), flush=True)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("STRESS", "FAILED" if bad else "PASSED")
    sys.exit(1 if bad else 0)
