"""randomized parity stress of the two ways a stream crosses launches (test infrastructure; run on the GPU box):
    python tests/stress_pieces.py [rounds] [seed]
  * batches in slices of positions (md_set_option "deflate_workspace_cap_mib" small enough to force them): every byte,
    status and checksum equal to the same batch taken whole, and to the oracle;
  * the md_def_* encoder fed in pieces of a random size: every byte equal to the oracle handed the same pieces."""
import ctypes, os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import decompress_amd
from tests import oracle_lib
from tests.stress_inflate import plain


def encode_in_pieces(eng, fmt, data, piece, level, queue):
    lib = eng.lib
    params = eng._params(level, queue, 0, True)
    o = ctypes.create_string_buffer(8192)
    s = lib.md_def_encoder(eng.ctx, fmt, ctypes.byref(params), o, len(o))
    out, pos = bytearray(), 0
    while True:
        sig = lib.md_def_encode(s)
        if sig == 0:
            chunk = data[pos:pos + piece]
            pos += len(chunk)
            assert lib.md_def_src(s, chunk, 0, len(chunk)) == 0
        elif sig in (1, 2):
            out += o.raw[:len(o) - lib.md_def_dst_rem(s)]
            if sig == 2:
                break
            lib.md_def_dst(s, o, len(o))
        else:
            raise AssertionError(lib.md_def_status(s))
    st = lib.md_def_status(s)
    lib.md_def_free(s)
    return st, bytes(out)


def run(rounds, seed):
    rng = random.Random(seed)
    eng, orc = decompress_amd.Engine(0), oracle_lib.load()
    bad = 0
    for r in range(rounds):
        # ---- slices of positions
        level, queue = rng.randrange(1, 10), rng.choice((256, 1024, 4096, 4096))
        driver = rng.choice((0, 0, 1, 2))
        fmt = rng.choice((decompress_amd.FORMAT_ZLIB, decompress_amd.FORMAT_GZIP, decompress_amd.FORMAT_DEFLATE)) if driver == 0 else decompress_amd.FORMAT_DEFLATE
        lens = [rng.choice((0, 1, 3, 32768, 65536, 65537, 98304, 131072)) + rng.choice((0, 0, 1, -1, 262, -262, 5000)) if rng.random() < 0.5
                else rng.randrange(0, 600000) for _ in range(rng.choice((1, 5, 20)))]
        bufs = [plain(rng, max(0, n)) for n in lens]
        eng.set_option("deflate_workspace_cap_mib", 0)
        want = eng.deflate_many(bufs, fmt, level=level, queue=queue, driver=driver)
        eng.set_option("deflate_workspace_cap_mib", rng.choice((1, 2, 3, 5, 9)))
        got = eng.deflate_many(bufs, fmt, level=level, queue=queue, driver=driver)
        eng.set_option("deflate_workspace_cap_mib", 0)
        for i, (w, g) in enumerate(zip(want, got)):
            if w != g:
                bad += 1
                print("SLICES differ: round %d stream %d len %d fmt %d level %d queue %d driver %d: %s / %s" % (
                    r, i, len(bufs[i]), fmt, level, queue, driver, (w[0], len(w[1])), (g[0], len(g[1]))), flush=True)
        if fmt == decompress_amd.FORMAT_ZLIB and driver == 0:
            for b, (st, z, _) in zip(bufs, got):
                if st != 0 or z != orc.zl_deflate(b, level=level, queue=queue):
                    bad += 1
                    print("SLICES vs oracle: round %d len %d level %d queue %d" % (r, len(b), level, queue), flush=True)
        # ---- the encoder in pieces
        piece = rng.choice((1, 7, 263, 1000, 4096, 32768, 50000, 65536, 100000, 300000))
        n = rng.randrange(0, 400) if piece < 100 else rng.randrange(0, 30000) if piece < 2000 else rng.randrange(0, 500000)
        data = plain(rng, n)
        level, queue = rng.randrange(0, 10), rng.choice((16, 256, 4096, 4096))
        fmt = rng.choice((decompress_amd.FORMAT_ZLIB, decompress_amd.FORMAT_GZIP))
        eng.set_option("encoder_piece_bytes", piece)
        eng.set_option("deflate_test_flags", 16 if rng.random() < 0.3 else 0)
        st, got1 = encode_in_pieces(eng, fmt, data, piece, level, queue)
        eng.set_option("encoder_piece_bytes", 1 << 20)
        eng.set_option("deflate_test_flags", 0)
        with orc.src_piece(piece):
            want1 = orc.zl_deflate(data, level=level, queue=queue) if fmt == decompress_amd.FORMAT_ZLIB else orc.gz_deflate(data, level=level, queue=queue)
        if st != 0 or got1 != want1:
            bad += 1
            print("PIECES differ: round %d n %d piece %d fmt %d level %d queue %d: status %d, %d / %d bytes" % (
                r, n, piece, fmt, level, queue, st, len(got1), len(want1)), flush=True)
        if (r + 1) % 10 == 0:
            print("round %d: %d mismatches so far" % (r + 1, bad), flush=True)
    print("done: %d mismatches in %d rounds" % (bad, rounds))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
