"""Differential fuzzing of the HIP path against the oracle, in the spirit of the reference's
fuzz/fuzz.ml + fuzz/fuzz_ns.ml (random bytes must never crash the decoder, anything the
encoder emits must decode back): thousands of garbage / corrupted streams in one batch, every
status, count and output byte compared with the oracle.  Needs an MI355X: `pytest -m gpu`."""
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import decompress_amd
    return decompress_amd.Engine(0)


def _plain(rng, n):
    kind = rng.randrange(4)
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 1:
        words = [bytes(rng.choice(b"abcdefgh ") for _ in range(rng.randrange(1, 9))) for _ in range(30)]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words)
        return bytes(out[:n])
    if kind == 2:
        return bytes([rng.randrange(4)]) * n
    return bytes(rng.choice(b"01") for _ in range(n))


def _corrupt(rng, z):
    b = bytearray(z)
    for _ in range(rng.randrange(1, 4)):
        how = rng.randrange(5)
        if not b:
            break
        i = rng.randrange(len(b))
        if how == 0:
            b[i] ^= 1 << rng.randrange(8)
        elif how == 1:
            b[i] = rng.getrandbits(8)
        elif how == 2:
            del b[i:i + rng.randrange(1, 5)]
        elif how == 3:
            b[i:i] = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 5)))
        else:
            del b[i:]
    return bytes(b)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_inflate_garbage_and_corruption(eng, oracle, seed):
    rng = random.Random(seed)
    srcs, caps = [], []
    for _ in range(700):  # pure garbage: mostly invalid block kinds / dictionaries / distances
        n = rng.choice((0, 1, 2, 3, 5, 8, 13, 40, 200, 1500))
        g = bytearray(rng.getrandbits(8) for _ in range(n))
        if g and rng.random() < 0.7:
            g[0] = (g[0] & 0xf8) | rng.choice((1, 3, 5, 4, 2, 0))  # steer BFINAL/BTYPE
        srcs.append(bytes(g)); caps.append(rng.choice((0, 7, 300, 5000)))
    for _ in range(500):  # corrupted valid streams of every block kind
        data = _plain(rng, rng.choice((0, 1, 50, 700, 5000, 40000)))
        strat = rng.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE))
        co = zlib.compressobj(rng.choice((0, 1, 6, 9)), zlib.DEFLATED, -15, rng.choice((1, 8, 9)), strat)
        z = co.compress(data) + co.flush()
        srcs.append(_corrupt(rng, z) if rng.random() < 0.85 else z)
        caps.append(max(0, len(data) + rng.choice((-3, -1, 0, 0, 1, 100))))
    res = eng.inflate_many(srcs, caps)
    seen = set()
    for k, (src, cap, (st, used, out, _)) in enumerate(zip(srcs, caps, res)):
        ost, oused, oout = oracle.de_inflate(src, cap)
        assert (st, used) == (ost, oused), (seed, k, len(src), cap, st, ost, used, oused)
        assert out == oout, (seed, k)
        seen.add(st)
    assert seen >= {0, 1, 2, 3, 4, 5, 6, 7}  # every De.Inf.Ns error variant was exercised


def test_zlib_and_gzip_frames_corrupted(eng, oracle):
    import decompress_amd
    rng = random.Random(77)
    zs, gs, caps = [], [], []
    for _ in range(300):
        data = _plain(rng, rng.choice((0, 10, 3000, 30000)))
        z = zlib.compress(data, rng.choice((1, 6, 9)))
        g = oracle.gz_deflate(data, level=rng.choice((1, 4, 6)), name=b"n" if rng.random() < 0.5 else None,
                              comment=b"c" if rng.random() < 0.3 else None, hcrc=rng.random() < 0.5)
        zs.append(_corrupt(rng, z) if rng.random() < 0.8 else z)
        gs.append(_corrupt(rng, g) if rng.random() < 0.8 else g)
        caps.append(len(data) + rng.choice((0, 0, 5, -1 if data else 0)))
    for fmt, srcs, ref in ((decompress_amd.FORMAT_ZLIB, zs, oracle.zl_inflate),
                           (decompress_amd.FORMAT_GZIP, gs, lambda s, c: oracle.gz_inflate(s, c)[:3])):
        res = eng.inflate_many(srcs, caps, fmt)
        for k, (src, cap, (st, used, out, _)) in enumerate(zip(srcs, caps, res)):
            ost, oused, oout = ref(src, cap)
            assert (st, used) == (ost, oused), (fmt, k, st, ost, used, oused)
            if st in (0, 9, 12):
                assert out == oout


@pytest.mark.parametrize("seed", [5, 6])
def test_deflate_random_parameters(oracle, seed):
    """random data kinds x level x queue x driver x matcher: the oracle's bytes, and they decode"""
    import decompress_amd
    rng = random.Random(seed)
    for matcher in (0, 1):
        e = decompress_amd.Engine(0)
        e.set_matcher(matcher)
        for _ in range(6):
            level, q, driver = rng.randrange(10), rng.choice((4, 16, 256, 4096, 16384)), rng.randrange(3)
            dyn = rng.random() < 0.8
            bufs = [_plain(rng, rng.choice((0, 1, 2, 3, 4, 5, 100, 263, 5000, 33000, 70000))) for _ in range(10)]
            wants = [oracle.deflate_raw(b, level, q, driver, dyn, matcher)[0] for b in bufs]
            res = e.deflate_many(bufs, level=level, queue=q, driver=driver, dynamic=dyn,
                                 caps=[len(w) + 1 if w is not None else 64 for w in wants])
            for b, w, (st, out, adler) in zip(bufs, wants, res):
                if w is None:  # De.Queue.Full in the reference (CLI driver's extra EOB into a full queue)
                    assert st == 13 and out == b""
                    continue
                assert st == 0 and out == w, (seed, matcher, level, q, driver, dyn, len(b))
                assert zlib.decompress(out, -15) == b and adler == zlib.adler32(b)


def test_random_streams_every_field():
    """tests/stress_inflate.py, a bounded run: random plaintexts of nine kinds (random, text, runs, periods, corpus
    slices, n-gram soup, two-symbol, repeated halves, ASCII) compressed at random levels / memory levels /
    strategies, some truncated or with a flipped bit, output capacities exact, generous or short, batches of 7,
    300 and 2300 streams: status, consumed count AND the bytes written before a failure equal the oracle's"""
    from tests import stress_inflate
    assert stress_inflate.run(8, 20260928, verbose=False) == 0


def test_random_buffers_deflate_every_byte():
    """tests/stress_deflate.py, a bounded run: random buffers of nine kinds through Zl.Def, De.Higher, the CLI
    driver and the Lz matcher at random levels, queue sizes and block kinds, exact-fit and one-byte-short output
    capacities: every compressed byte and status equals the oracle's (Queue.Full included)"""
    from tests import stress_deflate
    assert stress_deflate.run(12, 20260928, verbose=False) == 0


def test_streaming_decoder_random_pieces(eng):
    """The decoder protocol fed in random pieces with random piece sizes (md_inf_chunk_bytes): streams of mixed block
    kinds (stored / fixed / dynamic / empty blocks, so block ends fall on every bit offset and stored blocks cross the
    pieces), raw, ZLIB and GZip framed, some cut short.  What comes out is what zlib inflates, signal by signal."""
    import ctypes
    import gzip
    from decompress_amd import de
    import decompress_amd
    rng = random.Random(0xfeed)
    lib = eng.lib
    for case in range(200):
        body, plain = bytearray(), bytearray()
        nseg = rng.randrange(1, 7)
        for k in range(nseg):
            data = _plain(rng, rng.choice((0, 1, 7, 300, 5000, 40000, 70000, 150000)))
            level, strat = rng.choice(((0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY)))
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            body += co.compress(data) + (co.flush() if k == nseg - 1 else co.flush(rng.choice((zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH))))
            plain += data
        raw, plain = bytes(body), bytes(plain)
        fmt = rng.choice((decompress_amd.FORMAT_DEFLATE, decompress_amd.FORMAT_ZLIB, decompress_amd.FORMAT_GZIP))
        if fmt == decompress_amd.FORMAT_ZLIB:
            src = b"\x78\x9c" + raw + zlib.adler32(plain).to_bytes(4, "big")
        elif fmt == decompress_amd.FORMAT_GZIP:
            src = b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + raw + zlib.crc32(plain).to_bytes(4, "little") + (len(plain) & 0xffffffff).to_bytes(4, "little")
        else:
            src = raw
        cut = rng.random() < 0.2 and len(src) > 20
        if cut:
            src = src[:rng.randrange(10, len(src) - 1)]
        o_len = rng.choice((1, 100, 4096, 65536))
        o = ctypes.create_string_buffer(o_len)
        d = lib.md_inf_decoder(eng.ctx, fmt, o, o_len)
        lib.md_inf_chunk_bytes(d, rng.choice((1, 50, 700, 9000, 100000)))
        out, pos = bytearray(), 0
        try:
            for _ in range(10_000_000):
                sig = lib.md_inf_decode(d)
                if sig == de.AWAIT:
                    k = min(len(src) - pos, rng.choice((1, 13, 1000, 30000, 1 << 20)))
                    lib.md_inf_src(d, src, pos, k)  # k == 0: the end of the input
                    pos += k
                else:
                    out += o.raw[:o_len - lib.md_inf_dst_rem(d)]
                    lib.md_inf_flush(d)
                    if sig in (de.END, de.MALFORMED):
                        break
            st = lib.md_inf_status(d)
        finally:
            lib.md_inf_free(d)
        if cut:
            assert sig == de.MALFORMED and st != 0, (case, fmt, st)
            assert plain.startswith(bytes(out)), (case, fmt, len(out))
        else:
            assert sig == de.END and st == 0, (case, fmt, st)
            assert bytes(out) == plain, (case, fmt, len(out), len(plain))
