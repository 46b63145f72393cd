"""Parity of the HIP inflate path (through the C ABI) with the CPU oracle and
the reference's golden vectors.  Needs an MI355X: `pytest -m gpu`."""
import ctypes
import random
import zlib

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

NS = load_golden("inflate_ns.json")
STREAM = load_golden("inflate_stream.json")
ZL = load_golden("zlib_frames.json")


@pytest.fixture(scope="module")
def eng():
    import decompress_amd
    return decompress_amd.Engine(0)


@pytest.fixture(params=[2, 1], ids=["two-wavefronts", "one-wavefront"])
def eng_ring(eng, request):
    """both forms of the inflate kernel (decoder + copier wavefront per stream, the default; one wavefront doing both)"""
    eng.set_option("inflate_waves", request.param)
    yield eng
    eng.set_option("inflate_waves", 2)


def test_golden_ns(eng_ring):
    """test/test_ns.ml vectors: status, (consumed, written) and bytes."""
    res = eng_ring.inflate_many([bytes.fromhex(c["src"]) for c in NS], [c["dst_cap"] for c in NS])
    for c, (st, used, out, _) in zip(NS, res):
        assert st == c["status"], (c["name"], st)
        if st == 0:
            assert used == c["consumed"], c["name"]
            assert len(out) == c["written"], c["name"]
            if "dst" in c:
                assert out == bytes.fromhex(c["dst"]), c["name"]


def test_golden_stream(eng_ring):
    eng = eng_ring
    res = eng.inflate_many([bytes.fromhex(c["src"]) for c in STREAM], [65536] * len(STREAM))
    for c, (st, _, out, _) in zip(STREAM, res):
        assert st == c["status"], c["name"]
        if st == 0:
            assert out == bytes.fromhex(c["dst"]), c["name"]


def test_golden_zlib(eng):
    import decompress_amd
    for c in ZL:
        src = bytes.fromhex(c["src"])
        st, used, out, adler = eng.inflate_many([src], [65536], decompress_amd.FORMAT_ZLIB)[0]
        assert (st, used) == (0, len(src))
        assert out == zlib.decompress(src)
        assert adler == zlib.adler32(out)


def _mk(rng, n, kind):
    if kind == "text":
        words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 9))) for _ in range(300)]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words) + b" "
        return bytes(out[:n])
    if kind == "rand":
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(256)]) * rng.randrange(1, 700)
        return bytes(out[:n])
    if kind == "far":  # long-distance matches: repeat 20-30 KiB apart
        base = bytes(rng.getrandbits(8) for _ in range(30000))
        out = bytearray()
        while len(out) < n:
            out += base[: rng.randrange(20000, 30000)]
        return bytes(out[:n])
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["text", "rand", "runs", "far"])
def test_vs_oracle_raw(eng_ring, oracle, kind):
    rng = random.Random(hash(kind) & 0xffff)
    srcs, caps, plains = [], [], []
    for n in (0, 1, 2, 63, 64, 65, 1023, 1024, 1025, 4095, 4097, 40000, 70001, 300000):
        data = _mk(rng, n, kind)
        for level, strat in ((1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (0, 0)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            srcs.append(co.compress(data) + co.flush())
            caps.append(n + rng.randrange(0, 3))
            plains.append(data)
    res = eng_ring.inflate_many(srcs, caps)
    for src, cap, plain, (st, used, out, adler) in zip(srcs, caps, plains, res):
        ost, oused, oout = oracle.de_inflate(src, cap)
        assert (st, used) == (ost, oused)
        assert out == oout == plain
        assert adler == zlib.adler32(plain)


def test_vs_oracle_zlib_batch(eng, oracle):
    import decompress_amd
    from decompress_amd import workloads
    streams = workloads.c2_streams(24, nbytes=256 * 1024, workers=0)
    res = eng.inflate_many(streams, [256 * 1024] * len(streams), decompress_amd.FORMAT_ZLIB)
    for z, (st, used, out, adler) in zip(streams, res):
        assert (st, used) == (0, len(z))
        assert out == zlib.decompress(z)
        assert adler == zlib.adler32(out)


def test_errors_match_oracle(eng_ring, oracle):
    eng = eng_ring
    """Truncations, bit flips, short outputs: same status / same bytes as the oracle."""
    import decompress_amd
    rng = random.Random(99)
    data = _mk(rng, 20000, "text")
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    srcs, caps = [], []
    for cut in list(range(0, 40)) + list(range(40, len(raw), 97)):
        srcs.append(raw[:cut]); caps.append(30000)
    for k in range(200):
        b = bytearray(raw)
        i = rng.randrange(len(b))
        b[i] ^= 1 << rng.randrange(8)
        srcs.append(bytes(b)); caps.append(30000)
    for cap in (0, 1, 100, 19999, 20000):
        srcs.append(raw); caps.append(cap)
    res = eng.inflate_many(srcs, caps)
    for src, cap, (st, used, out, _) in zip(srcs, caps, res):
        ost, oused, oout = oracle.de_inflate(src, cap)
        assert st == ost, (len(src), cap, st, ost)
        assert used == oused
        assert out == oout
    z = zlib.compress(data)
    bad = [z[:-1] + bytes([z[-1] ^ 1]), b"\x79\x9c" + z[2:], b"\x78", z[:5], z]
    res = eng.inflate_many(bad, [len(data)] * len(bad), decompress_amd.FORMAT_ZLIB)
    for src, (st, used, out, _) in zip(bad, res):
        ost, oused, oout = oracle.zl_inflate(src, len(data))
        assert (st, used) == (ost, oused)


def test_hand_over_between_block_kinds(eng_ring, oracle):
    """Streams whose blocks change kind all the time — stored, fixed, dynamic, empty (sync-flush) blocks of every size
    around the staging buffer — with room for all of the output, for a part of it, and cut short: the decoder and the
    copier wavefront hand over Huffman rounds and stored bytes in every order.  Same status / count / bytes as the oracle."""
    rng = random.Random(0x2a7e)
    sizes = [0, 1, 5, 64, 700, 5551, 5552, 5553, 5568, 6200, 11104, 20000, 65535, 65536, 70000]
    kinds = ["text", "rand", "runs", "far"]
    srcs, caps, plains = [], [], []
    for case in range(48):
        body, plain = bytearray(), bytearray()
        nseg = rng.randrange(2, 9)
        for k in range(nseg):
            data = _mk(rng, rng.choice(sizes), rng.choice(kinds))
            level, strat = rng.choice(((0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY)))
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            # every segment but the last ends on a byte boundary with non-final blocks; a segment starts without history
            body += co.compress(data) + (co.flush() if k == nseg - 1 else co.flush(zlib.Z_SYNC_FLUSH))
            plain += data
        raw, plain = bytes(body), bytes(plain)
        assert zlib.decompress(raw, -15) == plain
        for cap in (len(plain), len(plain) + 7, len(plain) // 2, max(0, len(plain) - 1)):
            srcs.append(raw); caps.append(cap); plains.append(plain)
        cut = rng.randrange(1, len(raw))
        srcs.append(raw[:cut]); caps.append(len(plain)); plains.append(plain)
    res = eng_ring.inflate_many(srcs, caps)
    for src, cap, plain, (st, used, out, _) in zip(srcs, caps, plains, res):
        ost, oused, oout = oracle.de_inflate(src, cap)
        assert (st, used) == (ost, oused), (len(src), cap, st, ost)
        assert out == oout
        if st == 0:
            assert out == plain


def test_runs_of_tiny_blocks(eng_ring, oracle):
    """Runs of empty fixed blocks (zlib's Z_PARTIAL_FLUSH: header + end-of-block code, ten bits) and empty stored blocks
    (Z_SYNC_FLUSH) between small data blocks: the kernel walks such runs through the window it has in LDS instead of
    spending a round on each - whole, cut at every byte of a run, and with too little output room: status / count / bytes
    as the oracle's."""
    rng = random.Random(0xe3b)
    srcs, caps = [], []
    for case in range(24):
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, rng.choice((0, zlib.Z_FIXED)))
        body, plain = bytearray(), bytearray()
        for k in range(rng.randrange(2, 12)):
            data = _mk(rng, rng.choice((0, 0, 1, 40, 300, 6000)), rng.choice(("text", "runs")))
            body += co.compress(data)
            plain += data
            for _ in range(rng.choice((1, 1, 3, 40, 700))):
                body += co.flush(rng.choice((zlib.Z_PARTIAL_FLUSH, zlib.Z_PARTIAL_FLUSH, zlib.Z_SYNC_FLUSH)))
        body += co.flush()
        raw, plain = bytes(body), bytes(plain)
        assert zlib.decompress(raw, -15) == plain
        srcs += [raw, raw, raw]
        caps += [len(plain), len(plain) + 3, len(plain) // 2]
        for cut in sorted({rng.randrange(1, len(raw)) for _ in range(12)} | {len(raw) - 1, len(raw) - 2}):
            srcs.append(raw[:cut])
            caps.append(len(plain))
    # ten-bit blocks by hand: a long run that ends the stream, one that ends without a final block, one followed by noise
    def bits_to_bytes(bits):
        b = bytearray((len(bits) + 7) // 8)
        for i, v in enumerate(bits):
            b[i >> 3] |= v << (i & 7)
        return bytes(b)
    empty, final = [0, 1, 0] + [0] * 7, [1, 1, 0] + [0] * 7
    for run in (1, 2, 3, 100, 1700, 1800, 5000):
        srcs += [bits_to_bytes(empty * run + final), bits_to_bytes(empty * run), bits_to_bytes(empty * run + [0, 1, 1]),
                 bits_to_bytes(empty * run + [0, 0, 0]) + b"\x05\x00\xfa\xffhello" + bits_to_bytes(final)]
        caps += [16, 16, 16, 16]
    res = eng_ring.inflate_many(srcs, caps)
    for src, cap, (st, used, out, _) in zip(srcs, caps, res):
        ost, oused, oout = oracle.de_inflate(src, cap)
        assert (st, used, out) == (ost, oused, oout), (len(src), cap, st, ost)


def test_stored_config1(eng_ring, oracle):
    eng = eng_ring
    """BASELINE config 1: 64 KiB of stored blocks (65535 + 1)."""
    from decompress_amd import workloads
    payload = np.random.default_rng(1).integers(0, 256, 65536, dtype=np.uint8).tobytes()
    s = workloads.stored_stream(payload)
    assert len(s) == 65546
    st, used, out, _ = eng.inflate_many([s], [65536])[0]
    assert (st, used, out) == (0, len(s), payload)
    assert oracle.de_inflate(s, 65536) == (0, len(s), payload)


def test_c_abi_single_stream(eng):
    """md_de_inf_ns_inflate / md_zl_inf_ns_inflate with host pointers."""
    from decompress_amd import _lib
    lib = _lib.load()
    ctx = lib.md_create(0, None)
    assert ctx
    data = b"De.Inf.Ns.inflate " * 500
    z = zlib.compress(data, 6)
    dst = ctypes.create_string_buffer(len(data))
    used, written = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib.md_zl_inf_ns_inflate(ctx, z, len(z), dst, len(data), ctypes.byref(used), ctypes.byref(written))
    assert (rc, used.value, written.value) == (0, len(z), len(data))
    assert dst.raw == data
    rc = lib.md_de_inf_ns_inflate(ctx, z[2:-4], len(z) - 6, dst, 10, ctypes.byref(used), ctypes.byref(written))
    assert rc == 2  # Unexpected_end_of_output
    lib.md_destroy(ctx)


def test_python_mirror(eng):
    from decompress_amd import de, zl
    assert de.Inf.Ns.inflate(b"\x01\x04\x00\xfb\xff\xde\xad\xbe\xef", 65536) == ("Ok", (9, 4), b"\xde\xad\xbe\xef")
    assert de.Inf.Ns.inflate(b"\x06", 65536) == ("Error", "Invalid_kind_of_block")
    z = zlib.compress(b"abc" * 100)
    assert zl.Inf.Ns.inflate(z, 300) == ("Ok", (len(z), 300), b"abc" * 100)


def test_large_streams(eng):
    """a multi-MiB stream (hundreds of rounds, dozens of blocks) through the host and the device entry points"""
    import ctypes
    import decompress_amd
    from decompress_amd import workloads
    data = workloads.text(77, 6 * 1024 * 1024)
    z = zlib.compress(data, 6)
    dst = ctypes.create_string_buffer(len(data))
    used, wrote = ctypes.c_size_t(), ctypes.c_size_t()
    st = eng.lib.md_zl_inf_ns_inflate(eng.ctx, z, len(z), dst, len(data), ctypes.byref(used), ctypes.byref(wrote))
    assert (st, used.value, wrote.value) == (0, len(z), len(data)) and dst.raw == data
    res = eng.inflate_many([z, z[:len(z) // 2]], [len(data), len(data)], decompress_amd.FORMAT_ZLIB)
    assert res[0][0] == 0 and res[0][2] == data and res[0][3] == zlib.adler32(data)
    assert res[1][0] == 1  # Unexpected end of input


def _par_last(eng):
    v = eng.lib.md_set_option(eng.ctx, b"inflate_parallel_last", 0)
    return v & 0xffffff, v >> 24  # pieces (0: the serial path took the stream), decode rounds


def _higher(eng, fmt, z, cap):
    import ctypes
    dst = ctypes.create_string_buffer(max(cap, 1))
    used, wrote = ctypes.c_size_t(), ctypes.c_size_t()
    if fmt == "zl":
        st = eng.lib.md_zl_higher_uncompress(eng.ctx, z, len(z), dst, cap, ctypes.byref(wrote))
    elif fmt == "de":
        st = eng.lib.md_de_higher_uncompress(eng.ctx, z, len(z), dst, cap, ctypes.byref(wrote))
    elif fmt == "zlns":
        st = eng.lib.md_zl_inf_ns_inflate(eng.ctx, z, len(z), dst, cap, ctypes.byref(used), ctypes.byref(wrote))
    else:
        st = eng.lib.md_gz_higher_uncompress(eng.ctx, z, len(z), dst, cap, ctypes.byref(used), ctypes.byref(wrote), None)
    return st, used.value, dst.raw[:wrote.value]


def test_one_long_stream_by_the_whole_chip(eng):
    """De / Zl / Gz.Higher.uncompress on ONE 64 MiB stream (lib/de.ml:4555-4571, lib/zl.ml:650-666; VERDICT r5 missing 1):
    decoded in pieces from candidate block starts with placeholder windows, resolved afterwards (csrc/inflate_chunked.hip) -
    the bytes are libz's, the pieces are many, and it is not the 130-160 MiB/s of one pair of wavefronts any more."""
    import time
    from decompress_amd import workloads
    data = workloads.text(77, 64 << 20)
    z = zlib.compress(data, 6)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(data) + co.flush()
    for fmt, src in (("zl", z), ("zlns", z), ("de", raw), ("gz", gz)):
        t0 = time.perf_counter()
        st, used, out = _higher(eng, fmt, src, len(data))
        dt = time.perf_counter() - t0
        assert st == 0 and out == data, fmt
        if fmt in ("zlns", "gz"):
            assert used == len(src)
        pieces, rounds = _par_last(eng)
        assert pieces > 100 and rounds == 1, (fmt, pieces, rounds)
        if fmt != "zl":  # (the first call allocates the context's scratch; host to host from pageable memory: 4.3 GiB/s measured)
            assert len(data) / dt > 0.5 * 2**30, (fmt, dt)
    # what is not a well-formed stream that fits is the serial path's: same statuses as with the parallel path off
    bad = bytearray(z)
    bad[-1] ^= 1
    cases = [(bytes(bad), len(data)), (z[:len(z) // 2], len(data)), (z, len(data) - 5), (z + b"trailing", len(data))]
    got = [_higher(eng, "zlns", c, cap) for c, cap in cases[:3]] + [_higher(eng, "zlns", *cases[3])]
    assert _par_last(eng)[0] > 100  # (bytes behind the stream are not its business)
    eng.set_option("inflate_parallel_min", 0)
    try:
        want = [_higher(eng, "zlns", c, cap) for c, cap in cases]
        assert _par_last(eng)[0] == 0
    finally:
        eng.set_option("inflate_parallel_min", 96)
    assert [g[:2] for g in got] == [w[:2] for w in want] and [g[0] for g in got] == [9, 1, 2, 0]
    assert got[3][2] == data and got[3][1] == len(z)


def test_long_pieces_of_the_decode_protocol(eng):
    """De.Inf.decode on a long stream (lib/de.ml:1427-1474, the tool's loop bin/decompress.ml:77-100): a piece of the
    protocol that is long enough is decoded by the whole chip up to its last block boundary, the incomplete block behind
    it by the serial path (capi.cpp continue_parallel) - same bytes, same signals at the end, Adler-32 and CRC-32 carried
    across the pieces; a stream cut inside a block still ends in the serial path's `Malformed"""
    import decompress_amd
    from decompress_amd import de, engine, workloads
    eng = engine.default_engine(0)  # (the context de.Inf drives)
    plain = workloads.text(0x52, 24 << 20)
    z = zlib.compress(plain, 6)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(plain) + co.flush()
    for fmt, src in ((decompress_amd.FORMAT_ZLIB, z), (decompress_amd.FORMAT_GZIP, gz), (decompress_amd.FORMAT_DEFLATE, z[2:-4])):
        for chunk, step in ((3 << 20, 700001), (8 << 20, len(src))):
            verdict, out, sigs = de.Inf.decode_chunks((src[i:i + step] for i in range(0, len(src), step)), o_len=1 << 20, fmt=fmt,
                                                      chunk_bytes=chunk)
            assert verdict == "Ok" and out == plain, (fmt, chunk)
            assert _par_last(eng)[0] > 0 or chunk < (8 << 20)
    # the whole input at once: every block but the last few by the pieces
    verdict, out, _ = de.Inf.decode_chunks([z], o_len=65536, fmt=decompress_amd.FORMAT_ZLIB)
    assert verdict == "Ok" and out == plain and _par_last(eng)[0] > 50
    # cut inside a block, a damaged trailer - at once and in pieces: what the serial path says, with everything decoded before
    # the error handed out
    feeds = {"cut": z[:len(z) // 2], "bad": z[:-1] + bytes([z[-1] ^ 1])}
    steps = lambda src: [src[i:i + (1 << 20)] for i in range(0, len(src), 1 << 20)]
    eng.set_option("inflate_parallel_min", 0)
    try:
        want = {k: (de.Inf.decode_chunks([v], o_len=65536, fmt=decompress_amd.FORMAT_ZLIB)[:2],
                    de.Inf.decode_chunks(steps(v), o_len=65536, fmt=decompress_amd.FORMAT_ZLIB, chunk_bytes=3 << 20)[:2]) for k, v in feeds.items()}
    finally:
        eng.set_option("inflate_parallel_min", 96)
    for k, v in feeds.items():
        got = (de.Inf.decode_chunks([v], o_len=65536, fmt=decompress_amd.FORMAT_ZLIB)[:2],
               de.Inf.decode_chunks(steps(v), o_len=65536, fmt=decompress_amd.FORMAT_ZLIB, chunk_bytes=3 << 20)[:2])
        assert got == want[k] and got[0][0] != "Ok" and got[0][0] == got[1][0], k


def test_a_few_long_streams_in_one_host_batch(eng, oracle):
    """md_inflate_batch_host with a handful of big streams: each long one is decoded by the whole chip, the short and the
    broken ones beside them through the batch kernel - every result as the one-pair-of-wavefronts path gives it"""
    import decompress_amd
    from decompress_amd import workloads
    plains = [workloads.text(700 + i, (6 + i) << 20) for i in range(3)] + [workloads.text(9, 5000), workloads.markov_text(3, 3 << 20)]
    streams = [zlib.compress(p, 6) for p in plains]
    streams.append(streams[0][:len(streams[0]) // 2])          # cut: Unexpected end of input
    streams.append(streams[1][:-3] + b"\x00\x00\x00")           # a wrong checksum
    caps = [len(p) for p in plains] + [len(plains[0]), len(plains[1])]
    n = len(streams)
    in_len = np.array([len(z) for z in streams], dtype=np.uint64)
    in_off = np.zeros(n, dtype=np.uint64)
    in_off[1:] = np.cumsum((in_len + 15) // 16 * 16)[:-1]
    cap = np.array(caps, dtype=np.uint64)
    out_off = np.zeros(n, dtype=np.uint64)
    out_off[1:] = np.cumsum((cap + 255) // 256 * 256)[:-1]
    h_in, h_out = eng.host_buffer(int(in_off[-1] + in_len[-1]) + 64), eng.host_buffer(int(out_off[-1] + cap[-1]) + 64)
    for z, o in zip(streams, in_off):
        h_in[int(o):int(o) + len(z)] = np.frombuffer(z, dtype=np.uint8)
    got = {}
    for par in (512, 0):
        eng.set_option("inflate_parallel_min", par)
        h_out[:] = 0
        r = eng.inflate_batch_host(decompress_amd.FORMAT_ZLIB, h_in, in_off, in_len, h_out, out_off, cap)
        got[par] = ([a.copy() for a in r], [h_out[int(out_off[i]):int(out_off[i]) + int(r[0][i])].tobytes() for i in range(n)], _par_last(eng)[0])
    eng.set_option("inflate_parallel_min", 96)
    assert got[512][2] > 100 and got[0][2] == 0
    out_len, consumed, status, checksum = got[512][0]
    assert list(status) == list(got[0][0][2]) == [0, 0, 0, 0, 0, 1, 9]
    for i in range(5):
        assert got[512][1][i] == plains[i] and consumed[i] == len(streams[i]) and checksum[i] == zlib.adler32(plains[i])
    ok = got[0][0][2] == 0
    for a, b in zip(got[512][0][:2], got[0][0][:2]):
        assert (a[ok] == b[ok]).all()
    assert got[512][1][5] == got[0][1][5] and got[512][1][6] == got[0][1][6]  # what was decoded in front of the error


def test_long_stream_kinds_and_false_candidates(eng):
    """other data through the pieces: flush markers (pigz-like), levels, incompressible stretches in stored blocks, and a
    stream whose stored blocks CONTAIN valid block headers (compressed data inside the plaintext): candidates that are
    no block starts are found out by the chain of pieces and dropped"""
    from decompress_amd import workloads
    t = workloads.text(5, 6 << 20)
    inner = zlib.compress(workloads.text(6, 8 << 20), 6)  # ~3 MiB of DEFLATE data: incompressible, full of dynamic headers
    eng.set_option("inflate_parallel_chunk", 16)
    try:
        mixed = t[:2 << 20] + inner + t[2 << 20:4 << 20] + workloads.ascii_uniform(9, 1 << 20) + inner[::-1] + t[4 << 20:]
        z = zlib.compress(mixed, 6)
        st, _, out = _higher(eng, "zl", z, len(mixed))
        pieces, rounds = _par_last(eng)
        assert st == 0 and out == mixed and pieces > 50 and rounds >= 2, (st, pieces, rounds)
        for lvl in (1, 9):
            z = zlib.compress(t, lvl)
            st, _, out = _higher(eng, "zl", z, len(t))
            assert st == 0 and out == t and _par_last(eng)[0] > 20
        # stored blocks need no finder: their starts are read off the stream (level 0; noise, which zlib stores)
        noise = np.random.default_rng(4).integers(0, 256, size=5 << 20, dtype=np.uint8).tobytes()
        for data, lvl in ((t, 0), (noise, 6), (t[:1 << 20] + noise + t[1 << 20:], 6)):
            z = zlib.compress(data, lvl)
            st, _, out = _higher(eng, "zl", z, len(data))
            assert st == 0 and out == data and _par_last(eng)[0] > 50, (lvl, _par_last(eng))
        co = zlib.compressobj(6)
        parts = []
        for i in range(0, len(t), 30011):
            parts.append(co.compress(t[i:i + 30011]))
            parts.append(co.flush(zlib.Z_FULL_FLUSH if i % 2 else zlib.Z_SYNC_FLUSH))
        parts.append(co.flush())
        st, _, out = _higher(eng, "zl", b"".join(parts), len(t))
        assert st == 0 and out == t and _par_last(eng)[0] > 20
        # tiny fixed-Huffman units between flush markers: the markers are the only candidates
        co = zlib.compressobj(6)
        parts = []
        small = t[:600000]
        for i in range(0, len(small), 40):
            parts.append(co.compress(small[i:i + 40]))
            parts.append(co.flush(zlib.Z_FULL_FLUSH))
        parts.append(co.flush())
        st, _, out = _higher(eng, "zl", b"".join(parts), len(small))
        assert st == 0 and out == small and _par_last(eng)[0] > 20
        # a reference in front of the start of the stream, far behind a block boundary: Invalid distance from the serial path
        # (raw stream: [stored 40000 zeros][dynamic blocks of text whose matches are fine]...) - the corrupted case: drop
        # the first 100 KiB of plaintext from a stream by decoding from a later block start is not expressible with libz;
        # the device-side check is exercised by the fuzz tests' short streams instead.
    finally:
        eng.set_option("inflate_parallel_chunk", 64)


def test_default_stream_ordering(eng):
    """Engine() enqueues on torch's current (default) stream: a torch kernel that produces the input and one that
    consumes the output need no device-wide synchronisation around the batch call."""
    import torch
    import decompress_amd
    from decompress_amd import workloads
    streams = workloads.c2_streams(64, nbytes=64 * 1024, workers=0)
    blob, in_off, in_len = workloads.pack(streams)
    dev = eng.device
    staged = torch.from_numpy(blob).to(dev)
    n, nb = len(streams), 64 * 1024
    t = lambda a: torch.from_numpy(a).to(dev)
    out_off = t(np.arange(n, dtype=np.int64) * nb)
    out_cap = t(np.full(n, nb, dtype=np.int64))
    d_in_off, d_in_len = t(in_off), t(in_len)
    for _ in range(3):
        d_in = torch.zeros_like(staged)
        d_in += staged  # produced by a torch kernel on the current stream, no sync
        d_out = torch.empty(n * nb, dtype=torch.uint8, device=dev)
        res = eng.inflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_in_off, d_in_len, d_out, out_off, out_cap)
        total = d_out.to(torch.int64).sum()  # consumed by a torch kernel, no sync in between
        bad = (res[2] != 0).sum()
        want = sum(sum(zlib.decompress(z)) for z in streams)
        assert int(bad.item()) == 0 and int(total.item()) == want


def test_oversize_descriptor_is_rejected(eng):
    """lengths beyond the 32-bit cursors: a call-level error from the host entry point, status -1 per stream from
    the device entry point (mdeflate.h limits)"""
    import torch
    import decompress_amd
    dev = eng.device
    d_in = torch.zeros(64, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(64, dtype=torch.uint8, device=dev)
    t = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
    res = eng.inflate_batch(decompress_amd.FORMAT_DEFLATE, d_in, t([0, 0]), t([1 << 30, 5]), d_out, t([0, 32]), t([32, 32]))
    torch.cuda.synchronize(dev)
    assert int(res[2][0].item()) == -1 and int(res[0][0].item()) == 0
    assert int(res[2][1].item()) != -1


def test_streaming_decoder_shim(eng, oracle):
    """De.Inf.decoder / src / decode / flush / dst_rem (lib/de.mli:82-144) above the batch ABI: chunks in,
    `Await until the end of input, then `Flush per output buffer and `End (or `Malformed after the bytes that were
    decoded before the error)"""
    import decompress_amd
    from decompress_amd import de, workloads
    data = workloads.text(31, 300000)
    z = zlib.compress(data, 6)
    chunks = [z[i:i + 7000] for i in range(0, len(z), 7000)]
    verdict, out, sigs = de.Inf.decode_chunks(chunks, o_len=65536, fmt=decompress_amd.FORMAT_ZLIB)
    assert (verdict, out) == ("Ok", data)
    assert sigs.count(de.AWAIT) == len(chunks) + 1 and sigs.count(de.FLUSH) == len(data) // 65536 and sigs[-1] == de.END
    raw = z[2:-4]
    verdict, out, sigs = de.Inf.decode_chunks([raw[:5000], raw[5000:20000]], o_len=4096)
    ost, _, oout = oracle.de_inflate(raw[:20000], 1 << 22)
    assert verdict == "Unexpected_end_of_input" and ost == 1 and out == oout and sigs[-1] == de.MALFORMED
    assert de.Inf.decode_chunks([], o_len=16)[0] == "Unexpected_end_of_input"
    assert de.Inf.decode_chunks([b"\x03\x00"], o_len=16)[:2] == ("Ok", b"")


def _continue(eng, src, start_bit, hist, cap, adler, flags=1):
    """md_de_inf_continue_host on one piece (with the CRC-32 of its new output) -> (status, output position, new bytes, resume)"""
    from decompress_amd import _lib
    dst = ctypes.create_string_buffer(bytes(hist), max(1, cap))
    n, st, rs = ctypes.c_size_t(), ctypes.c_int(), _lib.InfResume()
    eng._check(eng.lib.md_de_inf_continue_host(eng.ctx, bytes(src), len(src), start_bit, dst, len(hist), cap, adler, flags,
                                               ctypes.byref(n), ctypes.byref(st), ctypes.byref(rs)))
    return st.value, n.value, dst.raw[len(hist):n.value], rs


def test_stream_decoded_in_pieces_by_hand(eng_ring):
    """md_de_inf_continue_host: a raw stream cut at arbitrary bytes is decoded piece by piece — every piece starts at the
    block boundary (a bit position) the last one reported, with the last 32 KiB of output in front of its buffer and the
    checksum state handed on — and the pieces add up to the stream: bytes, Adler-32, where it ends."""
    eng = eng_ring
    rng = random.Random(0x51ce)
    for kind, level, strat in (("text", 6, 0), ("far", 6, 0), ("runs", 9, 0), ("rand", 0, 0), ("text", 6, zlib.Z_FIXED)):
        plain = _mk(rng, 600000, kind)
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
        raw = co.compress(plain) + co.flush() + b"trailing"
        got, hist, adler, pos, bit, pieces = bytearray(), b"", 1, 0, 0, 0
        step = max(200, len(raw) // 7)
        end = step
        while True:
            piece = raw[pos:min(end, len(raw))]
            st, n, new, rs = _continue(eng, piece, bit, hist, len(hist) + 700000, adler)
            pieces += 1
            if st == 0:
                got += new
                assert rs.last == 1 and rs.out == n
                assert rs.crc_end == rs.crc_out == zlib.crc32(new)
                assert zlib.adler32(bytes(got)) == rs.checksum == rs.adler
                assert pos + rs.consumed == len(raw) - len(b"trailing")
                break
            assert st == 1, st  # MD_UNEXPECTED_END_OF_INPUT: the piece ended inside a block
            assert end < len(raw), "the whole stream was there"
            upto = rs.out - len(hist)
            assert new[:upto] == plain[len(got):len(got) + upto]  # (what follows belongs to the incomplete block: valid too)
            assert new == plain[len(got):len(got) + len(new)]
            assert rs.crc_out == zlib.crc32(new[:upto]) and rs.crc_end == zlib.crc32(new)
            got += new[:upto]
            assert rs.adler == zlib.adler32(bytes(got))
            hist = bytes(got[-32768:])
            adler = rs.adler
            pos += rs.bits >> 3
            bit = rs.bits & 7
            end += step
        assert bytes(got) == plain and pieces >= 3


def test_batch_of_streams_in_two_pieces(eng):
    """md_inflate_continue_batch_device: 300 streams of different kinds, each cut somewhere in the middle — the first
    launch decodes every first half up to its last block boundary, the second goes on from there with the window in
    front of its output; together they are the streams (bytes and Adler-32), all 300 in two launches."""
    import torch
    rng = random.Random(0xba7c)
    n = 300
    plains, raws = [], []
    for i in range(n):
        data = _mk(rng, rng.choice((3000, 70000, 200000)), rng.choice(("text", "far", "runs", "rand")))
        co = zlib.compressobj(rng.choice((0, 1, 6, 9)), zlib.DEFLATED, -15, 8, rng.choice((0, 0, zlib.Z_FIXED)))
        plains.append(data)
        raws.append(co.compress(data) + co.flush())
    dev = eng.device
    t = lambda a, dt: torch.tensor(a, dtype=dt, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())

    def launch(pieces, start_bits, hists, adlers, cap):
        blob = b"".join(x + b"\0" * (-len(x) % 16) for x in pieces)
        offs = np.cumsum([0] + [len(x) + (-len(x) % 16) for x in pieces[:-1]]).astype(np.int64)
        d_in = torch.from_numpy(np.frombuffer(blob + b"\0" * 16, dtype=np.uint8).copy()).to(dev)
        d_out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
        for i, h in enumerate(hists):
            if h:
                d_out[i * cap:i * cap + len(h)] = torch.from_numpy(np.frombuffer(h, dtype=np.uint8).copy()).to(dev)
        d_in_off, d_in_len = t(offs, torch.int64), t([len(x) for x in pieces], torch.int64)
        d_out_off, d_out_cap = t([i * cap for i in range(n)], torch.int64), t([cap] * n, torch.int64)
        d_bit, d_hist, d_adl = t(start_bits, torch.int32), t([len(h) for h in hists], torch.int32), t([a - (1 << 32) if a >= 1 << 31 else a for a in adlers], torch.int32)
        out_len, consumed = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
        status, checksum = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        r_bits, r_out = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
        r_adl, r_last = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        eng._check(eng.lib.md_inflate_continue_batch_device(eng.ctx, n, p(d_in), p(d_in_off), p(d_in_len), p(d_out), p(d_out_off), p(d_out_cap),
                                                            p(d_bit), p(d_hist), p(d_adl), p(out_len), p(consumed), p(status), p(checksum),
                                                            p(r_bits), p(r_out), p(r_adl), p(r_last)))
        torch.cuda.synchronize()
        u32 = lambda x: [v & 0xffffffff for v in x.cpu().tolist()]
        return (d_out.cpu().numpy(), status.cpu().tolist(), out_len.cpu().tolist(), r_bits.cpu().tolist(), r_out.cpu().tolist(),
                u32(r_adl), r_last.cpu().tolist(), u32(checksum))

    cap = 32768 + 210000
    cuts = [rng.randrange(1, len(r)) for r in raws]
    out1, st1, len1, bits1, ro1, adl1, last1, _ = launch([r[:c] for r, c in zip(raws, cuts)], [0] * n, [b""] * n, [1] * n, cap)
    firsts, hists, rest, sbits = [], [], [], []
    for i in range(n):
        assert st1[i] in (0, 1), (i, st1[i])  # complete already (the cut fell behind the final block's last bit) or cut inside a block
        upto = len1[i] if st1[i] == 0 else ro1[i]
        firsts.append(out1[i * cap:i * cap + upto].tobytes())
        assert plains[i].startswith(firsts[i])
        assert adl1[i] == zlib.adler32(firsts[i]) or st1[i] == 0
        hists.append(firsts[i][-32768:])
        rest.append(raws[i][bits1[i] >> 3:] if st1[i] == 1 else b"\x03\x00")  # (finished streams get an empty final block)
        sbits.append(bits1[i] & 7 if st1[i] == 1 else 0)
    adlers = [adl1[i] if st1[i] == 1 else zlib.adler32(firsts[i]) for i in range(n)]
    out2, st2, len2, _, _, _, last2, sum2 = launch(rest, sbits, hists, adlers, cap)
    for i in range(n):
        assert st2[i] == 0 and last2[i] == 1, (i, st2[i])
        whole = firsts[i] + out2[i * cap + len(hists[i]):i * cap + len2[i]].tobytes()
        assert whole == plains[i], (i, len(whole), len(plains[i]))
        assert sum2[i] == zlib.adler32(plains[i])


def test_streaming_decoder_hands_out_before_the_end(eng, oracle):
    """The De.Inf.decode protocol on streams longer than a piece: output comes out through `Flush while input is still
    being supplied, the result is the whole-buffer one (bytes, checksum, status, message, unread input) — for raw and
    ZLIB streams, intact, corrupted in the middle, cut short, with a wrong checksum."""
    import decompress_amd
    from decompress_amd import de
    rng = random.Random(0x57e)
    plain = _mk(rng, 900000, "text") + _mk(rng, 300000, "far") + _mk(rng, 200000, "runs")
    z = zlib.compress(plain, 6)
    raw = z[2:-4]
    cases = [
        ("raw", decompress_amd.FORMAT_DEFLATE, raw + b"xyz", 0, 3),
        ("zlib", decompress_amd.FORMAT_ZLIB, z, 0, 0),
        ("zlib-checksum", decompress_amd.FORMAT_ZLIB, z[:-1] + bytes([z[-1] ^ 1]), 9, 0),
        ("zlib-cut", decompress_amd.FORMAT_ZLIB, z[:len(z) // 2], 1, 0),
        ("raw-flipped", decompress_amd.FORMAT_DEFLATE, raw[:len(raw) // 2] + bytes([raw[len(raw) // 2] ^ 0x10]) + raw[len(raw) // 2 + 1:], None, None),
    ]
    # GZip members: header with FEXTRA (big-endian length, as the reference reads it), FNAME, FCOMMENT and FHCRC
    def gz_member(body, data, crc_xor=0, size_add=0, hcrc_xor=0):
        fixed = bytes([0x1f, 0x8b, 8, 2 | 4 | 8 | 16]) + b"\0\0\0\0" + bytes([0, 3])
        extra, name, comment = b"\x00\x05hello", b"a name\0", b"a comment\0"
        hcrc = ((zlib.crc32(fixed + name + comment) >> 16) & 0xffff) ^ hcrc_xor
        head = fixed + extra + name + comment + bytes([hcrc >> 8, hcrc & 0xff])
        tail = ((zlib.crc32(data) ^ crc_xor) & 0xffffffff).to_bytes(4, "little") + ((len(data) + size_add) & 0xffffffff).to_bytes(4, "little")
        return head + body + tail
    cases += [
        ("gzip", decompress_amd.FORMAT_GZIP, gz_member(raw, plain) + b"rest", 0, 4),
        ("gzip-crc", decompress_amd.FORMAT_GZIP, gz_member(raw, plain, crc_xor=1), 9, 0),
        ("gzip-size", decompress_amd.FORMAT_GZIP, gz_member(raw, plain, size_add=1), 12, 0),
        ("gzip-hcrc", decompress_amd.FORMAT_GZIP, gz_member(raw, plain, hcrc_xor=1), 11, 0),
        ("gzip-cut", decompress_amd.FORMAT_GZIP, gz_member(raw, plain)[:len(raw) // 2], 1, 0),
    ]
    for name, fmt, src, want_status, want_rem in cases:
        for chunk in (4096, 150000):
            pieces = [src[i:i + 50000] for i in range(0, len(src), 50000)]
            state = {"fed": 0, "out_before_end": 0}

            def feed():
                for pc in pieces:
                    state["fed"] += 1
                    yield pc
            eng_ = eng
            lib = eng_.lib
            o = ctypes.create_string_buffer(65536)
            d = lib.md_inf_decoder(eng_.ctx, fmt, o, 65536)
            lib.md_inf_chunk_bytes(d, chunk)
            out, it = bytearray(), feed()
            try:
                while True:
                    sig = lib.md_inf_decode(d)
                    if sig == de.AWAIT:
                        c = next(it, b"")
                        lib.md_inf_src(d, bytes(c), 0, len(c))
                    else:
                        out += o.raw[:65536 - lib.md_inf_dst_rem(d)]
                        lib.md_inf_flush(d)
                        if sig == de.FLUSH and state["fed"] < len(pieces):
                            state["out_before_end"] += 1
                        if sig in (de.END, de.MALFORMED):
                            st, rem, msg = lib.md_inf_status(d), lib.md_inf_src_rem(d), lib.md_inf_message(d).decode()
                            break
            finally:
                lib.md_inf_free(d)
            # the whole-buffer answer
            if name == "zlib-cut":  # (the streaming decoder takes everything that is there; Zl.Inf.Ns sets the last 4 bytes aside)
                ost, oused, oout = oracle.de_inflate(src[2:], len(plain) + 16)
            elif name == "gzip-cut":
                ost, oused, oout = oracle.de_inflate(src[len(gz_member(b"", b"")) - 8:], len(plain) + 16)
            elif fmt == decompress_amd.FORMAT_GZIP:
                ost, oused, oout, _ = oracle.gz_inflate(src, len(plain) + 16)
                if ost in (9, 12):  # a wrong trailer: everything was inflated and handed out before it was read
                    oout = plain
            elif fmt == decompress_amd.FORMAT_ZLIB:
                ost, oused, oout = oracle.zl_inflate(src, len(plain) + 16)
            else:
                ost, oused, oout = oracle.de_inflate(src, len(plain) + 16)
            assert st == ost, (name, chunk, st, ost)
            assert bytes(out) == oout, (name, chunk, len(out), len(oout))
            if want_status is not None:
                assert st == want_status
            if want_rem is not None and st == 0:
                assert rem == want_rem
            if name in ("zlib-checksum", "gzip-crc"):
                assert msg.startswith("Invalid checksum (expect:") and "has:" in msg
            if name == "gzip-size":
                assert msg == "Invalid input size (expect:%d, inflated:%d)" % (len(plain) + 1, len(plain))
            if name == "gzip-hcrc":
                assert len(out) == 0 and state["out_before_end"] == 0
                continue
            assert state["out_before_end"] > 0, (name, chunk)  # output was handed out while input was still arriving


def test_malformed_strings_and_reset(eng):
    """the `Malformed strings of Zl.Inf / Gz.Inf with their numbers (lib/zl.ml:179-181, lib/gz.ml:287-293), byte for
    byte what the reference's format strings give; De.Inf.reset (lib/de.ml:1512-1532) re-arms the same decoder"""
    import ctypes
    import gzip
    import decompress_amd
    from decompress_amd import de
    data = b"a string that goes through the frame " * 40
    z = bytearray(zlib.compress(data, 6))
    want = int.from_bytes(z[-4:], "big")
    z[-1] ^= 0x5a
    bad = int.from_bytes(z[-4:], "big")
    verdict, out, _ = de.Inf.decode_chunks([bytes(z)], fmt=decompress_amd.FORMAT_ZLIB)
    assert verdict == "Invalid_checksum" and out == data
    assert de.Inf.last_message == "Invalid checksum (expect:%04x, has:%04x)" % (bad, want)
    g = bytearray(gzip.compress(data, mtime=0))
    crc = int.from_bytes(g[-8:-4], "little")
    g[-8] ^= 1
    verdict, out, _ = de.Inf.decode_chunks([bytes(g)], fmt=decompress_amd.FORMAT_GZIP)
    assert verdict == "Invalid_checksum"
    assert de.Inf.last_message == "Invalid checksum (expect:%04x, has:%04x)" % (crc ^ 1, crc)
    g = bytearray(gzip.compress(data, mtime=0))
    g[-1] = 0x80  # ISIZE with the sign bit: %ld prints it as a negative 32-bit number
    verdict, out, _ = de.Inf.decode_chunks([bytes(g)], fmt=decompress_amd.FORMAT_GZIP)
    isize = int.from_bytes(g[-4:], "little") - (1 << 32)
    assert de.Inf.last_message == "Invalid input size (expect:%d, inflated:%d)" % (isize, len(data))
    assert de.Inf.decode_chunks([b"\x00"], fmt=decompress_amd.FORMAT_ZLIB)[0] == "Unexpected_end_of_input"
    assert de.Inf.last_message == "Unexpected end of input"
    # one decoder, two streams
    lib = eng.lib
    o = ctypes.create_string_buffer(1 << 16)
    d = lib.md_inf_decoder(eng.ctx, decompress_amd.FORMAT_ZLIB, o, len(o))
    for payload in (b"first " * 100, b"second stream " * 300):
        zz = zlib.compress(payload)
        assert lib.md_inf_decode(d) == de.AWAIT
        lib.md_inf_src(d, zz, 0, len(zz))
        lib.md_inf_src(d, b"", 0, 0)
        assert lib.md_inf_decode(d) == de.END
        assert o.raw[:len(o) - lib.md_inf_dst_rem(d)] == payload and lib.md_inf_checksum(d) == zlib.adler32(payload)
        lib.md_inf_reset(d)
    lib.md_inf_free(d)
    # a stream at DEFLATE's highest ratio: one launch sized from the 1032x bound, not eight doublings
    big = bytes(4 << 20)
    zz = zlib.compress(big, 9)
    assert len(zz) * 1032 >= len(big)
    verdict, out, sigs = de.Inf.decode_chunks([zz], fmt=decompress_amd.FORMAT_ZLIB, o_len=1 << 20)
    assert verdict == "Ok" and out == big


def test_full_batch_c2(eng):
    """BASELINE config 2 as one batch, as bench.py runs it: 4096 x 256 KiB zlib streams, ALL DISTINCT (even =
    corpus slices, odd = order-2 Markov text), every status, consumed count, length and Adler-32 checked, and EVERY output
    byte compared on the device with libz's plaintexts"""
    import torch
    import decompress_amd
    from decompress_amd import workloads
    n, nb = 4096, 256 * 1024
    streams = workloads.c2_streams(n, nbytes=nb)
    assert len(set(streams[:64])) == 64
    blob, in_off, in_len = workloads.pack(streams)
    dev = eng.device
    t = lambda a: torch.from_numpy(a).to(dev)
    d_out = torch.empty(n * nb, dtype=torch.uint8, device=dev)
    res = eng.inflate_batch(decompress_amd.FORMAT_ZLIB, t(blob), t(in_off), t(in_len), d_out,
                            t(np.arange(n, dtype=np.int64) * nb), t(np.full(n, nb, dtype=np.int64)))
    torch.cuda.synchronize(dev)
    out_len, consumed, status, checksum = [x.cpu().numpy() for x in res]
    assert (status == 0).all() and (out_len == nb).all() and (consumed == in_len).all()
    plains = [zlib.decompress(z) for z in streams]
    want = np.array([zlib.adler32(p) for p in plains], dtype=np.uint32)
    assert (checksum.view(np.uint32) == want).all()
    expect = torch.from_numpy(np.frombuffer(b"".join(plains), dtype=np.uint8).copy()).to(dev)
    assert torch.equal(d_out, expect)


def test_launch_order_of_large_batches(eng, oracle):
    """batches of more streams than the chip holds at once are started longest first (csrc/inflate_wave.hip,
    inflate_order_kernel): 2500 streams of very different lengths, broken and oversize-capacity ones among them,
    every result equals the oracle's — the order changes nothing a caller can see"""
    import decompress_amd
    from decompress_amd import workloads
    rng = random.Random(2500)
    srcs, caps = [], []
    for i in range(2500):
        k = i % 7
        plain = workloads.text(i, rng.randrange(0, 3000)) if k < 4 else bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 600)))
        z = zlib.compress(plain, 1 + i % 9)
        if k == 5:
            z = z[:rng.randrange(0, len(z))]           # truncated
        if k == 6 and len(z) > 8:
            z = z[:6] + bytes([z[6] ^ 0x5a]) + z[7:]   # damaged
        srcs.append(z)
        caps.append(len(plain) + (8 if i % 11 else -3 if len(plain) > 3 else 0))
    res = eng.inflate_many(srcs, caps, decompress_amd.FORMAT_ZLIB)
    for z, cap, (st, used, out, _) in zip(srcs, caps, res):
        ost, oused, oout = oracle.zl_inflate(z, cap)
        assert (st, used) == (ost, oused) and (st != 0 or out == oout)


def test_match_shapes(eng):
    """the copy phases of a round (csrc/inflate_wave.hip: copy_far, copy_near_all, wave_copy) on inputs made for
    them: self-overlapping matches of every small period and of periods around 8, 16, 32 and 258 (the whole wave
    copies those), long far matches, matches that straddle round starts, and text so dense in short matches that a
    round's record pool fills before its staging buffer"""
    import decompress_amd
    rng = random.Random(77)
    rb = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
    plains = []
    for p in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40, 63, 64, 65, 100, 200, 257, 258, 259, 300):
        unit = rb(p)
        plains.append(rb(50) + unit * (4000 // p + 2) + rb(20) + unit * (700 // p + 1))
    block = rb(400)
    plains.append(block + rb(20000) + block + rb(33) + block[:300] + rb(7000) + block[100:])           # long far matches
    plains.append(b"".join(block[i:i + 258] + rb(3) for i in range(0, 140)) * 12)                        # 258-byte matches, near and far
    grams = [rb(3) for _ in range(40)]
    plains.append(b"".join(rng.choice(grams) + bytes([rng.randrange(256)]) for _ in range(30000)))        # a match every 4 bytes
    plains.append(b"".join(rng.choice(grams) for _ in range(40000)))                                      # nothing but matches
    plains.append(bytes(200000))                                                                          # one run
    plains.append((b"ab" * 70 + b"c") * 900)
    srcs, caps = [], []
    for i, pl in enumerate(plains):
        for level in (1, 6, 9):
            srcs.append(zlib.compress(pl, level))
            caps.append(len(pl))
    res = eng.inflate_many(srcs, caps, decompress_amd.FORMAT_ZLIB)
    for k, (st, used, out, adler) in enumerate(res):
        pl = plains[k // 3]
        assert (st, used) == (0, len(srcs[k])), (k, st)
        assert out == pl and adler == zlib.adler32(pl), k


def test_two_contexts_in_two_threads(oracle):
    """two contexts with their own HIP streams, driven from two host threads at once (large batches, so the launch
    order scratch of each context is in use): every result is its own — no state is shared between handles"""
    import threading
    import decompress_amd
    from decompress_amd import workloads
    plains = [[workloads.text(1000 * t + i, 3000 + 37 * i) for i in range(2200)] for t in range(2)]
    srcs = [[zlib.compress(p, 6) for p in ps] for ps in plains]
    out, errs = [None, None], []

    def work(t):
        try:
            eng = decompress_amd.Engine(0, own_stream=True)
            for _ in range(3):
                out[t] = eng.inflate_many(srcs[t], [len(p) for p in plains[t]], decompress_amd.FORMAT_ZLIB)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(2):
        for p, z, (st, used, o, adler) in zip(plains[t], srcs[t], out[t]):
            assert (st, used, o, adler) == (0, len(z), p, zlib.adler32(p))


@pytest.mark.parametrize("pinned", [True, False], ids=["pinned", "pageable"])
def test_host_entry_points_in_slices(eng, pinned):
    """md_inflate_batch_host / md_deflate_batch_host with a batch large enough to go through in pipelined slices of
    streams (copies under kernels on three HIP streams): results equal the one-slice form's and the oracle's, whatever the
    layout of the blobs (ascending here; scattered: a reversed order of the streams in the blob)."""
    import decompress_amd
    from tests import oracle_lib
    from decompress_amd import workloads
    orc = oracle_lib.load()
    rng = random.Random(5)
    n, nb = 4400, 24576
    plains = [workloads.text(1000 + (i % 64), nb)[: nb - (i % 7) * 100] for i in range(n)]
    zs = [zlib.compress(p, 6) for p in plains[:64]]
    streams = [zs[i % 64] if (i % 7) == 0 else None for i in range(n)]
    for i in range(n):
        if streams[i] is None:
            streams[i] = zlib.compress(plains[i], 1 + i % 9)
    bad = rng.randrange(n)
    streams[bad] = streams[bad][:-9]  # one truncated stream somewhere: its status, nobody else's
    for order in ("ascending", "reversed"):
        lens = np.array([len(z) for z in streams], dtype=np.uint64)
        idx = list(range(n)) if order == "ascending" else list(range(n - 1, -1, -1))
        in_off = np.zeros(n, dtype=np.uint64)
        pos = 0
        for i in idx:
            in_off[i] = pos
            pos += (len(streams[i]) + 15) // 16 * 16
        cap = 32768
        out_off = np.zeros(n, dtype=np.uint64)
        for k, i in enumerate(idx):
            out_off[i] = k * cap
        mk = eng.host_buffer if pinned else (lambda m: np.zeros(m, dtype=np.uint8))
        h_in, h_out = mk(pos + 64), mk(n * cap)
        for i in range(n):
            h_in[int(in_off[i]):int(in_off[i]) + len(streams[i])] = np.frombuffer(streams[i], dtype=np.uint8)
        got = {}
        for slices in (16, 1):
            eng.set_option("host_pipeline_slices", slices)
            h_out[:] = 0
            res = eng.inflate_batch_host(decompress_amd.FORMAT_ZLIB, h_in, in_off, lens, h_out, out_off, np.full(n, cap, dtype=np.uint64))
            got[slices] = (tuple(a.copy() for a in res), h_out.copy())
        eng.set_option("host_pipeline_slices", 16)
        for a, b in zip(got[16][0], got[1][0]):
            assert (a == b).all(), order
        assert (got[16][1] == got[1][1]).all(), order
        out_len, consumed, status, checksum = got[16][0]
        for i in range(0, n, 37):
            rc, used, out = orc.zl_inflate(streams[i], cap)
            assert status[i] == rc, (order, i)
            if rc == 0:
                assert (int(out_len[i]), int(consumed[i])) == (len(out), used) and checksum[i] == zlib.adler32(out)
                assert got[16][1][int(out_off[i]):int(out_off[i]) + len(out)].tobytes() == out, (order, i)
        assert status[bad] != 0 and (np.delete(status, bad) == 0).all()
        # ... and back: the deflate entry point on the plaintexts (4 400 >= 4 096 streams: still one slice; 8 800: two)
        if order == "ascending":
            pl = plains + plains
            m = len(pl)
            p_len = np.array([len(p) for p in pl], dtype=np.uint64)
            p_off = np.arange(m, dtype=np.uint64) * nb
            h_p = mk(m * nb)
            for i, p in enumerate(pl):
                h_p[i * nb:i * nb + len(p)] = np.frombuffer(p, dtype=np.uint8)
            h_c = mk(m * cap)
            dres = {}
            for slices in (16, 1):
                eng.set_option("host_pipeline_slices", slices)
                h_c[:] = 0
                r = eng.deflate_batch_host(decompress_amd.FORMAT_ZLIB, h_p, p_off, p_len, h_c, np.arange(m, dtype=np.uint64) * cap,
                                           np.full(m, cap, dtype=np.uint64), level=4, queue=1024)
                dres[slices] = (tuple(a.copy() for a in r), h_c.copy())
            eng.set_option("host_pipeline_slices", 16)
            assert all((a == b).all() for a, b in zip(dres[16][0], dres[1][0])) and (dres[16][1] == dres[1][1]).all()
            assert (dres[16][0][1] == 0).all()
            for i in range(0, m, 211):
                z = dres[16][1][i * cap:i * cap + int(dres[16][0][0][i])].tobytes()
                assert zlib.decompress(z) == pl[i] and z == orc.zl_deflate(pl[i], 4, queue=1024), i
    eng.set_option("release_workspace", 0)


@pytest.mark.parametrize("pinned", [True, False], ids=["pinned", "pageable"])
def test_deflate_host_entry_point_in_slices_of_positions(eng, oracle, pinned):
    """md_deflate_batch_host on LONG buffers at equal distances (C3's shape): the batch goes through the kernels in slices
    of positions with its input arriving and its finished output leaving as strided copies under them (capi.cpp
    deflate_host_positions) - bytes, lengths, statuses and Adler-32 equal the one-after-the-other form's and the oracle's;
    ragged lengths, an empty buffer, one buffer too small for its output"""
    import decompress_amd
    from decompress_amd import workloads
    pitch, cap = 700032, 760000
    lens = [700000, 650001, 300000, 100, 0, 699999, 262144, 262145, 524288, 1, 33000, 700032]
    bufs = [(workloads.text(300 + i, n) if i % 3 else workloads.ascii_uniform(300 + i, n)) for i, n in enumerate(lens)]
    n = len(bufs)
    mk = eng.host_buffer if pinned else (lambda m: np.zeros(m, dtype=np.uint8))
    h_in, h_out = mk(n * pitch), mk(n * cap)
    for i, b in enumerate(bufs):
        h_in[i * pitch:i * pitch + len(b)] = np.frombuffer(b, dtype=np.uint8)
    in_off, in_len = np.arange(n, dtype=np.uint64) * pitch, np.array(lens, dtype=np.uint64)
    out_off, out_cap = np.arange(n, dtype=np.uint64) * cap, np.full(n, cap, dtype=np.uint64)
    out_cap[5] = 1000  # no room: its status, nobody else's
    for fmt, level in ((decompress_amd.FORMAT_ZLIB, 6), (decompress_amd.FORMAT_DEFLATE, 1), (decompress_amd.FORMAT_ZLIB, 9)):
        got = {}
        for slices in (16, 1):
            eng.set_option("host_pipeline_slices", slices)
            h_out[:] = 0
            r = eng.deflate_batch_host(fmt, h_in, in_off, in_len, h_out, out_off, out_cap, level=level)
            got[slices] = ([a.copy() for a in r], [h_out[i * cap:i * cap + int(r[0][i])].tobytes() for i in range(n)])
        eng.set_option("host_pipeline_slices", 16)
        good = got[1][0][1] == 0
        for a, b in zip(got[16][0], got[1][0]):  # (the checksum of a stream that failed is nobody's business)
            assert (a[good] == b[good]).all(), (fmt, level)
        assert (got[16][0][1] == got[1][0][1]).all() and got[16][1] == got[1][1], (fmt, level)
        out_len, status, adler = got[16][0]
        assert status[5] != 0 and (np.delete(status, 5) == 0).all()
        for i, b in enumerate(bufs):
            if i == 5:
                continue
            want = oracle.zl_deflate(b, level) if fmt == decompress_amd.FORMAT_ZLIB else oracle.deflate_raw(b, level)[0]
            assert got[16][1][i] == want, (fmt, level, i)
            assert (int(adler[i]) & 0xffffffff) == zlib.adler32(b)
    eng.set_option("release_workspace", 0)
