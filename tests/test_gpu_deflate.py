"""Parity of the HIP deflate path (through the C ABI) with the CPU oracle: identical
compressed bytes at the same level / queue / driver.  Needs an MI355X: `pytest -m gpu`."""
import ctypes
import os
import random
import zlib

import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import decompress_amd
    return decompress_amd.Engine(0)


def _planted(seed, n, dists):
    """random ASCII with short matches planted at exact distances (window-edge cases:
    the head candidate is admitted at distance MAX_DIST = 32506, chain links only below it)"""
    from decompress_amd import workloads
    rng = random.Random(seed)
    b = bytearray(workloads.ascii_uniform(seed, n))
    p = 33000
    while p + 300 < n:
        d = rng.choice(dists)
        ln = rng.choice((3, 4, 5, 9, 40, 258))
        b[p:p + ln] = b[p - d:p - d + ln]
        p += ln + rng.randrange(1, 700)
    return bytes(b)


def _datasets():
    from decompress_amd import workloads
    rng = random.Random(5)
    return {
        "edge": _planted(11, 140000, (32504, 32505, 32506, 32507, 32508, 32768, 4096, 4097, 1)),
        "planted": _planted(12, 100000, (1, 2, 3, 100, 5000, 20000, 32000, 32506)),
        "empty": b"", "one": b"x", "abcde": b"abcde", "aaaaa": b"aaaaa", "runs": b"a" * 70000 + b"b" * 300,
        "text": workloads.text(3, 150000), "ascii": workloads.ascii_uniform(4, 70000),
        "rand": bytes(rng.getrandbits(8) for _ in range(66000)), "zeros": bytes(100000),
        "tail33k": workloads.text(7, 33000), "tail65300": workloads.text(8, 65300),
        "tail98500": workloads.text(9, 98500),
    }


@pytest.mark.parametrize("driver", [0, 1, 2], ids=["Zl.Def", "De.Higher", "CLI"])
def test_bytes_equal_oracle(eng, oracle, driver):
    data = _datasets()
    names = list(data)
    for level in range(10):
        if driver == 1 and level != 4:
            continue
        for q in (16, 4096):
            wants = [oracle.deflate_raw(data[k], level, q, driver) for k in names]
            res = eng.deflate_many([data[k] for k in names], level=level, queue=q, driver=driver,
                                   caps=[len(w[0]) + 3 if w[0] is not None else 64 for w in wants])
            for k, (st, out, adler), (want, wadler) in zip(names, res, wants):
                if want is None:  # De.Queue.Full in the reference (test_cli_driver_queue_full)
                    assert (st, out) == (13, b""), (k, level, q)
                    continue
                assert st == 0, (k, level, q)
                assert out == want, (k, level, q, len(out), len(want))
                assert adler == wadler == zlib.adler32(data[k])


def test_zlib_frame_and_fixed(eng, oracle):
    import decompress_amd
    from decompress_amd import workloads
    bufs = [workloads.text(20 + i, 40000 + 1000 * i) for i in range(6)]
    for level, dyn in ((6, True), (6, False), (1, True), (0, True), (9, True)):
        res = eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=level, dynamic=dyn)
        for b, (st, out, _) in zip(bufs, res):
            assert st == 0
            assert out == oracle.zl_deflate(b, level, 4096, dyn)
            assert zlib.decompress(out) == b


def test_round_trip_on_gpu(eng):
    """deflate on the GPU, inflate on the GPU: the reference's own corpus test shape."""
    import decompress_amd
    from decompress_amd import workloads
    bufs = [workloads.text(40 + i, 100000) for i in range(4)] + [workloads.ascii_uniform(50, 120000)]
    res = eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=6)
    back = eng.inflate_many([r[1] for r in res], [len(b) for b in bufs], decompress_amd.FORMAT_ZLIB)
    for b, (st, used, out, adler) in zip(bufs, back):
        assert st == 0 and out == b and adler == zlib.adler32(b)


def test_corpus_by_level_matrix(eng, oracle):
    """the reference's own deflate test shape, test/test_deflate.ml:19-120: every file of test/corpus at every level 0 .. 9
    through the Zl driver - bytes = the oracle's, Adler-32 = libz's, and the streams inflate back on the GPU (and in libz)"""
    from concurrent.futures import ThreadPoolExecutor
    import decompress_amd
    from decompress_amd import workloads
    files = list(workloads.corpus().items())
    assert len(files) == 15
    bufs = [b for _, b in files]
    with ThreadPoolExecutor(16) as ex:  # (ctypes releases the GIL)
        want = {lvl: list(ex.map(lambda b, lvl=lvl: oracle.zl_deflate(b, lvl, 4096, True), bufs)) for lvl in range(10)}
    for lvl in range(10):
        res = eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=lvl)
        for (name, b), (st, out, adler), w in zip(files, res, want[lvl]):
            assert st == 0 and out == w, (name, lvl, len(out), len(w))
            assert adler == zlib.adler32(b) and zlib.decompress(out) == b
        back = eng.inflate_many([r[1] for r in res], [len(b) for b in bufs], decompress_amd.FORMAT_ZLIB)
        for (name, b), (st, used, out, adler), r in zip(files, back, res):
            assert (st, used, adler) == (0, len(r[1]), zlib.adler32(b)) and out == b, (name, lvl)


def test_equal_hashes_inside_a_step(eng, oracle):
    """the link kernel sorts out positions that share a hash inside one 64-position step (and across the steps of a
    group, and across the wavefronts' groups) by ballots: inputs made of short periods and runs put dozens of equal
    hashes into every step"""
    from decompress_amd import workloads
    bufs = [workloads.text(60, 90000), workloads.ascii_uniform(61, 150000), b"abc" * 30000, b"q" * 5, b"q" * 100000,
            b"ab" * 50000, bytes(range(256)) * 400, (b"x" * 63 + b"y") * 2000, b"abcdefgh" * 20000]
    for level in (4, 6, 9):
        for b, (st, out, _) in zip(bufs, eng.deflate_many(bufs, level=level)):
            assert st == 0 and out == oracle.deflate_raw(b, level)[0]
    for b, (st, out, _) in zip(bufs, eng.deflate_many(bufs, level=6, matcher=1)):
        assert st == 0 and out == oracle.deflate_raw(b, 6, matcher=1)[0]


def test_front_workspace_shapes(eng, oracle):
    """the position-indexed workspace of the link / match kernels over awkward batches: empty and tiny streams between
    large ones (slots of zero positions), many small streams (more streams than chunks), one stream of many chunks,
    lengths around the 64 / 256 / 512-position granules; with and without the size hint; a hint that is too small
    refuses the batch on the device instead of writing past the workspace"""
    import decompress_amd
    import numpy as np
    import torch
    from decompress_amd import workloads
    rng = np.random.default_rng(5)
    big = workloads.text(400, 3 << 20)
    shapes = [b"", big[:1], big[:3], big[:4], big[:5], big[:63], big[:64], big[:65], big[:255], big[:256], big[:257], big[:511],
              big[:512], big[:513], b"", big[:70000], big, workloads.ascii_uniform(9, 200000), big[:4], b"", big[:1000]]
    for level in (1, 6):
        for (st, z, adler), data in zip(eng.deflate_many(shapes, level=level), shapes):
            assert st == 0 and z == oracle.deflate_raw(data, level)[0] and adler == zlib.adler32(data)
    small = [bytes(rng.integers(97, 101, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 300, size=5000)]
    res = eng.deflate_many(small, level=4)
    for k in range(0, len(small), 97):
        assert res[k][0] == 0 and res[k][1] == oracle.deflate_raw(small[k], 4)[0]
    assert all(st == 0 for st, _, _ in res)
    # no hint (the totals are read back) and a hint that is too small
    bufs = [big[:50000], big[:70000]]
    blob, off, ln = workloads.pack(bufs)
    dev = eng.device
    t = lambda a: torch.from_numpy(a).to(dev)
    cap = np.array([200000, 200000], dtype=np.int64)
    d_out = torch.zeros(400000, dtype=torch.uint8, device=dev)
    for hint, ok in ((0, True), (int(ln.sum()), True), (1000, False)):
        out_len, status, _ = eng.deflate_batch(decompress_amd.FORMAT_DEFLATE, t(blob), t(off), t(ln), d_out, t(np.array([0, 200000])),
                                               t(cap), level=6, total_in=hint)
        torch.cuda.synchronize(dev)
        if ok:
            assert status.tolist() == [0, 0]
            got = d_out[:int(out_len[0])].cpu().numpy().tobytes()
            assert got == oracle.deflate_raw(bufs[0], 6)[0]
        else:
            assert status.tolist() == [-1, -1] and out_len.tolist() == [0, 0]


def test_lz_matcher_equals_oracle(oracle):
    """lib/lz.ml's match finder; the oracle's variant is pinned to libz's own
    decisions (tests/test_oracle_lz.py), the GPU must produce the oracle's bytes."""
    import decompress_amd
    e = decompress_amd.Engine(0)
    e.set_matcher(1)
    data = _datasets()
    names = list(data)
    for driver in (0, 1, 2):
        for level in ((4,) if driver == 1 else (0, 3, 4, 6, 9)):
            for q in (16, 4096):
                wants = [oracle.deflate_raw(data[k], level, q, driver, matcher=1) for k in names]
                res = e.deflate_many([data[k] for k in names], level=level, queue=q, driver=driver,
                                     caps=[len(w[0]) + 3 if w[0] is not None else 64 for w in wants])
                for k, (st, out, adler), (want, wadler) in zip(names, res, wants):
                    if want is None:
                        assert (st, out) == (13, b""), (k, driver, level, q)
                        continue
                    assert st == 0 and out == want, (k, driver, level, q, len(out), len(want))
    from decompress_amd import lz
    assert lz.compress(data["text"], level=6) == oracle.deflate_raw(data["text"], 6, matcher=1)[0]


def test_cli_driver_queue_full(eng, oracle):
    """bin/decompress.ml:67 pushes an end-of-block command unconditionally at `End; when the final
    literal has just filled the queue, De.Queue.push_exn raises Queue.Full (lib/de.ml:2214-2217)"""
    assert oracle.deflate_raw(b"geg", 6, 4, 2)[0] is None
    st, out, _ = eng.deflate_many([b"geg"], level=6, queue=4, driver=2)[0]
    assert (st, out) == (13, b"")
    st, out, _ = eng.deflate_many([b"geg"], level=6, queue=8, driver=2)[0]
    assert st == 0 and zlib.decompress(out, -15) == b"geg"


def test_output_too_small(eng):
    from decompress_amd import workloads
    b = workloads.ascii_uniform(1, 10000)
    st, out, _ = eng.deflate_many([b], caps=[100])[0]
    assert st == 2  # Unexpected_end_of_output


KAT = {c["name"]: c for c in load_golden("deflate_kat.json")}


def test_kats_through_gpu(eng):
    """the reference's compressed-byte KATs on the HIP path: the encoder alone (test/test.ml:507-531
    huffman_length_extra, :1037-1055 flat) and the match finder alone (:798-813, "abcde")"""
    from decompress_amd import de
    c = KAT["huffman_length_extra"]
    assert de.Def.encode(c["cmds"], de.Def.DYNAMIC) == bytes.fromhex(c["out"])
    assert zlib.decompress(bytes.fromhex(c["out"]), -15) == bytes.fromhex(c["inflated"])
    c = KAT["flat"]
    assert de.Def.encode(c["cmds"], de.Def.FLAT) == bytes.fromhex(c["out"])
    c = KAT["lz77_1"]
    cmds, lits, dsts = de.Lz77.compress(bytes.fromhex(c["src"]), level=4)
    assert cmds == c["cmds"]
    assert [lits[x] for x in b"abcde"] == [1] * 5 and lits[256] == 1 and sum(dsts) == 0
    st, out, _ = eng.deflate_many([b"abcde"], level=4, driver=1)[0]
    assert st == 0 and zlib.decompress(out, -15) == b"abcde"


ENC = load_golden("encode_cases.json")


@pytest.mark.parametrize("case", ENC, ids=[c["name"] for c in ENC])
def test_encode_cases_through_gpu(eng, oracle, case):
    """the reference's encoder-built cases (test/test_ns.ml:221-252, :353-615, :837-915, :1016-1057; test/test.ml:507-611,
    :704-767, :910-1108): Def.encode driven on the GPU exactly as the test drives it (md_de_def_run) answers what the test
    demands, writes the bytes the reference pins (and the oracle's everywhere), and the GPU inflates them to the
    expected result"""
    from decompress_amd import de
    z, rcs = de.Def.run(case["ops"], case["queue_len"])
    assert rcs == case["rcs"], case["ref"]
    if "out" in case:
        assert z == bytes.fromhex(case["out"])
    assert z == oracle.def_script(case["ops"], case["queue_len"])[0]
    st, used, out, _ = eng.inflate_many([z], [case["dst_cap"]])[0]
    assert st == case["status"], case["ref"]
    if st == 0:
        assert out == bytes.fromhex(case["plain"])
        if case["decoder"] == "ns":
            assert used == len(z)  # Ok (De.bigstring_length src, String.length expected)
        else:  # test/test.ml: the streaming decoder over the same bytes
            assert de.Inf.decode_chunks([z, b""])[:2] == ("Ok", out)


def test_def_run_misuse(eng):
    """Queue.Full and malformed operation lists are statuses, not crashes"""
    from decompress_amd import de
    import decompress_amd
    with pytest.raises(decompress_amd.Error, match="Queue.Full"):
        de.Def.run([de.Def.OP_FILL, 5, 1, 2, 3, 4, 5], queue=4)
    for bad in ([99], [de.Def.OP_FILL, 3, 1], [de.Def.OP_BLOCK, 7, 1], [de.Def.OP_FILL, 1, 0x2000000 | (300 << 16)],
                [de.Def.OP_SUCC_LENGTH, 2]):
        with pytest.raises(decompress_amd.Error):
            de.Def.run(bad)
    assert de.Def.run([]) == (b"", [])


@pytest.mark.parametrize("name", ["tree_0", "tree_rfc5322_corpus"])
def test_tree_kats_through_gpu(oracle, name):
    """test/test.ml:1169-1237: the Huffman trees of a given histogram.  A command list with exactly that histogram
    goes through De.Def on the GPU; the block header carries the tree, so byte equality with the oracle (whose
    T.make reproduces the KAT's lengths and codes, tests/test_oracle_deflate.py) pins the GPU's tree."""
    from decompress_amd import de
    c = KAT[name]
    lb = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    cmds = []
    for sym, f in enumerate(c["freqs"][:286]):
        if sym < 256:
            cmds += [sym] * f
        elif sym > 256:
            cmds += [de.copy_cmd(1, lb[sym - 257])] * f
    cmds.append(de.EOB)
    mc, lens, codes, _ = oracle.tree_make(c["length"], list(c["freqs"]))
    for sym, l in c["lengths"].items():
        assert lens[int(sym)] == l
    assert de.Def.encode(cmds, de.Def.DYNAMIC) == oracle.encode_cmds(cmds, "dynamic")


def test_lz77_alone_equals_oracle(eng, oracle):
    """De.Lz77.compress / Lz.compress on their own (md_de_lz77_compress): the commands of every queue fill and the
    two cumulative histograms equal the oracle's"""
    from decompress_amd import de
    data = _datasets()
    for matcher in (0, 1):
        for level, q in ((0, 4096), (1, 16), (4, 4096), (6, 256), (9, 4096)):
            if matcher == 1 and level == 0:
                continue
            for k in ("empty", "one", "abcde", "aaaaa", "runs", "text", "ascii", "tail65300", "planted"):
                got = de.Lz77.compress(data[k], level=level, queue=q, matcher=matcher)
                want = oracle.lz77_all(data[k], level=level, queue=q, matcher=matcher)
                assert got[0] == want[0], (matcher, level, q, k, len(got[0]), len(want[0]))
                assert (got[1], got[2]) == (want[1], want[2]), (matcher, level, q, k)


def test_def_encode_equals_oracle(eng, oracle):
    """De.Def.encode on its own (md_de_def_encode): random command lists, the three block kinds"""
    from decompress_amd import de
    rng = random.Random(44)
    for trial in range(12):
        cmds = []
        for _ in range(rng.choice((0, 1, 5, 300, 5000))):
            if rng.random() < 0.6:
                cmds.append(rng.randrange(256) if trial % 2 else rng.choice(b"etaoin shrdlu"))
            else:
                cmds.append(de.copy_cmd(rng.randrange(1, 32769), rng.randrange(3, 259)))
        cmds.append(de.EOB)
        for kind, name in ((de.Def.FIXED, "fixed"), (de.Def.DYNAMIC, "dynamic")):
            assert de.Def.encode(cmds, kind) == oracle.encode_cmds(cmds, name), (trial, name, len(cmds))
    lits = [rng.randrange(256) for _ in range(70000)] + [de.EOB]
    assert de.Def.encode(lits, de.Def.FLAT) == oracle.encode_cmds(lits, "flat")


def test_c3_stream_size(eng, oracle):
    """BASELINE config 3 at its stream size: 8 x 1 MiB printable-ASCII buffers, level 6, queue 4096, Zl driver
    (32 window slides, ~256 queue fills each) — bytes equal the oracle's"""
    import decompress_amd
    from decompress_amd import workloads
    bufs = [workloads.ascii_uniform(0xC3 + i, 1 << 20) for i in range(8)]
    res = eng.deflate_many(bufs, decompress_amd.FORMAT_ZLIB, level=6, queue=4096)
    for b, (st, out, adler) in zip(bufs, res):
        assert st == 0 and out == oracle.zl_deflate(b, 6) and adler == zlib.adler32(b)
        assert zlib.decompress(out) == b


def test_streaming_encoder_shim(eng, oracle):
    """Zl.Def.encoder's protocol (`Await / `Flush / `End) above the batch ABI: md_def_*"""
    from decompress_amd import _lib, workloads
    lib = eng.lib
    data = workloads.text(5, 200000)
    params = eng._params(6, 4096, 0, True)
    o = ctypes.create_string_buffer(4096)
    s = lib.md_def_encoder(eng.ctx, 1, ctypes.byref(params), o, len(o))
    out, pos, sigs = bytearray(), 0, []
    while True:
        sig = lib.md_def_encode(s)
        sigs.append(sig)
        if sig == 0:  # `Await
            chunk = data[pos:pos + 30000]
            pos += len(chunk)
            assert lib.md_def_src(s, chunk, 0, len(chunk)) == 0
        elif sig == 1:  # `Flush
            out += o.raw[:len(o) - lib.md_def_dst_rem(s)]
            lib.md_def_dst(s, o, len(o))
        else:
            out += o.raw[:len(o) - lib.md_def_dst_rem(s)]
            break
    assert sig == 2 and lib.md_def_status(s) == 0 and lib.md_def_checksum(s) == zlib.adler32(data)
    lib.md_def_free(s)
    assert bytes(out) == oracle.zl_deflate(data, 6)
    assert sigs.count(0) == 8 and sigs.count(1) >= 10


def _rss_bytes():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def _push_through_encoder(eng, fmt, pieces, level, o_len=65536):
    """pieces: an iterator of bytes -> (sha256 of the output, its length, checksum, peak growth of the resident set while
    the pieces went through md_def_*)"""
    import hashlib
    lib = eng.lib
    params = eng._params(level, 4096, 0, True)
    o = ctypes.create_string_buffer(o_len)
    s = lib.md_def_encoder(eng.ctx, fmt, ctypes.byref(params), o, len(o))
    assert s
    h, n_out, done = hashlib.sha256(), 0, False
    rss0, peak = _rss_bytes(), 0
    it = iter(pieces)
    while True:
        sig = lib.md_def_encode(s)
        if sig == 0:  # `Await
            chunk = next(it, b"")
            assert lib.md_def_src(s, chunk, 0, len(chunk)) == 0
            peak = max(peak, _rss_bytes() - rss0)
        elif sig in (1, 2):  # `Flush / `End
            k = len(o) - lib.md_def_dst_rem(s)
            h.update(o.raw[:k])
            n_out += k
            if sig == 2:
                break
            lib.md_def_dst(s, o, len(o))
        else:
            raise AssertionError(lib.md_def_status(s))
    peak = max(peak, _rss_bytes() - rss0)
    status, checksum = lib.md_def_status(s), lib.md_def_checksum(s)
    lib.md_def_free(s)
    assert status == 0
    return h.hexdigest(), n_out, checksum, peak


def _encode_in_pieces(eng, fmt, data, piece, level, queue=4096, o_len=4096, driver=0, matcher=None):
    """data through md_def_* `piece` bytes per `Await -> (output, number of `Flush answers before the last piece went in)"""
    lib = eng.lib
    params = eng._params(level, queue, driver, True, matcher=matcher)
    o = ctypes.create_string_buffer(o_len)
    s = lib.md_def_encoder(eng.ctx, fmt, ctypes.byref(params), o, len(o))
    assert s
    out, pos, early = bytearray(), 0, 0
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
            early += pos < len(data)
            lib.md_def_dst(s, o, len(o))
        else:
            raise AssertionError(lib.md_def_status(s))
    assert lib.md_def_status(s) == 0
    lib.md_def_free(s)
    return bytes(out), early


@pytest.mark.parametrize("piece", [1000, 4096, 65536, 100000])
def test_encoder_in_pieces_vs_oracle_in_pieces(eng, oracle, piece):
    """The encoder goes on from device-resident state piece after piece (md_set_option "encoder_piece_bytes" = the piece:
    every `Await is a launch): the bytes are those of the oracle handed the input in the same pieces (orc_set_src_piece),
    and output comes out while input is still going in."""
    import random
    import decompress_amd
    from decompress_amd import workloads
    rng = random.Random(piece)
    eng.set_option("encoder_piece_bytes", piece)
    try:
        for t in range(6):
            n = rng.choice([0, 1, 262, 5000, 70000, 200000, 400000]) if t else 300000
            kind = t % 3
            data = (workloads.text(piece + t, n) if kind == 0 else bytes(rng.getrandbits(2) for _ in range(n)) if kind == 1
                    else (workloads.text(t, 3000) * (n // 3000 + 1))[:n])
            for level in (0, 1, 4, 6, 9):
                if piece < 4096 and n > 100000 and level not in (4, 6):
                    continue
                with oracle.src_piece(piece):
                    want_zl = oracle.zl_deflate(data, level)
                    want_gz = oracle.gz_deflate(data, level=level) if level == 4 else None
                got, early = _encode_in_pieces(eng, decompress_amd.FORMAT_ZLIB, data, piece, level)
                assert got == want_zl, (t, n, level)
                assert zlib.decompress(got) == data
                if n >= 200000 and level and kind != 2:
                    assert early > 0  # output before the end of the input
                if want_gz is not None:
                    got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_GZIP, data, piece, level)
                    assert got == want_gz, (t, n, level)
    finally:
        eng.set_option("encoder_piece_bytes", 1 << 20)


def test_encoder_origin_moves(eng, oracle):
    """The device counts positions in 32 bits from an origin that moves up as the stream grows (every 2 GiB; with
    deflate_test_flags bit 4 every 128 KiB): the bytes do not change, gzip's ISIZE is the whole length"""
    import decompress_amd
    from decompress_amd import workloads
    data = workloads.text(91, 1_500_000)
    eng.set_option("encoder_piece_bytes", 50000)
    eng.set_option("deflate_test_flags", 16)
    try:
        with oracle.src_piece(50000):
            want_zl, want_gz = oracle.zl_deflate(data, 6), oracle.gz_deflate(data, level=4)
        got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_ZLIB, data, 50000, 6)
        assert got == want_zl
        got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_GZIP, data, 50000, 4)
        assert got == want_gz and got[-4:] == len(data).to_bytes(4, "little")
    finally:
        eng.set_option("encoder_piece_bytes", 1 << 20)
        eng.set_option("deflate_test_flags", 0)


def test_encoder_pieces_flush_a_full_queue_of_matches(eng, oracle):
    """a piece in which the queue fills flushes the 4 096 commands it held back - short far matches at 3-4 bytes of
    output each: the room a piece gets counts them at their worst (6 bytes), not at a literal's (it was sized for
    literals once: `Unexpected end of output` from the encoder)"""
    import random
    import decompress_amd
    rng = random.Random(5)
    grams = [bytes(rng.getrandbits(8) for _ in range(rng.randrange(3, 6))) for _ in range(50)]
    data = b"".join(rng.choice(grams) for _ in range(50000))
    eng.set_option("encoder_piece_bytes", 1000)
    try:
        with oracle.src_piece(1000):
            want = oracle.zl_deflate(data, 6)
        got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_ZLIB, data, 1000, 6)
        assert got == want and zlib.decompress(got) == data
    finally:
        eng.set_option("encoder_piece_bytes", 1 << 20)


def test_encoder_in_pieces_other_drivers_and_matcher(eng, oracle):
    """De.Higher's driver, the CLI's, and lib/lz.ml's matcher (its fill_window is De.Lz77's, lib/lz.ml:382-426) through
    the encoder in pieces: raw DEFLATE equal to the oracle handed the same pieces"""
    import decompress_amd
    from decompress_amd import workloads
    data = workloads.text(314, 260000)
    eng.set_option("encoder_piece_bytes", 40000)
    try:
        for driver, matcher, level in ((1, 0, 4), (2, 0, 6), (0, 1, 6), (0, 1, 2)):
            with oracle.src_piece(40000):
                want, _ = oracle.deflate_raw(data, level=level, queue=4096, driver=driver, dynamic=True, matcher=matcher)
            got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_DEFLATE, data, 40000, level, driver=driver, matcher=matcher)
            assert got == want, (driver, matcher, level, len(got), len(want))
            assert zlib.decompressobj(-15).decompress(got) == data
    finally:
        eng.set_option("encoder_piece_bytes", 1 << 20)


def test_encoder_small_queue_in_pieces(eng, oracle):
    """a 16-command queue: a block every few commands, many of them per piece"""
    import decompress_amd
    from decompress_amd import workloads
    data = workloads.text(77, 50000)
    eng.set_option("encoder_piece_bytes", 3000)
    try:
        with oracle.src_piece(3000):
            want = oracle.zl_deflate(data, 6, queue=16)
        got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_ZLIB, data, 3000, 6, queue=16)
        assert got == want and zlib.decompress(got) == data
    finally:
        eng.set_option("encoder_piece_bytes", 1 << 20)


def test_encoder_64mib_in_pieces_bounded_host_memory(eng, oracle):
    """64 MiB pushed through md_def_* in 64 KiB pieces, zlib and gzip: host and device keep the last 64 KiB and the
    piece being gathered (1 MiB a launch) - the resident set grows by less than 4 MiB - and the bytes are those of the
    oracle handed the input 1 MiB at a time (which are also those of the one-shot oracle here)."""
    import hashlib
    import decompress_amd
    from decompress_amd import workloads
    piece, npieces = 65536, 1024
    base = [workloads.text(0x600 + i, piece) for i in range(16)]  # (the stream: these 16 pieces, cycled)
    for fmt in (decompress_amd.FORMAT_ZLIB, decompress_amd.FORMAT_GZIP):
        # (a first pass of 2 MiB: what the runtime maps when these kernels are launched for the first time in the process -
        # code objects, the context's scratch - is not the encoder's memory)
        _push_through_encoder(eng, fmt, (base[i % 16] for i in range(32)), 4)
        digest, n_out, checksum, peak = _push_through_encoder(eng, fmt, (base[i % 16] for i in range(npieces)), 4)
        assert peak < 4 << 20, peak
        whole = b"".join(base[i % 16] for i in range(npieces))
        with oracle.src_piece(1 << 20):
            want = oracle.zl_deflate(whole, 4) if fmt == decompress_amd.FORMAT_ZLIB else oracle.gz_deflate(whole, level=4)
        assert (n_out, digest) == (len(want), hashlib.sha256(want).hexdigest())
        assert checksum == (zlib.adler32(whole) if fmt == decompress_amd.FORMAT_ZLIB else zlib.crc32(whole))
        del whole, want


@pytest.mark.skipif(not os.environ.get("MD_SLOW"), reason="the reference's `Slow test: 4 GB through the encoder (set MD_SLOW=1)")
def test_gzip_huge(eng):
    """test/test.ml:1991-2014 'GZip with huge file': 4 000 055 296 zero bytes through Gz.Def at level 4 in io_buffer_size
    pieces; here the result is also checked (gzip -t semantics: libz inflates it back to that many zeros)"""
    import decompress_amd
    zero = bytes(65536)
    n = -(-4_000_000_000 // 65536)
    lib = eng.lib
    params = eng._params(4, 4096, 0, True)
    o = ctypes.create_string_buffer(65536)
    s = lib.md_def_encoder(eng.ctx, decompress_amd.FORMAT_GZIP, ctypes.byref(params), o, len(o))
    d, total, fed = zlib.decompressobj(31), 0, 0
    while True:
        sig = lib.md_def_encode(s)
        if sig == 0:
            chunk = zero if fed < n else b""
            fed += 1
            assert lib.md_def_src(s, chunk, 0, len(chunk)) == 0
        elif sig in (1, 2):
            for off in range(0, len(o) - lib.md_def_dst_rem(s), 4096):  # (inflate in small steps: zeros expand 1000-fold)
                blk = o.raw[off:min(off + 4096, len(o) - lib.md_def_dst_rem(s))]
                while blk:
                    out = d.decompress(blk, 1 << 24)
                    assert not any(out[:: 4099])
                    total += len(out)
                    blk = d.unconsumed_tail
            if sig == 2:
                break
            lib.md_def_dst(s, o, len(o))
        else:
            raise AssertionError(lib.md_def_status(s))
    assert lib.md_def_status(s) == 0 and d.eof and total == n * 65536
    lib.md_def_free(s)


def test_encoder_params_checked_at_construction(eng):
    """md_def_encoder refuses bad parameters when the encoder is made (a zero queue_len used to divide by zero at the end of
    the input); md_de_lz77_compress reports the room it needs when cmds_cap is too small"""
    import decompress_amd
    from decompress_amd import _lib, de
    lib = eng.lib
    o = ctypes.create_string_buffer(64)
    for bad in (_lib.DeflateParams(), _lib.DeflateParams(6, 4095, 0, 1, 0, None, 0, 0), _lib.DeflateParams(10, 4096, 0, 1, 0, None, 0, 0),
                _lib.DeflateParams(6, 4096, 7, 1, 0, None, 0, 0), _lib.DeflateParams(6, 4096, 0, 1, 3, None, 0, 0),
                _lib.DeflateParams(6, 4096, 0, 1, 0, None, 12, 0)):
        assert not lib.md_def_encoder(eng.ctx, 1, ctypes.byref(bad), o, len(o))
    assert not lib.md_def_encoder(eng.ctx, 9, ctypes.byref(eng._params(6, 4096, 0, True)), o, len(o))
    s = lib.md_def_encoder(eng.ctx, 1, ctypes.byref(_lib.DeflateParams(6, 4096, 0, 1, 0, None, 15, 0)), o, len(o))
    assert s
    lib.md_def_free(s)
    src = b"abcdefgh" * 64
    full, _, _ = de.Lz77.compress(src, level=6)
    cmds, n = (ctypes.c_uint32 * 4)(), ctypes.c_size_t()
    st = lib.md_de_lz77_compress(eng.ctx, 6, 4096, 0, src, len(src), cmds, 4, ctypes.byref(n), None, None)
    assert st == 2 and n.value == len(full)  # Unexpected_end_of_output, *ncmds = the number needed


def test_c_abi_single(eng):
    from decompress_amd import _lib
    lib = _lib.load()
    ctx = lib.md_create(0, None)
    data = b"Zl.Higher.compress " * 400
    dst = ctypes.create_string_buffer(2 * len(data) + 64)
    n = ctypes.c_size_t()
    rc = lib.md_zl_higher_compress(ctx, 6, 1, 4096, data, len(data), dst, len(dst), ctypes.byref(n))
    assert rc == 0 and zlib.decompress(dst.raw[: n.value]) == data
    rc = lib.md_de_higher_compress(ctx, 4096, data, len(data), dst, len(dst), ctypes.byref(n))
    assert rc == 0 and zlib.decompress(dst.raw[: n.value], -15) == data
    assert lib.md_zl_higher_compress(ctx, 11, 1, 4096, data, len(data), dst, len(dst), ctypes.byref(n)) == -1
    assert lib.md_zl_higher_compress(ctx, 6, 1, 1000, data, len(data), dst, len(dst), ctypes.byref(n)) == -1
    lib.md_destroy(ctx)


def test_python_mirror(eng):
    from decompress_amd import de, zl
    data = b"mirror " * 3000
    z = zl.Higher.compress(data, level=6)
    assert zl.Higher.uncompress(z, len(data)) == data
    r = de.Higher.compress(data)
    assert de.Higher.uncompress(r, len(data)) == data
    # `Error (`Msg s): the reference's strings (lib/de.ml:702-730, lib/zl.ml:177-183)
    assert de.Higher.uncompress(r[:50], len(data)) == ("Error", "Unexpected end of input")
    assert de.Higher.uncompress(b"\x06", 10) == ("Error", "Invalid kind of block")
    assert zl.Higher.uncompress(b"\x79\x9c" + z[2:], len(data)) == ("Error", "Invalid Zlib header")
    assert zl.Higher.uncompress(z[:-1] + bytes([z[-1] ^ 1]), len(data)) == ("Error", "Invalid checksum")


def test_workspace_cap_and_release(eng):
    """md_set_option "deflate_workspace_cap_mib": a batch whose per-position workspace would exceed the cap goes through
    the kernels in slices of positions (here 64 KiB of every stream per launch, 33 KiB seen again, in groups of streams
    where even that is too much) - same bytes; "release_workspace" gives the grow-only scratch back and the next call
    grows it again."""
    import decompress_amd
    from decompress_amd import workloads
    bufs = [workloads.text(300 + i, 30000 + 2500 * i) for i in range(24)] + [b"", b"x"]
    want = eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=6)
    eng.set_option("deflate_workspace_cap_mib", 1)
    try:
        assert eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=6) == want
        eng.set_option("release_workspace", 1)
        assert eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=6) == want
    finally:
        eng.set_option("deflate_workspace_cap_mib", 0)
    eng.set_option("release_workspace", 1)
    assert eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=6) == want
    assert all(st == 0 and zlib.decompress(z) == b for b, (st, z, _) in zip(bufs, want))


@pytest.mark.parametrize("cap_mib", [2, 8, 24])
def test_batch_in_slices_of_positions(eng, oracle, cap_mib):
    """streams of every length around the slice boundaries (32 KiB multiples) through a capped workspace: zlib, gzip and
    raw DEFLATE under the three drivers, levels 1..9 - the bytes of the uncapped batch, which are the oracle's"""
    import random
    import decompress_amd
    from decompress_amd import workloads
    rng = random.Random(cap_mib)
    lens = [0, 1, 262, 32768, 65535, 65536, 65537, 98304, 131072, 131073, 200000, 262144 + 5, 400000, 655360, 1 << 20, 1200001]
    bufs = []
    for i, n in enumerate(lens):
        kind = i % 3
        bufs.append(workloads.text(900 + i, n) if kind == 0 else bytes(rng.getrandbits(3) for _ in range(n)) if kind == 1
                    else (workloads.text(i, 5000) * (n // 5000 + 1))[:n])
    cases = [(decompress_amd.FORMAT_ZLIB, 6, decompress_amd.DRIVER_ZL), (decompress_amd.FORMAT_GZIP, 4, decompress_amd.DRIVER_ZL),
             (decompress_amd.FORMAT_DEFLATE, 9, decompress_amd.DRIVER_HIGHER), (decompress_amd.FORMAT_DEFLATE, 1, decompress_amd.DRIVER_CLI),
             (decompress_amd.FORMAT_ZLIB, 3, decompress_amd.DRIVER_ZL), (decompress_amd.FORMAT_DEFLATE, 7, -1)]
    for fmt, level, driver in cases:
        matcher = None
        if driver < 0:  # lib/lz.ml's matcher under the Zl driver
            driver, matcher = decompress_amd.DRIVER_ZL, 1
        want = eng.deflate_many(bufs, fmt=fmt, level=level, driver=driver, matcher=matcher)
        eng.set_option("deflate_workspace_cap_mib", cap_mib)
        try:
            got = eng.deflate_many(bufs, fmt=fmt, level=level, driver=driver, matcher=matcher)
        finally:
            eng.set_option("deflate_workspace_cap_mib", 0)
        for i, (w, g) in enumerate(zip(want, got)):
            assert w == g, (fmt, level, driver, lens[i], w[0], g[0], len(w[1]), len(g[1]))
        if fmt == decompress_amd.FORMAT_ZLIB and level == 6:
            for b, (st, z, _) in zip(bufs, got):
                assert st == 0 and z == oracle.zl_deflate(b, 6)


def test_batch_in_slices_small_output_room(eng):
    """a stream whose output room runs out in a later slice reports it like the whole batch does"""
    import decompress_amd
    from decompress_amd import workloads
    bufs = [workloads.text(40 + i, 300000) for i in range(4)]
    caps = [2 * 300000, 50000, 90000, 8]
    want = eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=4, caps=caps)
    eng.set_option("deflate_workspace_cap_mib", 6)
    try:
        got = eng.deflate_many(bufs, fmt=decompress_amd.FORMAT_ZLIB, level=4, caps=caps)
    finally:
        eng.set_option("deflate_workspace_cap_mib", 0)
    assert [(st, len(z)) for st, z, _ in got] == [(st, len(z)) for st, z, _ in want]
    assert got[0] == want[0] and want[1][0] != 0


@pytest.mark.parametrize("piece", [3000, 65536, 200000])
def test_many_encoders_at_once(eng, oracle, piece):
    """md_def_batch_*: n streaming encoders advanced together, one launch of the kernels per round of pieces, windows kept in
    device memory.  Every encoder's bytes are the oracle's handed the same pieces (and md_def_*'s); output is there before
    the inputs end; streams of different lengths end in different rounds, empty ones at once; not fetching the output of a
    round does not lose it."""
    import random
    import decompress_amd
    from decompress_amd import workloads
    lib = eng.lib
    rng = random.Random(piece)
    lens = [0, 1, 5, 262, 4095, 4096, 40000, 65536, 65537, 100000, 131072, 300000, 70000, 33000, 98500, 250000]
    lens += [rng.randrange(0, 220000) for _ in range(24)]
    datas = []
    for k, n in enumerate(lens):
        kind = k % 4
        datas.append(workloads.text(500 + k, n) if kind == 0 else workloads.ascii_uniform(600 + k, n) if kind == 1
                     else bytes(rng.getrandbits(2) for _ in range(n)) if kind == 2 else (workloads.text(k, 2000) * (n // 2000 + 1))[:n])
    for fmt, level, queue in ((decompress_amd.FORMAT_ZLIB, 6, 4096), (decompress_amd.FORMAT_GZIP, 4, 4096), (decompress_amd.FORMAT_ZLIB, 1, 64),
                              (decompress_amd.FORMAT_DEFLATE, 9, 1024)):
        if piece < 65536 and level == 9:
            continue
        params = eng._params(level, queue, 0, True)
        n = len(datas)
        b = lib.md_def_batch_open(eng.ctx, fmt, ctypes.byref(params), n)
        assert b
        outs = [bytearray() for _ in range(n)]
        pos = [0] * n
        sent_end = [False] * n
        early = 0
        buf = ctypes.create_string_buffer(1 << 20)
        rounds = 0
        while not all(lib.md_def_batch_status(b, i) == 2 for i in range(n)):  # MD_END
            rounds += 1
            assert rounds < 400
            for i in range(n):
                if sent_end[i]:
                    continue
                chunk = datas[i][pos[i]:pos[i] + piece]
                pos[i] += len(chunk)
                assert lib.md_def_batch_src(b, i, chunk, len(chunk)) == 0
                if len(chunk) == 0:
                    sent_end[i] = True
            assert lib.md_def_batch_encode(b) == 0
            for i in range(n):
                st = lib.md_def_batch_status(b, i)
                assert st in (0, 2), (i, st)  # MD_AWAIT / MD_END
                if rounds % 3 == 0 and st != 2:
                    continue  # (left for later: the next encode keeps it)
                while lib.md_def_batch_pending(b, i):
                    k = lib.md_def_batch_out(b, i, buf, rng.choice((len(buf), 100, 4096)))
                    assert k > 0
                    outs[i] += buf.raw[:k]
                    early += not sent_end[i]
        if piece <= 65536:
            assert early > 0  # output before the end of the input
        for i in range(n):
            assert lib.md_def_batch_pending(b, i) == 0
            with oracle.src_piece(piece):
                if fmt == decompress_amd.FORMAT_ZLIB:
                    want = oracle.zl_deflate(datas[i], level, queue)
                elif fmt == decompress_amd.FORMAT_GZIP:
                    want = oracle.gz_deflate(datas[i], level=level)
                else:
                    want = oracle.deflate_raw(datas[i], level, queue)[0]
            assert bytes(outs[i]) == want, (fmt, level, i, len(datas[i]), len(outs[i]), len(want))
            if fmt == decompress_amd.FORMAT_ZLIB:
                assert zlib.decompress(bytes(outs[i])) == datas[i]
                assert lib.md_def_batch_checksum(b, i) == zlib.adler32(datas[i])
        # after the end: more input is refused, another encode is a no-op
        assert lib.md_def_batch_src(b, 0, b"x", 1) < 0
        assert lib.md_def_batch_encode(b) == 0
        lib.md_def_batch_close(b)
    # ... and the single encoder gives the same bytes for the same pieces
    got, _ = _encode_in_pieces(eng, decompress_amd.FORMAT_ZLIB, datas[11], piece, 6)
    with oracle.src_piece(piece):
        assert got == oracle.zl_deflate(datas[11], 6)
