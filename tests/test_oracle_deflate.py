"""The CPU deflate oracle (oracle/de_deflate.c) against the reference's own
encoder vectors (tests/golden/deflate_kat.json <- test/test.ml) and, for
validity, against libz and the inflate oracle.

Deflate bytes are pinned ONLY by these KATs; everything else here is
round-trip (the reference's own test strategy)."""
import random
import zlib

import pytest

from tests.conftest import load_golden

KAT = load_golden("deflate_kat.json")


def _case(name):
    return next(c for c in KAT if c["name"] == name)


@pytest.mark.parametrize("name", ["huffman_length_extra", "flat"])
def test_encoder_bytes(oracle, name):
    c = _case(name)
    out = oracle.encode_cmds(c["cmds"], c["kind"])
    assert out == bytes.fromhex(c["out"]), c["ref"]
    rc, used, plain = oracle.de_inflate(out, 65536)
    assert (rc, used, plain) == (0, len(out), bytes.fromhex(c["inflated"]))


@pytest.mark.parametrize("name", ["tree_0", "tree_rfc5322_corpus"])
def test_huffman_trees(oracle, name):
    c = _case(name)
    mc, lens, codes, _ = oracle.tree_make(c["length"], c["freqs"])
    for sym, l in c["lengths"].items():
        assert lens[int(sym)] == l, (c["ref"], sym)
    for sym, code in c["codes"].items():
        assert codes[int(sym)] == code, (c["ref"], sym)
    if name == "tree_0":  # the only coded symbols are 0 (forced by pkzip) and 256
        assert [i for i, l in enumerate(lens) if l] == [0, 256]


def test_tree_make_mutates_histogram(oracle):
    """H2: pkzip writes freqs[0] <- 1, internal nodes land in freqs[length..] (lib/de.ml:1863-1874, 2037)."""
    f = [0] * 573
    f[256] = 1
    _, _, _, after = oracle.tree_make(286, f)
    assert after[0] == 1 and after[256] == 1 and after[286] == 2


def test_lz77_abcde(oracle):
    c = _case("lz77_1")
    assert oracle.lz77_cmds(bytes.fromhex(c["src"]), level=4) == c["cmds"]


def _datasets():
    from decompress_amd import workloads
    rng = random.Random(5)
    return {
        "empty": b"", "one": b"x", "abcde": b"abcde", "runs": b"a" * 70000 + b"b" * 300,
        "text": workloads.text(3, 150000), "ascii": workloads.ascii_uniform(4, 70000),
        "rand": bytes(rng.getrandbits(8) for _ in range(66000)),
        "zeros": bytes(200000),
    }


@pytest.mark.parametrize("driver", [0, 1, 2], ids=["Zl.Def", "De.Higher", "CLI"])
def test_roundtrip_all_levels(oracle, driver):
    """test/test_deflate.ml:19-120 (levels 0-9) — compress, inflate with libz AND the oracle."""
    for name, data in _datasets().items():
        for level in range(10):
            if driver == 1 and level != 4:
                continue  # H6: De.Higher.compress has no ?level
            for q in (16, 4096):
                raw, adler = oracle.deflate_raw(data, level, q, driver)
                d = zlib.decompressobj(-15)
                assert d.decompress(raw) == data and d.eof and not d.unused_data, (name, level, q)
                assert adler == zlib.adler32(data)
                rc, used, plain = oracle.de_inflate(raw, len(data))
                assert (rc, used, plain) == (0, len(raw), data)


def test_zlib_frame(oracle):
    """lib/zl.ml:511-522: header 0x78xx with FLEVEL 0/1/2/3 and Adler-32 trailer."""
    data = b"hello hello hello hello " * 100
    heads = {0: "7801", 1: "785e", 5: "785e", 6: "789c", 7: "78da", 9: "78da"}
    for level, h in heads.items():
        z = oracle.zl_deflate(data, level)
        assert z[:2].hex() == h
        assert zlib.decompress(z) == data
        assert oracle.zl_inflate(z, len(data)) == (0, len(z), data)


def test_level0_blocks_are_queue_sized(oracle):
    """A.3: Flat blocks carry min(queue length, 65535) literals after dropping the EOB —
    4095-byte stored blocks with the default 4096-entry queue."""
    data = bytes(range(256)) * 64
    raw, _ = oracle.deflate_raw(data, 0, 4096, 0)
    assert raw[0] == 0 and raw[1:3] == (4095).to_bytes(2, "little")
    assert zlib.decompress(raw, -15) == data


def test_block_boundary_is_queue_capacity(oracle):
    """H1/H5: with the Zl driver the first block holds capacity-1 commands; different
    queue sizes give different (valid) streams."""
    from decompress_amd import workloads
    data = workloads.text(9, 60000)
    a, _ = oracle.deflate_raw(data, 6, 256, 0)
    b, _ = oracle.deflate_raw(data, 6, 4096, 0)
    assert a != b
    assert zlib.decompress(a, -15) == zlib.decompress(b, -15) == data


def test_drivers_differ(oracle):
    """H5: three drivers, three bitstreams for the same commands."""
    from decompress_amd import workloads
    data = workloads.text(11, 50000)
    outs = {d: oracle.deflate_raw(data, 4, 1024, d)[0] for d in (0, 1, 2)}
    assert len({bytes(v) for v in outs.values()}) == 3
    for v in outs.values():
        assert zlib.decompress(v, -15) == data


def test_cli_driver_queue_full(oracle):
    """the CLI driver's extra end-of-block push into a full queue: De.Queue.Full in the reference"""
    assert oracle.deflate_raw(b"geg", 6, 4, 2)[0] is None
    assert oracle.deflate_raw(b"geg", 6, 8, 2)[0] is not None
    assert oracle.deflate_raw(b"geg", 6, 4, 0)[0] is not None


# ---- the reference's encoder-built cases (tests/golden/encode_cases.json <- test/test_ns.ml, test/test.ml) ----
ENC = load_golden("encode_cases.json")


@pytest.mark.parametrize("case", ENC, ids=[c["name"] for c in ENC])
def test_encode_cases(oracle, case):
    """Def.encode driven exactly as the reference's test drives it: the answers (`Ok / `Block) it demands, the bytes
    where it pins them, and what inflating the result must give."""
    res = oracle.def_script(case["ops"], case["queue_len"])
    assert res is not None, case["ref"]
    z, rcs = res
    assert rcs == case["rcs"], case["ref"]
    if "out" in case:
        assert z == bytes.fromhex(case["out"])
    rc, consumed, out = oracle.de_inflate(z, case["dst_cap"])
    assert rc == case["status"], case["ref"]
    if rc == 0:
        assert out == bytes.fromhex(case["plain"])
        if case["decoder"] == "ns":  # Ok (De.bigstring_length src, String.length expected)
            assert consumed == len(z)
        assert zlib.decompressobj(-15).decompress(z) == out


def test_input_in_pieces(oracle):
    """orc_set_src_piece: De.Lz77 handed the input p bytes per `Await (lib/de.ml:4181-4188, :4294-4342).  The stream stays
    valid and - away from the corner where a candidate sits exactly at the window base when fill_window slides - the same
    as for the whole input at once."""
    import zlib
    from decompress_amd import workloads
    data = workloads.text(12, 150000)
    whole = oracle.zl_deflate(data, 6)
    for piece in (1, 263, 4096, 65536, 149999, 150000, 1 << 20):
        with oracle.src_piece(piece):
            got = oracle.zl_deflate(data, 6)
            gz = oracle.gz_deflate(data, level=4)
        assert zlib.decompress(got) == data and got == whole
        assert zlib.decompress(gz, 31) == data
    assert oracle.zl_deflate(data, 6) == whole  # (the setting does not outlive the block)
