"""lib/lz.ml's match finder (`Lz.state` / `Lz.compress`).  The reference holds no
test or caller for it; lib/lz.ml is a transcription of zlib's deflate_slow (3-byte rolling hash,
memLevel 8, level table _4.._9 = zlib's 4..9, TOO_FAR 4096), so its LZ77 decisions must be the
ones libz itself makes on the same input: the oracle's token stream is compared with the token
stream of libz's own output (block boundaries differ — the queue is 4096 commands, libz's buffer
16383 — tokens do not)."""
import random
import zlib

import pytest

from tests.deflate_tokens import tokens


def _lz_tokens(data):
    return [t for t in tokens(data) if t[0] != "B"]


def _datasets():
    from decompress_amd import workloads
    rng = random.Random(12)
    return {
        "text": workloads.text(31, 120000), "ascii": workloads.ascii_uniform(32, 50000),
        "runs": b"a" * 40000 + b"ab" * 3000 + bytes(70000), "small": b"hello hello hello hello",
        "rand": bytes(rng.getrandbits(8) for _ in range(30000)),
        "text2": workloads.text(33, 200000),
    }


@pytest.mark.parametrize("level", [4, 5, 6, 7, 8, 9])
def test_lz_decisions_equal_libz(oracle, level):
    for name, data in _datasets().items():
        mine, adler = oracle.deflate_raw(data, level=level, matcher=1)
        assert zlib.decompress(mine, -15) == data and adler == zlib.adler32(data)
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8)
        ref = co.compress(data) + co.flush()
        if any(t[0] == "B" and t[1] == 0 for t in tokens(ref)):
            continue  # libz fell back to stored blocks (incompressible input): its decisions are not in the stream
        a, b = _lz_tokens(mine), _lz_tokens(ref)
        assert len(a) == len(b), (name, len(a), len(b))
        for k, (x, y) in enumerate(zip(a, b)):
            assert x == y, (name, k, x, y)


def test_lz_low_levels_are_level_4(oracle):
    """Lz.state: levels 0..4 all use _4 and there is no copy mode (lib/lz.ml:535)"""
    from decompress_amd import workloads
    data = workloads.text(34, 60000)
    want = oracle.deflate_raw(data, level=4, matcher=1)[0]
    for level in (0, 1, 2, 3):
        assert oracle.deflate_raw(data, level=level, matcher=1)[0] == want


def test_lz_differs_from_de_lz77(oracle):
    """the two match finders hash differently (3 bytes rolled vs 4 bytes multiplied): same input,
    different — both valid — streams"""
    from decompress_amd import workloads
    data = workloads.text(35, 80000)
    a = oracle.deflate_raw(data, level=6, matcher=0)[0]
    b = oracle.deflate_raw(data, level=6, matcher=1)[0]
    assert a != b and zlib.decompress(a, -15) == zlib.decompress(b, -15) == data


def test_lz_edges(oracle):
    for data in (b"", b"a", b"ab", b"abc", b"abcd", b"aaaa", b"abcabcabc"):
        for drv in (0, 1, 2):
            z = oracle.deflate_raw(data, level=6, driver=drv, matcher=1)[0]
            assert zlib.decompress(z, -15) == data
