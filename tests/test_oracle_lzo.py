"""The oracle's LZO1X restatement (oracle/lzo.c <- lib/lzo.ml) against the reference's vector and
against minilzo — the C library the reference's own tests and fuzzers use as their LZO oracle
(test/test.ml:2033-2097, fuzz/fuzz_lzo.ml), built from the reference tree into oracle/_ref/."""
import random

import pytest

from tests import oracle_lib
from tests.conftest import golden_bytes, load_golden


@pytest.fixture(scope="module")
def minilzo():
    m = oracle_lib.load_minilzo()
    if m is None:
        pytest.skip("oracle/_ref/libminilzo.so not built (needs the reference tree)")
    return m


def _datasets():
    from decompress_amd import workloads
    rng = random.Random(21)
    out = []
    for n in (0, 1, 2, 3, 4, 5, 10, 19, 20, 21, 22, 31, 32, 33, 100, 237, 239, 240, 1000, 5000, 49151, 49152, 49153,
              49172, 49173, 60000, 100000, 131072, 200000):
        out += [workloads.text(n, n), workloads.ascii_uniform(n, n), bytes(rng.getrandbits(8) for _ in range(n)),
                bytes(n), (b"abcabcabd" * (n // 9 + 1))[:n], bytes([rng.randrange(3) for _ in range(n)])]
    return out


def test_reference_vectors(oracle, minilzo):
    """every decoder case of test/test_lzo.ml (34 vectors): the expected bytes, or an error where the reference
    expects one; minilzo — the reference's own cross-check — agrees on every valid one"""
    cases = load_golden("lzo.json")
    assert len(cases) >= 30
    for case in cases:
        src = golden_bytes(case["src"])
        if case["status"] == 0:
            want = golden_bytes(case["out"])
            st, out = oracle.lzo_uncompress(src, len(want))
            assert (st, out) == (0, want), case["name"]
            assert minilzo.decompress(src, len(want)) == (0, want), case["name"]
        else:
            st, _ = oracle.lzo_uncompress(src, 1 << 16)
            assert st != 0, case["name"]


def test_cross_decompression(oracle, minilzo):
    """what the reference tests: either side decodes what the other emits"""
    for d in _datasets():
        st, z = oracle.lzo_compress(d)
        assert st == 0
        assert minilzo.decompress(z, len(d)) == (0, d)
        assert oracle.lzo_uncompress(minilzo.compress(d), len(d)) == (0, d)
        assert oracle.lzo_uncompress(z, len(d)) == (0, d)


def test_compress_bytes_vs_minilzo(oracle, minilzo):
    """lib/lzo.ml is a transcription of lzo1x_1_compress: the bytes agree except where the OCaml
    text deviates — the match extension stops at the last 20 bytes of a 48 KiB chunk instead of
    running into them (lib/lzo.ml:616-631 vs minilzo's m_len loop), and a 238-byte incompressible
    input takes the long first-byte form (`len < 238`, lib/lzo.ml:565)."""
    same = diff = 0
    for d in _datasets():
        z = oracle.lzo_compress(d)[1]
        zm = minilzo.compress(d)
        if z == zm:
            same += 1
        else:
            diff += 1
            assert len(z) >= len(zm)  # the deviations only ever cost bytes
    assert same > 8 * diff


def test_238_byte_quirk(oracle, minilzo):
    rng = random.Random(3)
    d = bytes(rng.getrandbits(8) for _ in range(238))
    z, zm = oracle.lzo_compress(d)[1], minilzo.compress(d)
    assert zm[0] == 17 + 238 and z[:2] == bytes([0, 238 - 18]) and len(z) == len(zm) + 1
    assert oracle.lzo_uncompress(z, 238) == (0, d)


def test_errors(oracle):
    d = b"hello hello hello hello hello hello hello, said the parrot" * 40
    z = oracle.lzo_compress(d)[1]
    assert oracle.lzo_uncompress(z, len(d)) == (0, d)
    assert oracle.lzo_uncompress(z, len(d) - 1)[0] == 16          # output not large enough
    assert oracle.lzo_uncompress(b"", 10)[0] == 1                   # Unexpected end of input
    assert oracle.lzo_uncompress(b"\x10", 10)[0] == 15              # No dictionary at offset 0 available
    assert oracle.lzo_uncompress(z[:-3], len(d))[0] in (1, 16)      # end marker cut
    assert oracle.lzo_uncompress(b"\x00\x00\x00", 100)[0] == 14     # count runs off the input: Invalid input
    assert oracle.lzo_uncompress(b"\x40\x00", 100)[0] in (1, 16)    # match before any output
    assert oracle.lzo_compress(d, cap=20)[0] == 16                  # lzo: output is not large enough
    for cut in range(len(z)):
        st, out = oracle.lzo_uncompress(z[:cut], len(d))
        assert st != 0 and out == b""
