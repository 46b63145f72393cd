"""The CPU oracle of De.Def.Ns / Zl.Def.Ns (oracle/de_def_ns.c) against the two compressed-byte vectors the reference
holds for this path (tests/golden/def_ns.json <- test/test_ns.ml:1189-1222) and the properties its other tests check
(test/test_ns.ml:1098-1177: compress, then inflate gives the input back; fuzz/fuzz_ns.ml).  Beyond the two vectors
the byte parity of this oracle is unpinned (oracle/de_def_ns.c header)."""
import random
import zlib

import pytest

from tests.conftest import load_golden

KAT = load_golden("def_ns.json")


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_reference_vectors(oracle, case):
    st, z = oracle.def_ns(bytes.fromhex(case["src"]), case["level"], cap=65536)
    assert (st, z) == (0, bytes.fromhex(case["out"])), case["ref"]


def test_round_trip_corpus(oracle):
    from decompress_amd import workloads
    for name, data in workloads.corpus().items():
        st, z = oracle.def_ns(data, 4)  # the reference's default level
        assert st == 0 and zlib.decompress(z, -15) == data, name
        st, zz = oracle.def_ns(data, 4, zl=True)
        assert st == 0 and zz[2:-4] == z and zlib.decompress(zz) == data
        assert zz[:2] == bytes([0x78, 0x5e])  # FLEVEL 1 for level 4 (lib/zl.ml:605-606)


def test_levels_and_edges(oracle):
    rng = random.Random(3)
    text = b" ".join(bytes(rng.choice(b"abcdefgh") for _ in range(rng.randrange(1, 9))) for _ in range(4000))
    sizes = set()
    for level in (1, 2, 3, 4):
        st, z = oracle.def_ns(text, level)
        assert st == 0 and zlib.decompress(z, -15) == text
        sizes.add(len(z))
    assert len(sizes) > 1
    for level in range(5, 13):  # compress_lazy is a stub upstream: Ok 0
        assert oracle.def_ns(text, level) == (0, b"")
    assert oracle.def_ns(text, 13)[0] == -1 and oracle.def_ns(text, -1)[0] == -1
    assert oracle.def_ns(text, 0)[0] == 2  # write_uncompressed_blocks never advances: Unexpected_end_of_output
    for n in (0, 1, 39, 40, 51, 52, 55):  # shorter than 56 - 4 * level: one stored block
        d = text[:n]
        st, z = oracle.def_ns(d, 1)
        assert st == 0 and zlib.decompress(z, -15) == d
        assert (z[0] == 1) == (n < 52)
    assert oracle.def_ns(text, 4, cap=7) == (0, b"")  # dst shorter than the end padding: Ok 0
    assert oracle.def_ns(text, 4, cap=200)[0] == 2
    noise = bytes(rng.randrange(256) for _ in range(20000))
    assert oracle.def_ns(noise, 4)[0] == 2  # incompressible: the uncompressed block type is chosen, which cannot end well upstream
    big = bytes(rng.choice(b"ab") for _ in range(700000))  # several blocks (soft maximum 300000)
    st, z = oracle.def_ns(big, 3)
    assert st == 0 and zlib.decompress(z, -15) == big
