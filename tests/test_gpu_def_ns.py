"""De.Def.Ns / Zl.Def.Ns on the GPU (csrc/deflate_ns.hip) through the C ABI: the reference's two compressed-byte
vectors (test/test_ns.ml:1189-1222), byte equality with the oracle on the reference's corpus and on synthetic inputs at
every level, the statuses of the stub levels / level 0 / short outputs, and the round trips the reference's own tests
run (test/test_ns.ml:1098-1177)."""
import random
import zlib

import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
KAT = load_golden("def_ns.json")


@pytest.fixture(scope="module")
def eng():
    import decompress_amd
    return decompress_amd.Engine(0)


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_reference_vectors(case):
    from decompress_amd import de
    verdict, z = de.Def.Ns.deflate(bytes.fromhex(case["src"]), case["level"], dst_len=65536)
    assert (verdict, z) == ("Ok", bytes.fromhex(case["out"])), case["ref"]


def test_corpus_bytes_equal_oracle(eng, oracle):
    """every file of the reference's test/corpus at levels 1 and 4, raw and zlib-framed, as one batch each"""
    import decompress_amd
    from decompress_amd import workloads
    files = list(workloads.corpus().values())
    for level in (1, 4):
        res = eng.def_ns_many(files, level=level)
        for data, (st, z, adler) in zip(files, res):
            assert (st, z) == oracle.def_ns(data, level)
            assert zlib.decompress(z, -15) == data and adler == zlib.adler32(data)
    for data, (st, z, _) in zip(files[:4], eng.def_ns_many(files[:4], level=4, fmt=decompress_amd.FORMAT_ZLIB)):
        assert (st, z) == oracle.def_ns(data, 4, zl=True) and zlib.decompress(z) == data


def test_levels_sizes_and_statuses(eng, oracle):
    from decompress_amd import de, zl, workloads
    rng = random.Random(11)
    text = workloads.text(77, 300000)
    bufs = [b"", b"a", text[:39], text[:40], text[:51], text[:52], text[:55], text[:56], text[:57], text[:1000], text[:32768],
            text[:32769], text[:65536 + 5], text, bytes(70000), bytes(rng.choice(b"ab") for _ in range(650000)),
            workloads.ascii_uniform(5, 50000), bytes(rng.randrange(256) for _ in range(30000))]
    for level in (0, 1, 2, 3, 4, 5, 9, 12):
        for data, (st, z, _) in zip(bufs, eng.def_ns_many(bufs, level=level)):
            ost, oz = oracle.def_ns(data, level)
            assert st == ost, (level, len(data))
            if st == 0:
                assert z == oz, (level, len(data))
    assert de.Def.Ns.deflate(text, 13) == ("Error", "Invalid_compression_level")
    assert de.Def.Ns.deflate(text, 4, dst_len=7) == ("Ok", b"")
    assert de.Def.Ns.deflate(text, 4, dst_len=300) == ("Error", "Unexpected_end_of_output")
    assert de.Def.Ns.deflate(text, 0) == ("Error", "Unexpected_end_of_output")
    assert zl.Def.Ns.deflate(text[:10], 4, dst_len=1) == ("Error", "Unexpected_end_of_output")
    verdict, z = zl.Def.Ns.deflate(text, 6)
    assert verdict == "Ok" and z == b""[:0] + z and oracle.def_ns(text, 6, zl=True) == (0, z)  # stub level inside the frame


def test_round_trip_through_gpu_inflate(eng):
    """compress with Def.Ns, inflate with Inf.Ns, both on the GPU (test/test_ns.ml:1098-1136)"""
    from decompress_amd import workloads
    bufs = [workloads.text(200 + i, 100000 + 7777 * i) for i in range(16)]
    res = eng.def_ns_many(bufs, level=4)
    assert all(st == 0 for st, _, _ in res)
    back = eng.inflate_many([z for _, z, _ in res], [len(b) for b in bufs])
    for b, (st, used, out, _), (_, z, _) in zip(bufs, back, res):
        assert (st, used, out) == (0, len(z), b)
