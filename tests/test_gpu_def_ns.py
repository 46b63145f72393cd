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


def _fuzz_buffer(rng):
    """inputs in the spirit of fuzz/fuzz_ns.ml: bytes of a small alphabet, runs, copies of earlier pieces, noise"""
    kind = rng.randrange(5)
    n = rng.choice([0, 1, 5, 55, 56, 57, 300, 4000, 33000, 70000]) + rng.randrange(200)
    if kind == 0:
        return bytes(rng.randrange(256) for _ in range(min(n, 6000)))
    if kind == 1:
        return bytes(rng.choice(b"abcd") for _ in range(n))
    if kind == 2:
        return b"".join(bytes([rng.randrange(256)]) * rng.randrange(1, 400) for _ in range(n // 60 + 1))
    out = bytearray(rng.randrange(256) for _ in range(rng.randrange(1, 64)))
    while len(out) < n:  # LZ-like: literals and copies at random distances and lengths
        if rng.random() < 0.3:
            out += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 6)))
        else:
            d = rng.randrange(1, min(len(out), 40000) + 1)
            for _ in range(rng.randrange(3, 300)):
                out.append(out[-d])
    return bytes(out[:n]) if kind == 3 else bytes(out[:n]) * 2


def test_fuzz_vs_oracle(eng, oracle):
    """differential fuzz of md_de_def_ns_deflate / md_zl_def_ns_deflate (fuzz/fuzz_ns.ml's compress side): status and
    bytes equal the oracle's for random (input, level, output room); whatever came out also inflates back to the input
    through the Inf.Ns oracle AND through libz"""
    import decompress_amd
    rng = random.Random(20260928)
    for round_ in range(6):
        bufs = [_fuzz_buffer(rng) for _ in range(48)]
        level = rng.choice([1, 2, 3, 4, 4, rng.randrange(0, 13)])
        zl = round_ % 3 == 2
        res = eng.def_ns_many(bufs, level=level, fmt=decompress_amd.FORMAT_ZLIB if zl else decompress_amd.FORMAT_DEFLATE)
        for data, (st, z, _) in zip(bufs, res):
            ost, oz = oracle.def_ns(data, level, zl=zl)
            assert st == ost, (level, len(data), zl)
            if st == 0:
                assert z == oz, (level, len(data), zl)
                if level in (1, 2, 3, 4) or len(data) < 56 - 4 * level:  # (a stub level compresses to nothing: Ok 0)
                    raw = z[2:-4] if zl else z
                    assert zlib.decompress(raw, -15) == data
                    assert oracle.de_inflate(raw, len(data)) == (0, len(raw), data)


def test_small_output_room(eng, oracle):
    """the failure cases follow the output room exactly as the oracle's add_bits / flush_bits accounting does"""
    from decompress_amd import de, workloads
    data = workloads.text(9, 5000)
    full = oracle.def_ns(data, 4)[1]
    for room in sorted({0, 1, 7, 8, 9, 16, 100, len(full) - 9, len(full) - 1, len(full), len(full) + 1, len(full) + 7, len(full) + 8, len(full) + 64}):
        if room < 0:
            continue
        ost, oz = oracle.def_ns(data, 4, cap=room)
        verdict, z = de.Def.Ns.deflate(data, 4, dst_len=room)
        assert (verdict == "Ok") == (ost == 0) and (ost != 0 or z == oz), room
