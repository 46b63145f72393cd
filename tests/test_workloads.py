"""The workloads of SURVEY.md 8(d) (CPU): the corpus fixture is the reference's test/corpus, C2 is built from it."""
import zlib

from decompress_amd import workloads


def test_corpus_fixture():
    c = workloads.corpus()
    assert len(c) == 15 and sum(len(v) for v in c.values()) == 3263944
    assert len(c["obj1"]) == 21504 and len(c["book1"]) == 768771 and "rfc5322.txt" in c


def test_c2_composition():
    total = 3263944
    cat = b"".join(workloads.corpus().values())
    for i in (0, 2, 798, 4094):
        p = workloads.c2_plain(i, 262144)
        off = (i * 4099) % total
        assert len(p) == 262144 and p[:1000] == (cat + cat)[off:off + 1000]
    assert workloads.c2_plain(1, 4096) == workloads.markov_text(0xC2 + 1, 4096)
    s = workloads.c2_streams(6, nbytes=65536, workers=0, first=4)
    assert all(((z[2] >> 1) & 3) == 2 for z in s)  # dynamic Huffman first block
    assert zlib.decompress(s[0]) == workloads.c2_plain(4, 65536)


def test_markov_text():
    """C2's odd streams: order-2 Markov text over 64 ASCII symbols, seeded, zlib-6 ratio ~0.40"""
    a, b = workloads.markov_text(0xC2 + 1, 262144), workloads.markov_text(0xC2 + 3, 262144)
    assert len(a) == 262144 and a != b and a == workloads.markov_text(0xC2 + 1, 262144)
    assert a[:4096] == workloads.markov_text(0xC2 + 1, 4096)  # a prefix of the same chain
    assert set(a) <= set(workloads._MARKOV_ALPHABET) and len(set(a)) == 64
    for p in (a, b):
        assert 0.39 <= len(zlib.compress(p, 6)) / len(p) <= 0.41
