"""LZO1X on the GPU (csrc/lzo_kernels.hip) against the oracle's restatement of lib/lzo.ml and, as the
reference's own tests do, against minilzo (oracle/_ref).  Needs an MI355X: `pytest -m gpu`."""
import random

import pytest

from tests import oracle_lib
from tests.conftest import golden_bytes, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import decompress_amd
    return decompress_amd.Engine(0)


def _datasets():
    from decompress_amd import workloads
    rng = random.Random(21)
    out = []
    for n in (0, 1, 2, 3, 4, 5, 10, 19, 20, 21, 22, 31, 32, 33, 100, 237, 238, 239, 1000, 5000, 49151, 49152, 49153,
              49172, 49173, 60000, 100000, 131072, 200000):
        out += [workloads.text(n, n), workloads.ascii_uniform(n, n), bytes(rng.getrandbits(8) for _ in range(n)),
                bytes(n), (b"abcabcabd" * (n // 9 + 1))[:n], bytes([rng.randrange(3) for _ in range(n)])]
    return out


def test_reference_vectors(eng, oracle):
    """the 34 decoder cases of test/test_lzo.ml in one batch: expected bytes, or the oracle's error"""
    cases = load_golden("lzo.json")
    srcs = [golden_bytes(c["src"]) for c in cases]
    caps = [len(golden_bytes(c["out"])) if c["status"] == 0 else 1 << 16 for c in cases]
    for c, src, cap, (st, out) in zip(cases, srcs, caps, eng.lzo_many(False, srcs, caps)):
        if c["status"] == 0:
            assert (st, out) == (0, golden_bytes(c["out"])), c["name"]
        else:
            assert st != 0 and st == oracle.lzo_uncompress(src, cap)[0], c["name"]


def test_compress_equals_oracle_and_round_trips(eng, oracle):
    from decompress_amd import lzo
    data = _datasets()
    res = eng.lzo_many(True, data, [lzo.max_compressed_length(len(d)) for d in data])
    zs = []
    for d, (st, z) in zip(data, res):
        ost, oz = oracle.lzo_compress(d)
        assert (st, z) == (ost, oz) and st == 0, len(d)
        zs.append(z)
    back = eng.lzo_many(False, zs, [len(d) for d in data])
    for d, (st, out) in zip(data, back):
        assert (st, out) == (0, d)
    m = oracle_lib.load_minilzo()
    if m is not None:  # cross-decompression, test/test.ml:2067-2097 / fuzz/fuzz_lzo.ml
        mz = [m.compress(d) for d in data]
        for d, (st, out) in zip(data, eng.lzo_many(False, mz, [len(d) for d in data])):
            assert (st, out) == (0, d)
        for d, z in zip(data, zs):
            assert m.decompress(z, len(d)) == (0, d)


def test_errors_equal_oracle(eng, oracle):
    rng = random.Random(5)
    d = b"hello hello hello hello hello hello hello, said the parrot; " * 60
    z = oracle.lzo_compress(d)[1]
    cases, caps = [], []
    for cut in range(len(z)):
        cases.append(z[:cut]); caps.append(len(d))
    for _ in range(400):
        b = bytearray(z)
        for _ in range(rng.randrange(1, 3)):
            b[rng.randrange(len(b))] = rng.getrandbits(8)
        cases.append(bytes(b)); caps.append(len(d) + rng.choice((0, 0, 7, 4000)))
    for _ in range(300):
        n = rng.choice((1, 2, 3, 5, 9, 30, 200))
        cases.append(bytes(rng.getrandbits(8) for _ in range(n))); caps.append(rng.choice((0, 10, 1000, 70000)))
    for cap in (0, 1, len(d) - 1, len(d)):
        cases.append(z); caps.append(cap)
    res = eng.lzo_many(False, cases, caps)
    seen = set()
    for k, (c, cap, (st, out)) in enumerate(zip(cases, caps, res)):
        ost, oout = oracle.lzo_uncompress(c, cap)
        assert (st, out) == (ost, oout), (k, len(c), cap, st, ost)
        seen.add(st)
    assert seen >= {0, 1, 16}
    # compress into too small a buffer: "lzo: output is not large enough"
    for cap in (0, 3, 20, len(z) - 1, len(z)):
        assert eng.lzo_many(True, [d], [cap])[0] == oracle.lzo_compress(d, cap=cap)


def test_c_abi_single_buffer(eng):
    from decompress_amd import lzo
    d = b"Salut les copains!"  # test/test.ml:2067-2079
    assert lzo.uncompress(lzo.compress(d), 128) == ("Ok", d)


def test_corpus_files(eng, oracle):
    """the reference's test/corpus through Lzo.compress / uncompress (test/test.ml:2067-2097's shape on real files): compressed
    bytes = the oracle's, minilzo reads them, and both minilzo's and our streams come back through the batched decoder - whole
    files (21 KB .. 769 KB: the decoder's windows, long literal runs, long matches, the slow path at the ends)"""
    from decompress_amd import lzo, workloads
    files = list(workloads.corpus().items())
    bufs = [b for _, b in files]
    res = eng.lzo_many(True, bufs, [lzo.max_compressed_length(len(b)) for b in bufs])
    zs = []
    for (name, b), (st, z) in zip(files, res):
        assert (st, z) == oracle.lzo_compress(b), name
        zs.append(z)
    m = oracle_lib.load_minilzo()
    srcs = zs + ([m.compress(b) for b in bufs] if m is not None else [])
    want = bufs + (bufs if m is not None else [])
    for k, (b, (st, out)) in enumerate(zip(want, eng.lzo_many(False, srcs, [len(b) for b in want]))):
        assert (st, out) == (0, b), k
    if m is not None:
        for b, z in zip(bufs, zs):
            assert m.decompress(z, len(b)) == (0, b)
    # exact room, one byte short, a cut stream: the oracle's answers
    for b, z in list(zip(bufs, zs))[:4]:
        cases = [(z, len(b) - 1), (z[:len(z) * 2 // 3], len(b)), (z[:-1], len(b))]
        for (src, cap), (st, out) in zip(cases, eng.lzo_many(False, [c[0] for c in cases], [c[1] for c in cases])):
            assert (st, out) == oracle.lzo_uncompress(src, cap)


def _planted(rng, n, plants):
    """n bytes without repeats of 4 bytes (a counter in base 251 under a permutation), then `plants`: (at, distance, length)
    copies of earlier bytes - the matches the compressor has to find (or not) where the opcode forms change"""
    perm = list(range(256))
    rng.shuffle(perm)
    d = bytearray()
    k = 0
    while len(d) < n:
        d += bytes((perm[(k // 251 ** j) % 251] for j in range(4)))
        k += 1
    d = d[:n]
    for at, dist, ln in plants:
        if at - dist >= 0 and at + ln <= n:
            d[at:at + ln] = d[at - dist:at - dist + ln] if dist >= ln else bytes(d[at - dist + (j % dist)] for j in range(ln))
    return bytes(d)


def test_compressor_step_edge_cases(eng, oracle):
    """round 6's step behind a match (csrc/lzo_kernels.hip near_step) where its cases change: match lengths around 12 and 16
    (decided in the hit lane's registers or by the 8-bytes-per-lane loop), 33 / 34 and 9 / 10 (a length that goes on over more
    bytes), offsets around 2 KiB, 16 KiB and the 48 KiB chunk (M2 / M3 / M4 forms), runs of 0..9, 16, 18, 19 and 300
    literals between matches (the count in the match, its own byte, the long form), periodic inputs (several probes of a
    step on one dictionary slot: the byte table in LDS sends the step to the general code), sizes around the chunk.
    Bytes = the oracle's, the round trip and minilzo's decoder give the input back."""
    from decompress_amd import lzo
    rng = random.Random(606)
    data = []
    for ln in (4, 5, 8, 9, 10, 11, 12, 13, 15, 16, 17, 19, 20, 27, 28, 33, 34, 35, 264, 265, 523, 524, 525, 600, 2000):
        for dist in (1, 2, 3, 4, 7, 8, 9, 2047, 2048, 2049, 16383, 16384, 16385, 40000, 49151):
            n = 60000
            plants = [(at, dist, ln) for at in range(50000, 50000 + 12 * (ln + 9), ln + 9)]  # runs of 9 literals between them
            data.append(_planted(rng, n, plants))
    for gap in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 18, 19, 20, 31, 32, 33, 300):
        plants, at = [], 20000
        for _ in range(40):
            plants.append((at, 5000 + rng.randrange(3), rng.choice((4, 6, 8, 11, 12, 16, 40))))
            at += plants[-1][2] + gap
        data.append(_planted(rng, 40000, plants))
    for period in range(1, 12):
        for n in (100, 5000, 49152 + 37):
            unit = bytes(rng.getrandbits(8) for _ in range(period))
            data.append((unit * (n // period + 1))[:n])
            data.append((unit * 40 + bytes(rng.getrandbits(8) for _ in range(23))) * (n // (40 * period + 23) + 1))
    for n in (49152 - 21, 49152 - 20, 49152 - 1, 49152, 49152 + 1, 49152 + 19, 49152 + 20, 49152 + 21, 2 * 49152, 2 * 49152 + 5, 3 * 49152 + 31):
        data.append(_planted(rng, n, [(n - 30, 9000, 25), (49140, 300, 30), (49150, 20000, 8), (98300, 47000, 12)]))
        data.append(bytes(n))
    res = eng.lzo_many(True, data, [lzo.max_compressed_length(len(d)) for d in data])
    zs = []
    for k, (d, (st, z)) in enumerate(zip(data, res)):
        assert (st, z) == oracle.lzo_compress(d), (k, len(d))
        zs.append(z)
    for k, (d, (st, out)) in enumerate(zip(data, eng.lzo_many(False, zs, [len(d) for d in data]))):
        assert (st, out) == (0, d), (k, len(d))
    m = oracle_lib.load_minilzo()
    if m is not None:
        for d, z in zip(data[::7], zs[::7]):
            assert m.decompress(z, len(d)) == (0, d)
