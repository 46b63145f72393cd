"""Parity of the HIP deflate path (through the C ABI) with the CPU oracle: identical
compressed bytes at the same level / queue / driver.  Needs an MI355X: `pytest -m gpu`."""
import ctypes
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


def test_order_free_head_reconstruction(oracle):
    """The look-ahead normally trusts same-address atomics of one wave to land in program order and
    checks it; force the fallback that rebuilds the hash heads without that assumption."""
    import decompress_amd
    from decompress_amd import workloads
    e = decompress_amd.Engine(0)
    e.set_option("deflate_test_flags", 1)
    bufs = [workloads.text(60, 90000), workloads.ascii_uniform(61, 150000), b"abc" * 30000, b"q" * 5]
    for b, (st, out, _) in zip(bufs, e.deflate_many(bufs, level=6)):
        assert st == 0 and out == oracle.deflate_raw(b, 6)[0]


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


def test_kats_through_gpu(eng):
    """test/test.ml:798-813: "abcde" is five literals and End -> one fixed/dynamic block."""
    st, out, _ = eng.deflate_many([b"abcde"], level=4, driver=1)[0]
    assert st == 0 and zlib.decompress(out, -15) == b"abcde"


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
