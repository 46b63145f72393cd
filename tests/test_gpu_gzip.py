"""GZip framing on the GPU (csrc/gz_kernels.hip + the DEFLATE kernels) against the oracle's
restatement of lib/gz.ml and the reference's own vectors.  Needs an MI355X: `pytest -m gpu`."""
import gzip
import io
import random
import zlib

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import decompress_amd
    return decompress_amd.Engine(0)


def test_reference_vectors(eng):
    """test/test.ml:1659-1989 through Gz.Higher.uncompress's mirror"""
    from decompress_amd import gz
    for case in load_golden("gzip.json"):
        r = gz.Higher.uncompress(bytes.fromhex(case["src"]), 4096)
        if "error" in case:
            assert r == ("Error", case["error"]), case["name"]
            continue
        assert r[0] == "Ok" and r[2] == bytes.fromhex(case["out"]), case["name"]
        if "filename" in case:
            assert r[1]["filename"] == bytes.fromhex(case["filename"])
        if "extra_key" in case:
            assert gz.extra(r[1]["extra"], bytes.fromhex(case["extra_key"])) == bytes.fromhex(case["extra_value"])


def _frames(oracle):
    from decompress_amd import workloads
    rng = random.Random(9)
    out = []
    for i in range(24):
        data = workloads.text(100 + i, rng.randrange(0, 90000)) if i % 3 else bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 5000)))
        if i % 2:
            buf = io.BytesIO()
            with gzip.GzipFile(filename="f%d.txt" % i if i % 4 == 1 else "", mode="wb", fileobj=buf, mtime=i, compresslevel=1 + i % 9) as f:
                f.write(data)
            out.append((buf.getvalue(), data))
        else:
            out.append((oracle.gz_deflate(data, level=i % 10, mtime=i * 77, os=(3, 11, 255)[i % 3], hcrc=i % 4 == 0,
                                          name=b"n%d" % i if i % 3 == 0 else None,
                                          comment=b"c" * (i % 7) if i % 6 == 0 else None), data))
    return out


def test_inflate_batch_equals_oracle(eng, oracle):
    import decompress_amd
    frames = _frames(oracle)
    res = eng.inflate_many([f for f, _ in frames], [len(d) + 8 for _, d in frames], decompress_amd.FORMAT_GZIP)
    for (f, d), (st, used, out, crc) in zip(frames, res):
        ost, oused, oout, _ = oracle.gz_inflate(f, len(d) + 8)
        assert (st, used, out) == (ost, oused, oout) == (0, len(f), d)
        assert crc == zlib.crc32(d)


def test_errors_equal_oracle(eng, oracle):
    """every truncation and a byte flip at every position of a small member, one batch"""
    import decompress_amd
    good = oracle.gz_deflate(b"some text, some text, some text; " * 3, level=6, name=b"nm", comment=b"cm", hcrc=True)
    cases = [good[:k] for k in range(len(good))]
    for k in range(len(good)):
        b = bytearray(good)
        b[k] ^= 0x40
        cases.append(bytes(b))
    cases.append(good)
    caps = [256] * len(cases)
    res = eng.inflate_many(cases, caps, decompress_amd.FORMAT_GZIP)
    seen = set()
    for c, (st, used, out, _) in zip(cases, res):
        ost, oused, oout, _ = oracle.gz_inflate(c, 256)
        assert (st, used) == (ost, oused), (len(c), st, ost)
        if st in (0, 9, 12):  # output fully produced
            assert out == oout
        seen.add(st)
    assert {0, 1, 9, 10, 11, 12} <= seen
    # too small an output buffer
    assert eng.inflate_many([good], [5], decompress_amd.FORMAT_GZIP)[0][0] == 2


def test_deflate_equals_oracle(eng, oracle):
    import decompress_amd
    from decompress_amd import workloads
    bufs = [b"", b"foo", b"foo & bar", workloads.text(5, 70000), workloads.ascii_uniform(6, 40000), bytes(3000)]
    for hdr in (dict(), dict(mtime=0x5e53f12d, os=3, filename=b"foo"), dict(hcrc=True, filename=b"foo.gz", comment=b"x y z"),
                dict(ascii=True, os=11, hcrc=True)):
        for level in (0, 1, 4, 6, 9):
            eng.gz_set_header(**hdr)
            res = eng.deflate_many(bufs, decompress_amd.FORMAT_GZIP, level=level)
            for b, (st, out, crc) in zip(bufs, res):
                want = oracle.gz_deflate(b, level=level, mtime=hdr.get("mtime", 0), os=hdr.get("os", 3),
                                         hcrc=hdr.get("hcrc", False), ascii=hdr.get("ascii", False),
                                         name=hdr.get("filename"), comment=hdr.get("comment"))
                assert st == 0 and out == want, (hdr, level, len(b))
                assert crc == zlib.crc32(b)
                if not hdr.get("hcrc"):
                    assert gzip.decompress(out) == b
    eng.gz_set_header()
    # exact-fit and one-byte-short capacities
    want = oracle.gz_deflate(bufs[3], level=6)
    assert eng.deflate_many([bufs[3]], decompress_amd.FORMAT_GZIP, level=6, caps=[len(want)])[0][1] == want
    assert eng.deflate_many([bufs[3]], decompress_amd.FORMAT_GZIP, level=6, caps=[len(want) - 1])[0][0] == 2


def test_every_os_value(eng, oracle):
    """test/test.ml:1926-1958 (`test_gzip_os` for the 15 Gz.os constructors): 256 random bytes, level 4, header CRC —
    the frame equals the oracle's, and the contents and the OS byte come back through Gz.Higher.uncompress"""
    import random
    from decompress_amd import gz
    rng = random.Random(1926)
    for name, os_ in gz.OS.items():
        data = bytes(rng.getrandbits(8) for _ in range(256))
        z = gz.Higher.compress(data, level=4, hcrc=True, os=name)
        assert z == oracle.gz_deflate(data, level=4, hcrc=True, os=os_)
        verdict, meta, out = gz.Higher.uncompress(z, 256)
        assert (verdict, out, meta["os"]) == ("Ok", data, os_)


def test_round_trip_c4_corpus(eng, oracle):
    """BASELINE config 4 on its own data: the 15 files of the reference's test/corpus (two cycles = 30 members) as
    gzip members, mtime 0, os Unix, no name, level 4: Gz.Def on the GPU = the oracle's bytes (and python's gzip
    reads them), then Gz.Inf on the GPU restores the files"""
    import gzip
    import decompress_amd
    from decompress_amd import workloads, gz
    files = list(workloads.corpus().values())
    assert len(files) == 15 and sum(map(len, files)) == 3263944
    bufs = files * 2
    z = gz.Def.deflate_batch(bufs, level=4)
    assert all(st == 0 for st, _, _ in z)
    for b, (_, o, crc) in zip(files, z):
        assert o == oracle.gz_deflate(b, level=4) and crc == zlib.crc32(b) and gzip.decompress(o) == b
    back = gz.Inf.inflate_batch([o for _, o, _ in z], [len(b) for b in bufs])
    for b, (st, used, out, crc), (_, o, _) in zip(bufs, back, z):
        assert (st, used, out, crc) == (0, len(o), b, zlib.crc32(b))
    assert gz.Higher.uncompress(gz.Higher.compress(b"hello gz", level=4, filename=b"h"), 64)[1]["filename"] == b"h"


def test_crc32_batch(eng):
    import torch
    rng = np.random.default_rng(3)
    lens = [0, 1, 15, 16, 17, 63, 64, 65, 1000, 4097, 100001, 262144, 300007]
    bufs = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in lens]
    off, blob = [], bytearray(b"\x55" * 3)  # unaligned start on purpose
    for b in bufs:
        off.append(len(blob))
        blob += b + b"\xaa" * 5
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).to(dev)
    crc = eng.crc32_batch(d, torch.tensor(off, dtype=torch.int64, device=dev), torch.tensor(lens, dtype=torch.int64, device=dev))
    eng.synchronize()
    got = [int(x) & 0xffffffff for x in crc.cpu().tolist()]
    assert got == [zlib.crc32(b) for b in bufs]
