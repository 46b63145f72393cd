"""The oracle's GZip framing (oracle/gz.c, a restatement of lib/gz.ml) against the reference's
own vectors (tests/golden/gzip.json <- test/test.ml:1659-1989) and against python's gzip/zlib."""
import gzip
import io
import struct
import zlib

import pytest

from tests.conftest import load_golden


def gz_extra(payload, key):
    """Gz.Inf.extra, lib/gz.ml:617-633: 2-char key, uint16_be length, value."""
    idx = 0
    while idx + 4 <= len(payload):
        k, ln = payload[idx:idx + 2], int.from_bytes(payload[idx + 2:idx + 4], "big")
        if k == key:
            return payload[idx + 4:idx + 4 + ln]
        idx += 4 + ln
    return None


@pytest.mark.parametrize("case", load_golden("gzip.json"), ids=lambda c: c["name"])
def test_reference_vectors(oracle, case):
    src = bytes.fromhex(case["src"])
    st, used, out, meta = oracle.gz_inflate(src, 65536)
    assert st == case["status"], oracle.status_string(st)
    if "error" in case:
        assert oracle.status_string(st) == case["error"]
        return
    assert out == bytes.fromhex(case["out"]) and used == len(src)
    if "filename" in case:
        assert meta["name"] == bytes.fromhex(case["filename"])
    if "extra_key" in case:
        assert gz_extra(meta["extra"], bytes.fromhex(case["extra_key"])) == bytes.fromhex(case["extra_value"])


def test_generated_frames_round_trip(oracle):
    """test/test.ml:1760-1843 (generate empty / with name / foo): encode, decode, compare;
    python's gzip module reads the frame as well (test_with_camlzip's role, test/test.ml:1845)."""
    for data, kw in ((b"", dict(level=3)), (b"", dict(level=4, name=b"foo")), (b"foo", dict(level=4, name=b"foo")),
                     (b"foo & bar", dict(level=4, name=b"foo.gz", hcrc=True)),
                     (bytes(range(256)) * 40, dict(level=6, comment=b"a comment", hcrc=True, os=11)),
                     (b"x" * 70000, dict(level=9, name=b"n", comment=b"c")), (b"abc" * 1000, dict(level=0))):
        z = oracle.gz_deflate(data, **kw)
        st, used, out, meta = oracle.gz_inflate(z, len(data) + 16)
        assert (st, used, out) == (0, len(z), data)
        assert meta["name"] == kw.get("name") and meta["comment"] == kw.get("comment")
        assert meta["os"] == kw.get("os", 3) and meta["xfl"] == (2 if kw["level"] == 9 else 0)
        if not kw.get("hcrc"):  # the reference's CRC16 is the upper half of the CRC-32: RFC readers reject it
            assert gzip.GzipFile(fileobj=io.BytesIO(z)).read() == data
        # body and trailer by the RFC: raw DEFLATE + CRC-32 + ISIZE little-endian
        assert struct.unpack("<II", z[-8:]) == (zlib.crc32(data), len(data) & 0xffffffff)


def test_every_os_value(oracle):
    """test/test.ml:1926-1958 (`test_gzip_os`, run for the 15 Gz.os constructors, lib/gz.ml:214-246): 256 random
    bytes, level 4, header CRC; the contents and the OS come back"""
    import random
    rng = random.Random(1926)
    for os_ in list(range(14)) + [255]:
        data = bytes(rng.getrandbits(8) for _ in range(256))
        z = oracle.gz_deflate(data, level=4, hcrc=True, os=os_)
        st, used, out, meta = oracle.gz_inflate(z, 256)
        assert (st, used, out, meta["os"]) == (0, len(z), data, os_)


def test_header_layout(oracle):
    z = oracle.gz_deflate(b"hello", level=4, mtime=0x01020304, os=3, name=b"f")
    assert z[:10] == bytes([0x1f, 0x8b, 8, 8, 1, 2, 3, 4, 0, 3])  # MTIME big-endian (lib/gz.ml:801)
    assert z[10:12] == b"f\0"


def test_errors(oracle):
    good = oracle.gz_deflate(b"some text, some text, some text", level=6, name=b"n", hcrc=True)
    n = len(good)
    for cut in range(0, n):
        st, used, out, _ = oracle.gz_inflate(good[:cut], 4096)
        assert st != 0 and used == 0
    assert oracle.gz_inflate(b"\x1f\x8c" + good[2:], 4096)[0] == 10           # Invalid GZip header
    bad = bytearray(good); bad[-8] ^= 1
    assert oracle.gz_inflate(bytes(bad), 4096)[0] == 9                         # Invalid checksum
    bad = bytearray(good); bad[-4] ^= 1
    assert oracle.gz_inflate(bytes(bad), 4096)[0] == 12                        # Invalid input size
    bad = bytearray(good); bad[12] ^= 1                                        # header CRC16 (after "n\0")
    assert oracle.gz_inflate(bytes(bad), 4096)[0] == 11
    assert oracle.gz_inflate(good, 3)[0] == 2                                  # Unexpected end of output
    assert oracle.status_string(10) == "Invalid GZip header"
