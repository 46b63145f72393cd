"""Property tests of the oracle (CPU only), in the spirit of the reference's fuzzers
(fuzz/fuzz.ml, fuzz/fuzz_ns.ml, fuzz/fuzz_lzo.ml): what the encoders emit decodes back, whatever
the parameters; the decoders never misbehave on arbitrary bytes and agree with libz on streams
libz accepts."""
import zlib

from hypothesis import given, settings, strategies as st

from tests import oracle_lib

ORC = oracle_lib.load()
data_st = st.one_of(
    st.binary(max_size=3000),
    st.builds(lambda w, n: (w * (n // max(len(w), 1) + 1))[:n], st.binary(min_size=1, max_size=12), st.integers(0, 5000)),
    st.builds(lambda n, a: bytes((i * a) & 0x3f for i in range(n)), st.integers(0, 4000), st.integers(1, 9)),
)


@settings(max_examples=150, deadline=None)
@given(data=data_st, level=st.integers(0, 9), q=st.sampled_from([16, 64, 4096]), driver=st.integers(0, 1),
       dynamic=st.booleans(), matcher=st.integers(0, 1))
def test_deflate_round_trip(data, level, q, driver, dynamic, matcher):
    z, adler = ORC.deflate_raw(data, level, q, driver, dynamic, matcher)
    assert z is not None and adler == zlib.adler32(data)
    assert zlib.decompress(z, -15) == data
    st_, used, out = ORC.de_inflate(z, len(data))
    assert (st_, used, out) == (0, len(z), data)


@settings(max_examples=150, deadline=None)
@given(data=data_st, level=st.integers(0, 9), hcrc=st.booleans(), name=st.one_of(st.none(), st.binary(min_size=0, max_size=20).filter(lambda b: 0 not in b)))
def test_gzip_round_trip(data, level, hcrc, name):
    z = ORC.gz_deflate(data, level=level, hcrc=hcrc, name=name)
    st_, used, out, meta = ORC.gz_inflate(z, len(data))
    assert (st_, used, out) == (0, len(z), data) and meta["name"] == name


@settings(max_examples=300, deadline=None)
@given(garbage=st.binary(max_size=400), cap=st.integers(0, 2000))
def test_inflate_never_misbehaves_and_agrees_with_libz(garbage, cap):
    st_, used, out = ORC.de_inflate(garbage, cap)
    assert 0 <= st_ <= 7 and len(out) <= cap and used <= len(garbage)
    d = zlib.decompressobj(-15)
    try:
        ref = d.decompress(garbage)
        ok = d.eof
    except zlib.error:
        ok = False
    if ok and len(ref) <= cap:  # a stream libz accepts completely has exactly one decoding
        assert (st_, out) == (0, ref) and used == len(garbage) - len(d.unused_data)
    if st_ == 0:
        assert ok and out == ref


@settings(max_examples=150, deadline=None)
@given(data=data_st)
def test_lzo_round_trip(data):
    st_, z = ORC.lzo_compress(data)
    assert st_ == 0 and ORC.lzo_uncompress(z, len(data)) == (0, data)


@settings(max_examples=300, deadline=None)
@given(garbage=st.binary(max_size=300), cap=st.integers(0, 3000))
def test_lzo_uncompress_never_misbehaves(garbage, cap):
    st_, out = ORC.lzo_uncompress(garbage, cap)
    assert st_ in (0, 1, 14, 15, 16) and len(out) <= cap
