"""The two placeholder windows of the long-stream path (csrc/inflate_chunked.hip) as arithmetic: a byte that
descends from window position j shows A[j] in one decode and B[j] in the other - the pair must never be equal (that is what
says "literal") and must name j.  The formulas are restated here; the GPU tests check the kernels that use them."""


def pat_a(j):
    return j & 255


def pat_b(j):
    a, h = j & 255, (j >> 8) << 1
    return h | (~a & 1) if h == (a & 0xfe) else h


def test_pairs_differ_and_name_the_position():
    seen = set()
    for j in range(32768):
        a, b = pat_a(j), pat_b(j)
        assert 0 <= a < 256 and 0 <= b < 256 and a != b
        assert a | ((b >> 1) << 8) == j
        seen.add((a, b))
    assert len(seen) == 32768


def test_launcher_keeps_a_long_stream_for_the_serial_path_on_cpu():
    """no GPU here: the library loads, exports what the long-stream path's options go through, and says so when asked for a device"""
    import ctypes
    from decompress_amd import _lib
    lib = _lib.load()
    assert hasattr(lib, "md_set_option") and hasattr(lib, "md_inflate_batch_host") and hasattr(lib, "md_de_inf_continue_host")
    assert lib.md_set_option(None, b"inflate_parallel_min", 512) < 0  # no context: a call-level error, not a crash
