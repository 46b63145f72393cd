"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/mdeflate.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from decompress_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_loads():
    build.build()
    lib = _lib.load()
    assert lib.md_version() == 0x000300


def test_header_symbols_exported():
    build.build()
    hdr = open(os.path.join(ROOT, "include", "mdeflate.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 12
    lib = ctypes.CDLL(_lib.SO)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound


def test_status_strings_match_reference():
    # lib/de.ml:1557-1567, lib/zl.ml:183
    lib = _lib.load()
    want = {0: "Ok", 1: "Unexpected end of input", 2: "Unexpected end of output",
            3: "Invalid kind of block", 4: "Invalid dictionary",
            5: "Invalid complement of length", 6: "Invalid distance",
            7: "Invalid distance code", 8: "Invalid Zlib header", 9: "Invalid checksum"}
    for k, v in want.items():
        assert lib.md_status_string(k).decode() == v


def test_no_device_is_loud():
    """Without a GPU the product path must fail, not fall back."""
    lib = _lib.load()
    if lib.md_device_count() == 0:
        assert not lib.md_create(0, None)
        assert lib.md_last_error_string(None)
        import pytest
        import decompress_amd
        with pytest.raises(decompress_amd.Error):
            decompress_amd.Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "decompress_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "orc_" not in txt, f


def test_shard_plan_equals_shard_by_bytes():
    """md_shard_plan (C ABI, for non-Python callers) cuts a batch exactly like shard.shard_by_bytes."""
    import random
    from decompress_amd import shard
    lib = _lib.load()
    rng = random.Random(7)
    cases = [[], [5], [1000] * 4096, [21504, 768771, 111261, 377109] * 50]
    cases += [[rng.randrange(0, 1 << rng.randrange(1, 22)) for _ in range(rng.randrange(1, 300))] for _ in range(40)]
    for lengths in cases:
        for world in (1, 2, 3, 8):
            n = len(lengths)
            arr = (ctypes.c_uint64 * max(n, 1))(*lengths)
            lo = (ctypes.c_uint64 * world)()
            hi = (ctypes.c_uint64 * world)()
            assert lib.md_shard_plan(n, arr, world, lo, hi) == 0
            assert [(int(a), int(b)) for a, b in zip(lo, hi)] == shard.shard_by_bytes(lengths, world)
    assert lib.md_shard_plan(3, None, 2, None, None) < 0 and lib.md_shard_plan(0, None, 0, None, None) < 0
