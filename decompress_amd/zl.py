"""Mirror of the reference's `Zl` module surface for the hot path
(lib/zl.ml:382-418 `Zl.Inf.Ns`), executed on the GPU through the C ABI."""
from . import engine as _engine


class Inf:
    class Ns:
        """Zl.Inf.Ns — zlib-framed whole-buffer inflate (lib/zl.ml:400-417)."""

        @staticmethod
        def inflate(src, dst_len, device=0):
            st, used, out, _ = _engine.default_engine(device).inflate_many(
                [src], [dst_len], _engine.FORMAT_ZLIB)[0]
            if st == 0:
                return "Ok", (used, len(out)), out
            return "Error", _engine.STATUS_NAMES[st]

        @staticmethod
        def inflate_batch(srcs, dst_lens, device=0):
            return _engine.default_engine(device).inflate_many(srcs, dst_lens, _engine.FORMAT_ZLIB)


class Def:
    class Ns:
        """Zl.Def.Ns (lib/zl.ml:596-629): De.Def.Ns inside a zlib frame"""

        @staticmethod
        def compress_bound(n, device=0):
            return _engine.default_engine(device).lib.md_zl_def_ns_compress_bound(n)

        @staticmethod
        def deflate(src, level=4, dst_len=None, device=0):
            from . import de
            return de.Def.Ns.deflate(src, level, dst_len, device, _zl=True)


class Higher:
    """Zl.Higher (lib/zl.ml:633-667)."""

    @staticmethod
    def compress(src, level=6, dynamic=True, queue=4096, device=0):
        """`Zl.Higher.compress ?level ?dynamic ~w ~q ~refill ~flush i o`: a zlib stream."""
        st, out, _ = _engine.default_engine(device).deflate_many(
            [src], _engine.FORMAT_ZLIB, level=level, queue=queue, driver=_engine.DRIVER_ZL, dynamic=dynamic)[0]
        if st != 0:
            raise _engine.Error(_engine.STATUS_NAMES[st])
        return out

    @staticmethod
    def uncompress(src, dst_len, device=0):
        """`Zl.Higher.uncompress ~allocate ~refill ~flush i o` (lib/zl.ml:650-666): bytes | ("Error", msg)."""
        import ctypes
        eng = _engine.default_engine(device)
        src = bytes(src)
        dst, n = ctypes.create_string_buffer(max(1, dst_len)), ctypes.c_size_t()
        st = eng.lib.md_zl_higher_uncompress(eng.ctx, src, len(src), dst, dst_len, ctypes.byref(n))
        if st < 0:
            eng._check(st)
        return dst.raw[:n.value] if st == 0 else ("Error", eng.lib.md_status_string(st).decode())
