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
        r = Inf.Ns.inflate(src, dst_len, device)
        return r[2] if r[0] == "Ok" else r
