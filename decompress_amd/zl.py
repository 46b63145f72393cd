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
