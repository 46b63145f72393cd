"""Mirror of the reference's `De` module surface for the hot path
(lib/de.mli:146-173 `De.Inf.Ns`), executed on the GPU through the C ABI."""
from . import engine as _engine


class Inf:
    class Ns:
        """De.Inf.Ns — whole-buffer inflate (lib/de.ml:1534-1823)."""

        @staticmethod
        def inflate(src, dst_len, device=0):
            """`De.Inf.Ns.inflate src dst` -> ("Ok", (consumed, written), bytes) or ("Error", variant).

            `dst_len` is the capacity of the destination bigstring."""
            st, used, out, _ = _engine.default_engine(device).inflate_many(
                [src], [dst_len], _engine.FORMAT_DEFLATE)[0]
            if st == 0:
                return "Ok", (used, len(out)), out
            return "Error", _engine.STATUS_NAMES[st]

        @staticmethod
        def inflate_batch(srcs, dst_lens, device=0):
            return _engine.default_engine(device).inflate_many(srcs, dst_lens, _engine.FORMAT_DEFLATE)


class Higher:
    """De.Higher (lib/de.ml:4517-4612): refill/flush callbacks become whole buffers."""

    @staticmethod
    def compress(src, queue=4096, device=0):
        """`De.Higher.compress ~w ~q ~refill ~flush i o` / `to_string`: raw DEFLATE at level 4
        (the reference's default: De.Higher has no ?level, lib/de.ml:4519)."""
        st, out, _ = _engine.default_engine(device).deflate_many(
            [src], _engine.FORMAT_DEFLATE, level=4, queue=queue, driver=_engine.DRIVER_HIGHER)[0]
        if st != 0:
            raise _engine.Error(_engine.STATUS_NAMES[st])
        return out

    @staticmethod
    def uncompress(src, dst_len, device=0):
        """`De.Higher.uncompress`: Ok bytes | Error (`Msg string)."""
        r = Inf.Ns.inflate(src, dst_len, device)
        return r[2] if r[0] == "Ok" else r
