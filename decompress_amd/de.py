"""Mirror of the reference's `De` module surface for the hot path (lib/de.mli: `De.Inf`, `De.Inf.Ns`, `De.Lz77`,
`De.Def`, `De.Higher`), executed on the GPU through the C ABI."""
import ctypes

from . import engine as _engine

AWAIT, FLUSH, END, MALFORMED = 0, 1, 2, 3
EOB = 256


def copy_cmd(off, length):
    """De.Queue.cmd (`Copy (off, len)), lib/de.ml:2245-2266"""
    return 0x2000000 | ((length - 3) << 16) | (off - 1)


class Lz77:
    """De.Lz77 (lib/de.mli:453-524)"""

    @staticmethod
    def compress(src, level=4, queue=4096, matcher=_engine.MATCHER_DE, device=0):
        """`De.Lz77.state ?level ~q ~w src` run to `End: (commands of every queue fill, literals[286], distances[30])"""
        eng = _engine.default_engine(device)
        src = bytes(src)
        cap = len(src) + len(src) // max(1, queue - 1) + 8
        cmds = (ctypes.c_uint32 * cap)()
        lits, dsts, n = (ctypes.c_uint32 * 286)(), (ctypes.c_uint32 * 30)(), ctypes.c_size_t()
        st = eng.lib.md_de_lz77_compress(eng.ctx, level, queue, matcher, src, len(src), cmds, cap, ctypes.byref(n), lits, dsts)
        if st < 0:
            eng._check(st)
        if st != 0:
            raise _engine.Error(_engine.STATUS_NAMES[st])
        return list(cmds[:n.value]), list(lits), list(dsts)


class Def:
    """De.Def (lib/de.mli:300-412)"""
    FLAT, FIXED, DYNAMIC = 0, 1, 2

    @staticmethod
    def encode(cmds, kind, device=0):
        """the commands as ONE last block of `kind` (`Def.encode e (`Block {kind; last = true})` then `Flush);
        Dynamic = dynamic_of_frequencies of the commands' own histogram (test/test.ml:84-95)"""
        eng = _engine.default_engine(device)
        arr = (ctypes.c_uint32 * max(1, len(cmds)))(*cmds)
        cap = 8 * len(cmds) + 1024
        dst, n = ctypes.create_string_buffer(cap), ctypes.c_size_t()
        st = eng.lib.md_de_def_encode(eng.ctx, kind, arr, len(cmds), dst, cap, ctypes.byref(n))
        if st < 0:
            eng._check(st)
        if st != 0:
            raise _engine.Error(_engine.STATUS_NAMES[st])
        return dst.raw[:n.value]


    # operations of Def.run (MD_OP_* of include/mdeflate.h)
    OP_FILL, OP_BLOCK, OP_FLUSH, OP_SUCC_LITERAL, OP_SUCC_LENGTH, OP_SUCC_DISTANCE, OP_NEW_FREQS, OP_QUEUE_RESET = range(1, 9)

    @staticmethod
    def run(ops, queue=4096, device=0):
        """`Def.encode` driven step by step (md_de_def_run): ops is the list of operations the reference's tests
        perform on an encoder with a `Buffer destination — fill the queue, count frequencies, `Block {kind; last},
        `Flush (test/test_ns.ml:388-615).  -> (bytes written, [0 for `Ok | 1 for `Block, ...])"""
        eng = _engine.default_engine(device)
        arr = (ctypes.c_uint32 * max(1, len(ops)))(*ops)
        cap = 8 * len(ops) + 1024
        dst, n = ctypes.create_string_buffer(cap), ctypes.c_size_t()
        res, nres = (ctypes.c_uint8 * 315)(), ctypes.c_size_t()
        st = eng.lib.md_de_def_run(eng.ctx, queue, arr, len(ops), dst, cap, ctypes.byref(n), res, 315, ctypes.byref(nres))
        if st < 0:
            eng._check(st)
        if st != 0:
            raise _engine.Error(_engine.STATUS_NAMES[st])
        return dst.raw[:n.value], list(res[:nres.value])


    class Ns:
        """De.Def.Ns (lib/de.ml:3040-4010): the whole-buffer compressor"""

        @staticmethod
        def compress_bound(n, device=0):
            return _engine.default_engine(device).lib.md_de_def_ns_compress_bound(n)

        @staticmethod
        def deflate(src, level=4, dst_len=None, device=0, _zl=False):
            """`De.Def.Ns.deflate ?level src dst` -> ("Ok", bytes written) | ("Error", variant); dst_len = the capacity of
            dst (default compress_bound).  Levels 5..12 are stubs upstream: ("Ok", b"")."""
            eng = _engine.default_engine(device)
            src = bytes(src)
            lib = eng.lib
            if dst_len is None:
                dst_len = (lib.md_zl_def_ns_compress_bound if _zl else lib.md_de_def_ns_compress_bound)(len(src))
            dst, n = ctypes.create_string_buffer(max(1, dst_len)), ctypes.c_size_t()
            fn = lib.md_zl_def_ns_deflate if _zl else lib.md_de_def_ns_deflate
            st = fn(eng.ctx, level, src, len(src), dst, dst_len, ctypes.byref(n))
            if st == -1 and not 0 <= level <= 12:
                return "Error", "Invalid_compression_level"
            if st < 0:
                eng._check(st)
            return ("Ok", dst.raw[:n.value]) if st == 0 else ("Error", _engine.STATUS_NAMES[st])


class Inf:
    last_message = ""

    @staticmethod
    def decode_chunks(chunks, o_len=65536, fmt=_engine.FORMAT_DEFLATE, device=0, chunk_bytes=None):
        """the streaming protocol of lib/de.mli:82-144 — decoder / src / decode / flush / dst_rem — driven the way
        De.Higher.uncompress drives it: -> ("Ok" | "Malformed", bytes, [signals]).  chunk_bytes: how much input the decoder
        buffers before it decodes a piece (md_inf_chunk_bytes; default 8 MiB)"""
        eng = _engine.default_engine(device)
        lib = eng.lib
        o = ctypes.create_string_buffer(o_len)
        d = lib.md_inf_decoder(eng.ctx, fmt, o, o_len)
        if chunk_bytes is not None:
            lib.md_inf_chunk_bytes(d, chunk_bytes)
        out, sigs, it = bytearray(), [], iter(chunks)
        try:
            while True:
                sig = lib.md_inf_decode(d)
                sigs.append(sig)
                if sig == AWAIT:
                    c = next(it, b"")
                    lib.md_inf_src(d, bytes(c), 0, len(c))
                elif sig == FLUSH:
                    out += o.raw[:o_len - lib.md_inf_dst_rem(d)]
                    lib.md_inf_flush(d)
                else:
                    out += o.raw[:o_len - lib.md_inf_dst_rem(d)]
                    st = lib.md_inf_status(d)
                    Inf.last_message = lib.md_inf_message(d).decode()  # the reference's `Malformed string, numbers included
                    if st < 0:
                        eng._check(st)
                    return ("Ok" if sig == END else _engine.STATUS_NAMES[st]), bytes(out), sigs
        finally:
            lib.md_inf_free(d)

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
        """`De.Higher.uncompress ~w ~refill ~flush i o` (lib/de.ml:4555-4571): bytes | ("Error", msg)."""
        eng = _engine.default_engine(device)
        src = bytes(src)
        dst, n = ctypes.create_string_buffer(max(1, dst_len)), ctypes.c_size_t()
        st = eng.lib.md_de_higher_uncompress(eng.ctx, src, len(src), dst, dst_len, ctypes.byref(n))
        if st < 0:
            eng._check(st)
        return dst.raw[:n.value] if st == 0 else ("Error", eng.lib.md_status_string(st).decode())
