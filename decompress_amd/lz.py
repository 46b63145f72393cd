"""Mirror of the reference's `Lz` module (decompress.lz, lib/lz.ml / lib/lz.mli): the variant match
finder `Lz.state ?level ~q ~w src` / `Lz.compress`, run on the GPU under the same drivers and
encoder as `De.Lz77`."""
from . import engine as _engine


def compress(src, level=4, queue=4096, fmt=_engine.FORMAT_DEFLATE, driver=_engine.DRIVER_ZL, dynamic=True, device=0):
    """`Lz.state ?level` (default 4, lib/lz.ml:525) feeding `De.Def` under `driver`; the driver pushes
    the end-of-block command at `End (`Lz.trailing` does not, lib/lz.ml:348-354)."""
    eng = _engine.default_engine(device)
    st, out, _ = eng.deflate_many([src], fmt, level=level, queue=queue, driver=driver, dynamic=dynamic,
                                  matcher=_engine.MATCHER_LZ)[0]
    if st != 0:
        raise _engine.Error(_engine.STATUS_NAMES[st])
    return out
