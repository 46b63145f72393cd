"""`decompress` — the reference's command line tool (bin/decompress.ml:226-344) over the GPU engine.

    python -m decompress_amd.cli [-d] [-f deflate|zlib|gzip|lzo] [-l <level>] [<input> [<output>]]

Same flags, defaults and drivers as the reference: inflate unless `-d`; format `deflate` unless `-f`; level 4
unless `-l`; stdin / stdout unless file names are given.  `-d` on raw DEFLATE is the CLI's own driver
(`run_deflate`, bin/decompress.ml:47-75: a Dynamic block per queue flush, then a Fixed last block that holds only
the end-of-block code — MD_DRIVER_CLI), zlib and gzip go through `Zl.Def` / `Gz.Def` (gzip: mtime = now, OS Unix,
bin/decompress.ml:136-155), LZO through `Lzo.compress` / `Lzo.uncompress_with_buffer`.  A malformed input prints
`decompress: <the reference's message>.` and exits with cmdliner's error status (124)."""
import sys
import time

FORMATS = ("deflate", "zlib", "gzip", "lzo")
CLI_ERROR = 124  # Cmdliner.Cmd.Exit.cli_error: what `Error (false, msg)` becomes (bin/decompress.ml:37, :344)
QUEUE = 4096     # bin/decompress.ml:5


def parse(argv):
    """-> (deflate, format, level, input, output); raises ValueError with the reference's messages"""
    deflate, fmt, level, pos = False, "deflate", 4, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-d":
            deflate = True
        elif a in ("-f", "--format") or a.startswith("-f") or a.startswith("--format="):
            if a in ("-f", "--format"):
                i += 1
                if i >= len(argv):
                    raise ValueError("option '-f' needs an argument")
                v = argv[i]
            else:
                v = a[len("--format="):] if a.startswith("--format=") else a[2:]
            if v.lower() not in FORMATS:
                raise ValueError('Invalid format: "%s"' % v)  # bin/decompress.ml:276
            fmt = v.lower()
        elif a in ("-l", "--level") or a.startswith("-l") or a.startswith("--level="):
            if a in ("-l", "--level"):
                i += 1
                if i >= len(argv):
                    raise ValueError("option '-l' needs an argument")
                v = argv[i]
            else:
                v = a[len("--level="):] if a.startswith("--level=") else a[2:]
            try:
                level = int(v)
            except ValueError:
                raise ValueError("Invalid level")  # bin/decompress.ml:293
            if level < 0:
                raise ValueError("The compression level must be positive")  # :292
        elif a.startswith("-") and a != "-":
            raise ValueError("unknown option '%s'" % a)
        else:
            pos.append(a)
        i += 1
    if len(pos) > 2:
        raise ValueError("too many arguments")
    return deflate, fmt, level, (pos[0] if pos else None), (pos[1] if len(pos) > 1 else None)


def run(deflate, fmt, level, data, now=None):
    """one run of the tool on `data` -> (exit status, output bytes, message for stderr)"""
    import decompress_amd
    from decompress_amd import de, engine, gz, lzo
    eng = engine.default_engine(0)
    if deflate:
        if fmt == "lzo":
            return 0, lzo.compress(data), None
        if fmt == "deflate":
            st, out, _ = eng.deflate_many([data], decompress_amd.FORMAT_DEFLATE, level=level, queue=QUEUE,
                                          driver=engine.DRIVER_CLI)[0]
        elif fmt == "zlib":
            st, out, _ = eng.deflate_many([data], decompress_amd.FORMAT_ZLIB, level=level, queue=QUEUE,
                                          driver=engine.DRIVER_ZL)[0]
        else:
            mtime = int(time.time() if now is None else now) & 0xffffffff
            st, out, _ = gz.Def.deflate_batch([data], level=level, queue=QUEUE, mtime=mtime, os=gz.OS["Unix"])[0]
        if st != 0:  # cannot happen with the room deflate_many gives a stream
            return CLI_ERROR, b"", engine.STATUS_NAMES[st]
        return 0, out, None
    if fmt == "lzo":
        cap = max(1 << 16, 8 * len(data))
        while True:  # Lzo.uncompress_with_buffer grows its buffer as it goes; here the capacity is retried
            verdict, out = lzo.uncompress(data, cap)
            if verdict == "Ok":
                return 0, out, None
            if cap >= (1 << 30) or "not large enough" not in eng.lib.md_status_string(engine.STATUS_CODES[out]).decode():
                return CLI_ERROR, b"", eng.lib.md_status_string(engine.STATUS_CODES[out]).decode() + "."
            cap *= 4
    f = {"deflate": decompress_amd.FORMAT_DEFLATE, "zlib": decompress_amd.FORMAT_ZLIB, "gzip": decompress_amd.FORMAT_GZIP}[fmt]
    verdict, out, _ = de.Inf.decode_chunks([data], o_len=1 << 16, fmt=f)  # De.io_buffer_size, the tool's `o`
    if verdict == "Ok":
        return 0, out, None
    msg = eng.lib.md_status_string(engine.STATUS_CODES[verdict]).decode()
    return CLI_ERROR, out, msg + "."  # `Malformed err -> `Error (false, str "%s." err)`; what was flushed stays written


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        deflate, fmt, level, fin, fout = parse(argv)
    except ValueError as e:
        sys.stderr.write("decompress: %s\n" % e)
        return CLI_ERROR
    data = open(fin, "rb").read() if fin else sys.stdin.buffer.read()
    status, out, msg = run(deflate, fmt, level, data)
    oc = open(fout, "wb") if fout else sys.stdout.buffer
    oc.write(out)
    oc.flush()
    if fout:
        oc.close()
    if msg:
        sys.stderr.write("decompress: %s\n" % msg)
    return status


if __name__ == "__main__":
    sys.exit(main())
