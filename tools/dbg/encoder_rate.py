"""md_def_* on one long stream: MiB/s of input through the encoder in pieces (one stream = one wavefront per launch)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
lib = eng.lib
for kind, level in (("text", 4), ("text", 6), ("ascii", 6)):
    data = (workloads.text(1, 1 << 20) if kind == "text" else workloads.ascii_uniform(1, 1 << 20)) * 32
    for piece_mib in (1, 8):
        eng.set_option("encoder_piece_bytes", piece_mib << 20)
        params = eng._params(level, 4096, 0, True)
        o = ctypes.create_string_buffer(1 << 20)
        s = lib.md_def_encoder(eng.ctx, decompress_amd.FORMAT_ZLIB, ctypes.byref(params), o, len(o))
        pos, n_out, t0 = 0, 0, time.perf_counter()
        while True:
            sig = lib.md_def_encode(s)
            if sig == 0:
                chunk = data[pos:pos + 65536]
                pos += len(chunk)
                lib.md_def_src(s, chunk, 0, len(chunk))
            elif sig in (1, 2):
                n_out += len(o) - lib.md_def_dst_rem(s)
                if sig == 2:
                    break
                lib.md_def_dst(s, o, len(o))
            else:
                raise SystemExit("status %d" % lib.md_def_status(s))
        dt = time.perf_counter() - t0
        lib.md_def_free(s)
        print("%s level %d, %d MiB in 64 KiB pieces, launches of %d MiB: %.1f MiB/s (ratio %.3f)" % (kind, level, len(data) >> 20, piece_mib, (len(data) >> 20) / dt, n_out / len(data)), flush=True)
