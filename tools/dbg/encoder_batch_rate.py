"""Run ON THE GPU BOX: n streaming encoders fed in pieces - md_def_batch_* (one launch per round of pieces) against
md_def_* (one launch per encoder and piece)."""
import ctypes, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import decompress_amd
from decompress_amd import workloads
eng = decompress_amd.Engine(0)
lib = eng.lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PIECE = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ROUNDS = 16
datas = [workloads.text(100 + i % 64, PIECE * ROUNDS) for i in range(N)]
params = eng._params(6, 4096, 0, True)
buf = ctypes.create_string_buffer(4 << 20)
for trial in range(2):
    b = lib.md_def_batch_open(eng.ctx, decompress_amd.FORMAT_ZLIB, ctypes.byref(params), N)
    outs = [bytearray() for _ in range(N)]
    t0 = time.perf_counter()
    t_enc = 0.0
    for r in range(ROUNDS + 1):
        for i in range(N):
            chunk = datas[i][r * PIECE:(r + 1) * PIECE]
            lib.md_def_batch_src(b, i, chunk, len(chunk))
        t1 = time.perf_counter()
        assert lib.md_def_batch_encode(b) == 0
        t_enc += time.perf_counter() - t1
        for i in range(N):
            while lib.md_def_batch_pending(b, i):
                k = lib.md_def_batch_out(b, i, buf, len(buf))
                outs[i] += buf.raw[:k]
    dt = time.perf_counter() - t0
    ok = all(lib.md_def_batch_status(b, i) == 2 for i in range(N)) and zlib.decompress(bytes(outs[5])) == datas[5]
    lib.md_def_batch_close(b)
print("batch: %d encoders x %d pieces of %d B: %.1f ms in all, %.1f ms in md_def_batch_encode (%.2f ms a round) = %.0f MiB/s; ok=%s"
      % (N, ROUNDS, PIECE, dt * 1e3, t_enc * 1e3, t_enc * 1e3 / (ROUNDS + 1), N * PIECE * ROUNDS / 2**20 / dt, ok))
# the single encoder, one launch per piece
eng.set_option("encoder_piece_bytes", PIECE)
M = 4
t0 = time.perf_counter()
for i in range(M):
    o = ctypes.create_string_buffer(1 << 20)
    s = lib.md_def_encoder(eng.ctx, decompress_amd.FORMAT_ZLIB, ctypes.byref(params), o, len(o))
    pos = 0
    while True:
        sig = lib.md_def_encode(s)
        if sig == 0:
            chunk = datas[i][pos:pos + PIECE]
            pos += len(chunk)
            lib.md_def_src(s, chunk, 0, len(chunk))
        elif sig == 1:
            lib.md_def_dst(s, o, len(o))
        else:
            break
    lib.md_def_free(s)
dt1 = (time.perf_counter() - t0) / M
print("single: %.1f ms per encoder (%d pieces) = %.1f MiB/s per encoder; %d of them one after the other: %.0f ms"
      % (dt1 * 1e3, ROUNDS, PIECE * ROUNDS / 2**20 / dt1, N, dt1 * N * 1e3))
