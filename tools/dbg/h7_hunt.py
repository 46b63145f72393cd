"""Run ON THE GPU BOX: hunt for H7's corner - what De.Lz77's window holds beyond the data when a stream ends inside a
window that earlier, shorter fills left partly unwritten (pieces smaller than the window).  Many encoders with stream
lengths around the window's slides, low-entropy data (so that the hash of the last strings finds candidates), fed in small
pieces through md_def_batch_*; every stream against the oracle handed the same pieces (its window is a real buffer)."""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import decompress_amd
from decompress_amd import workloads
from tests import oracle_lib
eng, orc = decompress_amd.Engine(0), oracle_lib.load()
lib = eng.lib
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = 160
bad = 0
buf = ctypes.create_string_buffer(1 << 20)
# round 6: the corner has a narrow door.  fill_window slides when strstart >= 65274 (lib/de.ml:4303); the matcher only gets
# there with lookahead >= 262 a step earlier, so the window is written up to 65278 at least by then - and up to 65536
# unless the pieces add up to something in between AND a long match carries strstart over the line.  What lies above
# that mark is then NOT "the byte 32 KiB earlier" one cycle later (first cycle: never written).  `first` = the first
# piece, chosen to stop the window short; lengths put the end of the stream under that stretch one cycle later.
TARGETED = len(sys.argv) > 2 and sys.argv[2] == "targeted"
for piece in ((700, 2999, 9000) if TARGETED else (700, 1000, 2999, 4096, 9000, 20000, 32768, 40000)):
    for level in (6, 1, 9):
        first = rng.choice((65280, 65300, 65400, 65500, 65535)) if TARGETED else 0
        lens = [(98304 - rng.randrange(0, 420) + rng.choice((0, 0, 32768))) if TARGETED else rng.choice((32768, 65536, 98304, 131072)) + rng.randrange(-300, 40000) for _ in range(N)]
        datas = []
        for k, n in enumerate(lens):
            kind = k % 4
            if kind == 0: d = bytes(rng.choice(b"ab") for _ in range(n))
            elif kind == 1: d = bytes(rng.getrandbits(2) for _ in range(n))
            elif kind == 2: d = (workloads.text(k, 700) * (n // 700 + 1))[:n]
            else:
                unit = bytes(rng.getrandbits(8) for _ in range(rng.randrange(3, 40)))
                d = (unit * (n // len(unit) + 1))[:n]
            datas.append(d)
        params = eng._params(level, 4096, 0, True)
        b = lib.md_def_batch_open(eng.ctx, decompress_amd.FORMAT_DEFLATE, ctypes.byref(params), N)
        outs = [bytearray() for _ in range(N)]
        pos, ended = [0] * N, [False] * N
        while not all(lib.md_def_batch_status(b, i) == 2 for i in range(N)):
            for i in range(N):
                if ended[i]: continue
                chunk = datas[i][pos[i]:pos[i] + (first if first and pos[i] == 0 else piece)]
                pos[i] += len(chunk)
                lib.md_def_batch_src(b, i, chunk, len(chunk))
                ended[i] = len(chunk) == 0
            assert lib.md_def_batch_encode(b) == 0
            for i in range(N):
                while lib.md_def_batch_pending(b, i):
                    k = lib.md_def_batch_out(b, i, buf, len(buf))
                    outs[i] += buf.raw[:k]
        lib.md_def_batch_close(b)
        for i in range(N):
            with (orc.src_pieces(first, piece) if first else orc.src_piece(piece)):
                want = orc.deflate_raw(datas[i], level, 4096)[0]
            if True:
                if bytes(outs[i]) != want:
                    bad += 1
                    print("MISMATCH piece %d level %d stream %d len %d kind %d: gpu %d bytes, oracle %d" % (piece, level, i, lens[i], i % 4, len(outs[i]), len(want)), flush=True)
        print("piece %d level %d: %d streams, %d mismatches so far" % (piece, level, N, bad), flush=True)
print("H7 HUNT", "FOUND %d" % bad if bad else "nothing found")
