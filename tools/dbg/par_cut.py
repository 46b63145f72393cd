import sys, zlib
sys.path.insert(0, "/root/repo")
import decompress_amd
from decompress_amd import de, engine, workloads
eng = engine.default_engine(0)
def last():
    v = eng.lib.md_set_option(eng.ctx, b"inflate_parallel_last", 0)
    return v & 0xffffff, v >> 24
plain = workloads.text(0x52, 24 << 20)
z = zlib.compress(plain, 6)
for name, src in (("whole", z), ("cut", z[:len(z) // 2]), ("cut3", z[:len(z) // 3]), ("bad", z[:-1] + bytes([z[-1] ^ 1]))):
    r = de.Inf.decode_chunks([src], o_len=65536, fmt=decompress_amd.FORMAT_ZLIB)
    print(name, r[0], len(r[1]), last(), r[1] == plain[:len(r[1])])
