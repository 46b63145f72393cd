import sys, os, zlib
sys.path.insert(0, "/root/repo")
import decompress_amd
eng = decompress_amd.Engine(0)
co = zlib.compressobj(9, zlib.DEFLATED, -15, 9)
z = co.compress(bytes(4 << 20)) + co.flush()
eng.set_option("profile", 1)
st, used, o, _ = eng.inflate_many([z], [4 << 20])[0]
p = eng.get_profile()
tot = sum(v for k, v in p.items() if k.startswith("cyc_"))
for k, v in p.items():
    print("%-14s %12d %s" % (k, v, "%5.1f%%" % (100.0 * v / tot) if k.startswith("cyc_") else ""))
