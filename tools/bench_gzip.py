#!/usr/bin/env python3
"""Secondary benchmark: BASELINE config 4's per-GPU share — 4096 gzip members, member i = file[i mod 15] of the
reference's test/corpus (the data fixture tests/golden/corpus.tar.xz; 21 504 ... 768 771 B, 3 263 944 B per
cycle), mtime 0, os Unix, no name, level 4: Gz.Def then Gz.Inf on one MI355X.
    python tools/bench_gzip.py --streams 4096
Prints one JSON line (MiB/s of uncompressed bytes for each direction)."""
import argparse, json, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--level", type=int, default=4)
    args = ap.parse_args()
    import torch
    import decompress_amd
    from decompress_amd import workloads
    dev = torch.device("cuda", 0)
    eng = decompress_amd.Engine(0)
    n = args.streams
    uniq = list(workloads.corpus().values())
    bufs = [uniq[i % len(uniq)] for i in range(n)]
    blob, off, ln = workloads.pack(bufs)
    cap = (ln + 8192).astype(np.int64)
    ooff = np.zeros(n, dtype=np.int64)
    np.cumsum(((cap + 255) // 256 * 256)[:-1], out=ooff[1:])
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_off, d_len = t(blob), t(off), t(ln)
    d_z = torch.empty(int(ooff[-1] + cap[-1]), dtype=torch.uint8, device=dev)
    d_zoff, d_zcap = t(ooff), t(cap)
    eng.gz_set_header(mtime=0, os=3)
    res = eng.deflate_batch(decompress_amd.FORMAT_GZIP, d_in, d_off, d_len, d_z, d_zoff, d_zcap, level=args.level)
    torch.cuda.synchronize()
    eng.timing_begin()
    res = eng.deflate_batch(decompress_amd.FORMAT_GZIP, d_in, d_off, d_len, d_z, d_zoff, d_zcap, level=args.level, results=res)
    ms_def = eng.timing_end()
    z_len, z_status, _ = res
    ok = bool((z_status == 0).all().item())
    d_back = torch.zeros(int(blob.size) + 64, dtype=torch.uint8, device=dev)
    r = eng.inflate_batch(decompress_amd.FORMAT_GZIP, d_z, d_zoff, z_len.to(torch.int64), d_back, d_off, d_len)
    torch.cuda.synchronize()
    eng.timing_begin()
    r = eng.inflate_batch(decompress_amd.FORMAT_GZIP, d_z, d_zoff, z_len.to(torch.int64), d_back, d_off, d_len, results=r)
    ms_inf = eng.timing_end()
    out_len, consumed, status, crc = r
    ok = ok and bool((status == 0).all().item()) and bool((out_len == d_len).all().item())
    ok = ok and bool(torch.equal(d_back[:blob.size], d_in[:blob.size]))
    crc = crc.cpu().numpy().view(np.uint32)
    ok = ok and all(int(crc[i]) == zlib.crc32(bufs[i]) for i in range(0, n, max(1, n // 64)))
    total = float(ln.sum())
    print(json.dumps({"metric": "MiB/s gzip deflate / inflate over N members (Gz.Def level %d, Gz.Inf)" % args.level,
                      "deflate_MiBps": round(total / 2**20 / (ms_def * 1e-3), 1), "deflate_ms": round(ms_def, 2),
                      "inflate_MiBps": round(total / 2**20 / (ms_inf * 1e-3), 1), "inflate_ms": round(ms_inf, 2),
                      "round_trip_ok": ok, "ratio": round(float(z_len.sum().item()) / total, 4),
                      "config": {"members": n, "bytes": int(total), "level": args.level, "unique": len(uniq)}}))


if __name__ == "__main__":
    main()
