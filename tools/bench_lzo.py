#!/usr/bin/env python3
"""Secondary benchmark: BASELINE config 5 — 8192 x 128 KiB buffers (half text, half printable-ASCII
noise), Lzo.compress then Lzo.uncompress on one MI355X.
    python tools/bench_lzo.py --streams 8192
Prints one JSON line (MiB/s of uncompressed bytes for each direction) + minilzo on one host core."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8192)
    ap.add_argument("--stream-kib", type=int, default=128)
    ap.add_argument("--unique", type=int, default=128)
    ap.add_argument("--kind", default="mix", help="mix (C5: half text, half ASCII noise) | text | ascii")
    args = ap.parse_args()
    import torch
    import decompress_amd
    from decompress_amd import workloads, lzo
    from tests import oracle_lib
    dev = torch.device("cuda", 0)
    eng = decompress_amd.Engine(0)
    n, nb = args.streams, args.stream_kib * 1024
    pick = lambda i: workloads.text if (args.kind == 'text' or (args.kind in ('mix', 'blocks') and i % 2 == 0)) else workloads.ascii_uniform
    uniq = [pick(i)(0xC5 + i, nb) for i in range(min(args.unique, n))]
    bufs = [uniq[i % len(uniq)] for i in range(n)]
    if args.kind == 'blocks':  # the mix's streams, all text first
        bufs = [b for i, b in enumerate(bufs) if i % 2 == 0] + [b for i, b in enumerate(bufs) if i % 2 == 1]
    blob, off, ln = workloads.pack(bufs, align=32)
    cap = np.full(n, lzo.max_compressed_length(nb), dtype=np.int64)
    zoff = np.arange(n, dtype=np.int64) * ((int(cap[0]) + 255) // 256 * 256)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_off, d_len = t(blob), t(off), t(ln)
    d_z = torch.empty(int(zoff[-1] + cap[-1]) + 64, dtype=torch.uint8, device=dev)
    d_zoff, d_zcap = t(zoff), t(cap)
    res = eng.lzo_batch(True, d_in, d_off, d_len, d_z, d_zoff, d_zcap)
    torch.cuda.synchronize()
    eng.timing_begin()
    res = eng.lzo_batch(True, d_in, d_off, d_len, d_z, d_zoff, d_zcap, results=res)
    ms_c = eng.timing_end()
    z_len, z_st = res
    ok = bool((z_st == 0).all().item())
    d_back = torch.zeros(int(blob.size) + 64, dtype=torch.uint8, device=dev)
    r = eng.lzo_batch(False, d_z, d_zoff, z_len, d_back, d_off, d_len)
    torch.cuda.synchronize()
    eng.timing_begin()
    r = eng.lzo_batch(False, d_z, d_zoff, z_len, d_back, d_off, d_len, results=r)
    ms_d = eng.timing_end()
    ok = ok and bool((r[1] == 0).all().item()) and bool((r[0] == d_len).all().item())
    ok = ok and bool(torch.equal(d_back[:blob.size], d_in[:blob.size]))
    orc = oracle_lib.load()
    zl = z_len.cpu().numpy()
    for k in range(0, min(n, len(uniq)), max(1, len(uniq) // 8)):
        got = d_z[int(zoff[k]):int(zoff[k]) + int(zl[k])].cpu().numpy().tobytes()
        ok = ok and got == orc.lzo_compress(bufs[k])[1]
    m = oracle_lib.load_minilzo()
    cpu = None
    if m is not None:
        t0 = time.perf_counter(); k = 0
        while k < len(uniq) and time.perf_counter() - t0 < 3:
            m.compress(uniq[k]); k += 1
        cpu_c = k * nb / 2**20 / (time.perf_counter() - t0)
        zs = [m.compress(u) for u in uniq[:16]]
        t0 = time.perf_counter()
        for z in zs:
            m.decompress(z, nb)
        cpu_d = len(zs) * nb / 2**20 / (time.perf_counter() - t0)
        cpu = {"compress_MiBps": round(cpu_c, 1), "uncompress_MiBps": round(cpu_d, 1), "cores": 1, "kind": "reference",
               "what": "oracle/_ref minilzo (incl. ctypes call overhead)"}
    total = float(ln.sum())
    print(json.dumps({"metric": "MiB/s LZO1X compress / uncompress over N buffers (Lzo.compress, Lzo.uncompress)",
                      "compress_MiBps": round(total / 2**20 / (ms_c * 1e-3), 1), "compress_ms": round(ms_c, 2),
                      "uncompress_MiBps": round(total / 2**20 / (ms_d * 1e-3), 1), "uncompress_ms": round(ms_d, 2),
                      "parity_ok": ok, "ratio": round(float(z_len.sum().item()) / total, 4),
                      "config": {"streams": n, "stream_bytes": nb, "unique": len(uniq)}, "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
