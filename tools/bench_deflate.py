#!/usr/bin/env python3
"""Secondary benchmark: batched De.Lz77 + De.Def (BASELINE config 3 shape) on one MI355X.
    python tools/bench_deflate.py --streams 512 --stream-kib 256 --level 6
Prints one JSON line (MiB/s of uncompressed input) plus the oracle's single-core rate."""
import argparse, json, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--stream-kib", type=int, default=256)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--kind", default="ascii", choices=["ascii", "text", "corpus"])
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import torch
    import decompress_amd
    from decompress_amd import workloads
    from tests import oracle_lib
    dev = torch.device("cuda", 0)
    eng = decompress_amd.Engine(0)
    n, nb = args.streams, args.stream_kib * 1024
    if args.kind == "corpus":  # the reference's corpus files cycled, whole files (--stream-kib is ignored)
        files = list(workloads.corpus().values())
        bufs = [files[i % len(files)] for i in range(n)]
        nb = max(len(b) for b in bufs)
    else:
        gen = workloads.ascii_uniform if args.kind == "ascii" else workloads.text
        bufs = [gen(0xC3 + i, nb) for i in range(n)]
    blob, off, ln = workloads.pack(bufs)
    cap = np.full(n, 2 * nb + 8192, dtype=np.int64)
    ooff = np.arange(n, dtype=np.int64) * (2 * nb + 8192)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_off, d_len = t(blob), t(off), t(ln)
    d_out = torch.empty(int(cap.sum()), dtype=torch.uint8, device=dev)
    d_ooff, d_cap = t(ooff), t(cap)
    total_in = sum(len(b) for b in bufs)
    res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=args.level, total_in=total_in)
    torch.cuda.synchronize()
    eng.timing_begin()
    for _ in range(args.steps):
        res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap,
                                level=args.level, results=res, total_in=total_in)
    ms = eng.timing_end() / args.steps
    out_len, status, _ = res
    ok = bool((status == 0).all().item())
    why = [] if ok else ["status!=0 on %d streams" % int((status != 0).sum().item())]
    # size-independent property on a spread of streams: zlib round trip restores the buffer
    for k in range(0, n, max(1, n // 32)):
        got = d_out[int(ooff[k]):int(ooff[k]) + int(out_len[k].item())].cpu().numpy().tobytes()
        if zlib.decompress(got) != bufs[k]:
            ok = False
            why.append("round trip of stream %d" % k)
    orc = oracle_lib.load()
    t0 = time.perf_counter(); k = 0
    while k < n and time.perf_counter() - t0 < 5:
        z = orc.zl_deflate(bufs[k], args.level)
        got = d_out[int(ooff[k]):int(ooff[k]) + int(out_len[k].item())].cpu().numpy().tobytes()
        if got != z:
            ok = False
            why.append("bytes of stream %d differ from the oracle's" % k)
        k += 1
    cpu = sum(len(b) for b in bufs[:k]) / 2**20 / (time.perf_counter() - t0)
    print(json.dumps({"metric": "MiB/s deflate (De.Lz77 + De.Def, Zl driver) over N buffers", "value": round(total_in / 2**20 / (ms * 1e-3), 1),
                      "unit": "MiB/s", "kernel_ms": round(ms, 2), "parity_ok": ok, "parity_fail": why[:4], "ratio": round(float(out_len.sum().item()) / total_in, 4),
                      "config": {"streams": n, "stream_bytes": nb, "level": args.level, "kind": args.kind, "queue": 4096},
                      "cpu_baseline": {"value": round(cpu, 1), "unit": "MiB/s", "cores": 1, "kind": "port", "sample": "%d buffers" % k}}))


if __name__ == "__main__":
    main()
