#!/usr/bin/env python3
"""bench.py — headline benchmark: batched zlib inflate on MI355X.

A "step" is one pass of the hot path (Zl.Inf.Ns semantics, one stream per
wavefront) over one batch of BASELINE.json config[1]:
    4096 x 256 KiB zlib streams, dynamic Huffman (libz level 6), per GPU.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank
inflates its own 4096 streams; value = total uncompressed MiB / s over all GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is for the hot path's kernels taken together
(md::v4::decode_kernel -> token log in HBM -> md::v4::resolve_kernel, for the two halves
of the batch on two forked streams; one "launch" = those four kernels, which overlap):
algorithmic bytes = compressed bytes read + uncompressed bytes written per launch, over
the launch duration measured with HIP events on the context's stream, which forks and
joins the side stream; peak = 8 TB/s HBM3E (MI355X_MICROARCH.md).  `traffic` =
FETCH_SIZE + WRITE_SIZE of those kernels per launch from the committed rocprofv3 --pmc
passes of this same command (profiles/r01_final/pmc_summary.json; raw counter bytes,
see the note there), null when the configuration differs from the profiled one.
`cpu_baseline` is the repo's C restatement of lib/de.ml (oracle/, kind "port")
timed single-threaded on a bounded sample of the same streams.
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(args, n, nbytes):
    """HBM bytes per launch from the committed PMC passes (collected by tools/profile_gpu.sh with
    this workload and the default kernels); null for any other configuration."""
    if (args.kernel or 3) != 3 or max(args.variant, 0) != 0 or n != 4096 or nbytes != 262144 or args.overlap not in (0, 2):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_final", "pmc_summary.json")) as f:
            return int(json.load(f)["traffic_bytes_per_launch_raw"])
    except (OSError, KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--stream-kib", type=int, default=256)
    ap.add_argument("--unique", type=int, default=0, help="distinct streams to generate (0 = all)")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--ring-log2", type=int, default=0)
    ap.add_argument("--kernel", type=int, default=0,
                    help="1 = serial per wave, 2 = lane-parallel fused, 3 = lane-parallel split (default)")
    ap.add_argument("--variant", type=int, default=-1, help="v2 geometry")
    ap.add_argument("--overlap", type=int, default=0, help="parts of the batch run on forked streams (1 = none, default 2)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--profile", action="store_true", help="print the in-kernel phase profile of stream 0 (stderr)")
    return ap.parse_args()


def cpu_baseline(streams, nbytes, budget_s):
    """Oracle (C port of De.Inf.Ns / Zl.Inf.Ns) on the host cores of this box, bounded sample:
    one thread first, then one stream per thread on all cores (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import oracle_lib
    orc = oracle_lib.load()

    def one(z):
        rc, used, out = orc.zl_inflate(z, nbytes)
        assert rc == 0 and used == len(z) and len(out) == nbytes
        return len(out)

    done = 0
    t0 = time.perf_counter()
    k = 0
    while k < len(streams):
        done += one(streams[k])
        k += 1
        if time.perf_counter() - t0 > budget_s * 0.4:
            break
    dt = time.perf_counter() - t0
    single = done / 2**20 / dt
    # anchor: libz on the same sample
    t1 = time.perf_counter()
    for z in streams[:k]:
        zlib.decompress(z)
    dz = time.perf_counter() - t1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = ""
    try:  # a container CPU quota below the visible CPUs is what the host really gives
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = "; cgroup cpu.max = %.1f of %d visible CPUs" % (float(q) / float(per), cores)
            cores = max(1, min(cores, int(-(-float(q) // float(per)))))
    except (OSError, ValueError):
        pass
    # all cores: every thread inflates its share of the batch inside ONE C call (GIL released)
    import ctypes
    blob = b"".join(streams)
    offs = np.zeros(len(streams), dtype=np.uint64)
    lens = np.array([len(z) for z in streams], dtype=np.uint64)
    np.cumsum(lens[:-1], out=offs[1:])
    fn = orc.lib.orc_zl_inf_ns_inflate_batch
    fn.restype = ctypes.c_size_t
    fn.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                   ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    shares = np.array_split(np.arange(len(streams)), cores)

    def share(idx):
        if len(idx) == 0:
            return 0
        scratch = ctypes.create_string_buffer(nbytes)
        tot = ctypes.c_uint64()
        o, l = np.ascontiguousarray(offs[idx]), np.ascontiguousarray(lens[idx])
        bad = fn(blob, o.ctypes.data, l.ctypes.data, len(idx), scratch, nbytes, ctypes.byref(tot))
        assert bad == 0
        return tot.value

    total, reps = 0, 0
    t2 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        while time.perf_counter() - t2 < budget_s * 0.4:
            total += sum(ex.map(share, shares))
            reps += 1
    da = time.perf_counter() - t2
    return {
        "value": round(total / 2**20 / da, 2), "unit": "MiB/s", "cores": cores, "kind": "port",
        "single_core_value": round(single, 2),
        "sample": "oracle/de_inflate.c Zl.Inf.Ns: %d x the batch's %d streams on %d threads (one stream per thread); "
                  "1 thread: %d streams (%d MiB out) at %.1f MiB/s; libz 1.2.11 inflate on that sample, 1 thread: "
                  "%.1f MiB/s%s" % (reps, len(streams), cores, k, done >> 20, single, done / 2**20 / dz, quota),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import decompress_amd
    from decompress_amd import workloads

    eng = decompress_amd.Engine(local_rank)
    if args.ring_log2:
        eng.set_option("ring_log2", args.ring_log2)
    if args.kernel:
        eng.set_option("kernel", args.kernel)
    if args.variant >= 0:
        eng.set_option("variant", args.variant)
    if args.overlap:
        eng.set_option("overlap", args.overlap)
    if os.environ.get("MD_SPLIT_PCT"):
        eng.set_option("split_pct", int(os.environ["MD_SPLIT_PCT"]))

    n = args.streams
    nbytes = args.stream_kib * 1024
    unique = args.unique or n
    # every rank owns different streams (seed offset by rank): weak scaling
    t_gen = time.perf_counter()
    streams = workloads.c2_streams(n, nbytes=nbytes, level=args.level, seed0=0xC2 + rank * n,
                                   unique=unique, workers=max(1, (os.cpu_count() or 8) // max(1, world)))
    t_gen = time.perf_counter() - t_gen
    assert all(((s[2] >> 1) & 3) == 2 for s in streams[:64]), "first block must be dynamic Huffman"
    blob, in_off, in_len = workloads.pack(streams, align=16)
    comp_bytes = int(in_len.sum())
    out_off = np.arange(n, dtype=np.int64) * nbytes
    out_cap = np.full(n, nbytes, dtype=np.int64)

    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_in_off, d_in_len = t(blob), t(in_off), t(in_len)
    d_out = torch.empty(n * nbytes, dtype=torch.uint8, device=dev)
    d_out_off, d_out_cap = t(out_off), t(out_cap)
    results = None

    def step():
        nonlocal results
        results = eng.inflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_in_off, d_in_len, d_out,
                                    d_out_off, d_out_cap, results)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    if args.profile and rank == 0:
        eng.set_option("profile", 1)
        step()
        prof = eng.get_profile()
        eng.set_option("profile", 0)
        cyc = sum(v for k, v in prof.items() if k.startswith("cyc_"))
        print("profile(stream 0): total %.2f Mcycles" % (cyc / 1e6), file=sys.stderr)
        for k, v in prof.items():
            print("  %-12s %12d %s" % (k, v, ("%5.1f%%" % (100.0 * v / cyc)) if k.startswith("cyc_") else ""), file=sys.stderr)
    eng.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = eng.timing_end() / max(1, args.steps)  # HIP events on the kernel's stream
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- verification (outside the timed region): every stream Ok, bytes exact
    out_len, consumed, status, checksum = results
    ok = bool((status == 0).all().item()) and bool((out_len == nbytes).all().item())
    ok = ok and bool((consumed == d_in_len).all().item())
    if not args.no_verify:
        host_sum = checksum.cpu().numpy().view(np.uint32)
        idx = list(range(0, n, max(1, n // 32)))
        for i in idx:
            plain = zlib.decompress(streams[i])
            got = d_out[i * nbytes:(i + 1) * nbytes].cpu().numpy().tobytes()
            ok = ok and got == plain and int(host_sum[i]) == zlib.adler32(plain)
    digest = int(checksum.to(torch.int64).bitwise_and(0xffffffff).sum().item())
    if world > 1:
        # the path's only exchange: gather per-rank result digests (RCCL all_gather)
        from decompress_amd import shard
        g = torch.tensor([digest, int(ok)], dtype=torch.int64, device=dev)
        gl = shard.gather_results(dist, g, world)
        ok = all(int(x[1].item()) for x in gl)
        digest = sum(int(x[0].item()) for x in gl) & 0xffffffffffff
    if not ok:
        print("bench: PARITY FAILURE (status/bytes/checksum mismatch)", file=sys.stderr)

    if rank == 0:
        total_out = world * n * nbytes * args.steps
        value = total_out / 2**20 / elapsed
        algo_bytes = comp_bytes + n * nbytes  # per launch: C read once + U written once
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "MiB/s inflate over N zlib streams (uncompressed bytes / wall second)",
            "value": round(value, 1),
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "parity_ok": ok,
            "config": {
                "workload": "C2: %d x %d KiB zlib streams per GPU, dynamic Huffman (libz level %d), "
                            "Zl.Inf.Ns semantics, one stream per wavefront" % (n, args.stream_kib, args.level),
                "streams_per_gpu": n, "stream_bytes": nbytes, "unique_streams": unique,
                "compressed_ratio": round(comp_bytes / (n * nbytes), 4),
                "kernel": args.kernel or 3, "variant": max(args.variant, 0), "gen_seconds": round(t_gen, 1),
                "result_digest": digest,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(args, n, nbytes),
                "kernel": {1: "inflate_kernel", 2: "inflate_v4_kernel", 3: "2 x (decode_kernel + resolve_kernel), overlapped", 5: "inflate_wave_kernel"}[args.kernel or 3],
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(streams, nbytes, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
