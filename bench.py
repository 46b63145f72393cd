#!/usr/bin/env python3
"""bench.py — headline benchmark: batched zlib inflate (and the deflate leg) on MI355X.

A "step" is one pass of the hot path (Zl.Inf.Ns semantics, one stream per workgroup of two wavefronts) over one
batch of BASELINE.json config[1]:
    4096 x 256 KiB zlib streams, dynamic Huffman (libz level 6), per GPU.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank inflates its own
4096 streams; value = total uncompressed MiB / s over all GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus 8            # re-executes itself under torch.distributed.run, one rank per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).
  roofline      the hot path's one kernel, md::wv::inflate_wave_kernel: algorithmic bytes = compressed bytes
                read + uncompressed bytes written per launch, over the launch duration
                measured with HIP events on the stream the kernel runs on; peak = 8 TB/s HBM3E
                (MI355X_MICROARCH.md).  `traffic` = FETCH_SIZE + WRITE_SIZE of that kernel per launch from the
                committed rocprofv3 --pmc passes of this same command (PMC_DIR below: profiles/<round>_final/pmc_summary.json,
                gfx950 correction of the guide applied; `traffic_source` names the file and the commit it was
                collected at), null when the configuration differs.
  cpu_baseline  the repo's C restatement of lib/de.ml (oracle/, kind "port") on the host cores of this box,
                bounded sample of the same streams.
  deflate       BASELINE.json config[2] on the same GPU(s), outside the inflate timed region: 4096 x 1 MiB
                printable-ASCII buffers, De.Lz77 + De.Def level 6, queue 4096, Zl driver — its own value,
                ms_per_step, roofline and cpu_baseline (BASELINE's metric is "inflate+deflate").  One step = the
                three kernels of the path (hash chains, longest_match ahead, parse + encode).
  r01_workload  the same kernel on round 1's input (every stream seeded word text), N = 1 only.
  gzip          BASELINE.json config[3], STRONG scaling (outside `value`): one batch of 32768 gzip members = the
                reference's corpus files cycled, sharded over the N ranks by bytes; Gz.Def level 4 then Gz.Inf on
                every shard; the compressed members are then gathered to rank 0 (sizes by all_gather, bytes by
                exact-size send/recv) — `results_gathered`, `gather_ms`.
  lzo           BASELINE.json config[4] (N = 1 only): 8192 x 128 KiB buffers through Lzo.compress then
                Lzo.uncompress — MiB/s, ms and HBM fraction per direction, round trip and oracle bytes checked.
  long_stream   ONE zlib stream of 64 MiB through Zl.Higher.uncompress (N = 1 only): decoded in pieces by the whole device
                (DESIGN 3b), host to host, beside the one-pair-of-wavefronts path the call took before round 6.
  def_ns        SURVEY 8(f) row 4 (N = 1 only): De.Def.Ns.deflate level 4 over 1024 x 256 KiB buffers, inflated back,
                oracle bytes checked on a sample.
  ranks_seen    an all_reduce over the process group: how many ranks really took part.
For N > 1 the per-stream results of every rank (sizes and Adler-32 from the kernel) are gathered with
decompress_amd.shard.gather_varlen — the path's only exchange (RCCL over xGMI).
"""
import argparse
import json
import os
import socket
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
PMC_DIR = os.path.join(ROOT, "profiles", "r06_final")


def _csrc_digest():
    """Digest of everything under csrc/ (kernels and the host glue that decides the launch pattern): a committed counter
    profile describes this tree only as long as it is unchanged."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "decompress_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp", ".h")):  # kernels AND host glue: slices, grids and the workspace cap live in capi.cpp
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _pmc_summary():
    """The committed rocprofv3 --pmc summary, or None when it is missing or STALE: collected from other kernel sources
    than the ones in this tree (tools/profile_gpu.sh stamps it with _csrc_digest())."""
    try:
        with open(os.path.join(PMC_DIR, "pmc_summary.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    return d if d.get("csrc_digest") == _csrc_digest() else None


def pmc_traffic(names, is_default, calls_key=None):
    """HBM bytes per launch from the COMMITTED rocprofv3 --pmc passes of this same command (tools/profile_gpu.sh on the
    default workload, profiles/<round>/pmc_summary.json) — not measured by this run: counters and timing cannot be
    collected in one process.  null when the workload differs or when a kernel source has changed since the profile
    was taken.  `names`: the kernels whose traffic adds up (the deflate path is three kernels)."""
    d = _pmc_summary() if is_default else None
    if d is None:
        return None
    try:
        if calls_key is None:
            return int(sum(d[k]["traffic_bytes_per_launch"] for k in names))
        # a call may be several launches of a kernel (a deflate batch in slices): the dispatches' mean x launches per call
        return int(sum(d[k]["traffic_bytes_per_launch"] * d[k]["launches"] / d[calls_key] for k in names))
    except (KeyError, ValueError, TypeError, ZeroDivisionError):
        return None


def pmc_source():
    d = _pmc_summary()
    if d is None:
        return "none (no committed counter profile matches the kernel sources of this tree)"
    return "%s/pmc_summary.json (committed; collected at %s, kernel sources %s)" % (
        os.path.relpath(PMC_DIR, ROOT), d.get("commit", "?"), d.get("csrc_digest"))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--stream-kib", type=int, default=256)
    ap.add_argument("--unique", type=int, default=0, help="distinct streams to generate (0 = all)")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-deflate", action="store_true", help="skip the deflate leg (config 3)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the gzip (config 4) and LZO (config 5) legs")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer legs (md_*_batch_host end to end, H2D / D2H alone)")
    ap.add_argument("--no-text-leg", action="store_true", help="skip the extra inflate measurement on round 1's workload")
    ap.add_argument("--deflate-streams", type=int, default=4096)
    ap.add_argument("--deflate-kib", type=int, default=1024)
    ap.add_argument("--deflate-steps", type=int, default=3)
    ap.add_argument("--gzip-members", type=int, default=32768, help="config 4: members of the ONE batch all ranks share")
    ap.add_argument("--inflate-waves", type=int, default=2, choices=[1, 2], help="wavefronts per stream of the inflate kernel (2 = decoder + copier, the default form)")
    ap.add_argument("--deflate-cap-mib", type=int, default=None, help="md_set_option deflate_workspace_cap_mib (default: the library's, a sixth of the device; 0 = none)")
    ap.add_argument("--profile", action="store_true", help="print the in-kernel phase profile of stream 0 (stderr)")
    return ap.parse_args(argv)


def share_device():
    """MD_BENCH_SHARE_DEVICE=1: every rank works on device 0 and the exchange goes over gloo with CPU tensors - the N > 1
    code path (sharding by bytes, weak-scaling seeds, gather of sizes and payload, max-over-ranks timing) on a box with ONE
    GPU (tests/test_gpu_multirank.py).  The numbers of such a run mean nothing; under a real launch (one device per rank,
    backend nccl = RCCL) only the backend, the device index and the device of the exchanged tensors differ."""
    return os.environ.get("MD_BENCH_SHARE_DEVICE", "") == "1"


def _xdev(dev):
    """the device of the tensors that cross ranks"""
    import torch
    return torch.device("cpu") if share_device() else dev


def respawn_command(args, argv, device_count):
    """`python bench.py --gpus N` without a launcher: the command that runs N ranks of this script under
    torch.distributed.run (None when this process is already a rank, or N <= 1).  Fails loudly when the box has
    fewer gfx950 devices than asked for."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return None
    if device_count < args.gpus and not share_device():
        raise SystemExit("bench: --gpus %d but only %d gfx950 device(s) are visible" % (args.gpus, device_count))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = ""
    try:  # a container CPU quota below the visible CPUs is what the host really gives
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = "; cgroup cpu.max = %.1f of %d visible CPUs" % (float(q) / float(per), cores)
            cores = max(1, min(cores, int(-(-float(q) // float(per)))))
    except (OSError, ValueError):
        pass
    return cores, quota


def cpu_baseline_inflate(streams, nbytes, budget_s):
    """Oracle (C port of De.Inf.Ns / Zl.Inf.Ns) on the host cores of this box, bounded sample:
    one thread first, then one stream per thread on all cores (ctypes releases the GIL)."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    from tests import oracle_lib
    orc = oracle_lib.load()

    done, k = 0, 0
    t0 = time.perf_counter()
    while k < len(streams):
        rc, used, out = orc.zl_inflate(streams[k], nbytes)
        assert rc == 0 and used == len(streams[k]) and len(out) == nbytes
        done += len(out)
        k += 1
        if time.perf_counter() - t0 > budget_s * 0.4:
            break
    single = done / 2**20 / (time.perf_counter() - t0)
    t1 = time.perf_counter()  # anchor: libz on the same sample
    for z in streams[:k]:
        zlib.decompress(z)
    dz = time.perf_counter() - t1
    cores, quota = host_cores()
    # all cores: every thread inflates its share of the batch inside ONE C call (GIL released)
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


def cpu_baseline_deflate(bufs, level, budget_s):
    """Oracle (C port of De.Lz77 + De.Def, Zl driver) on the host cores: one buffer per thread."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import oracle_lib
    orc = oracle_lib.load()
    t0 = time.perf_counter()
    orc.zl_deflate(bufs[0], level)
    single = len(bufs[0]) / 2**20 / (time.perf_counter() - t0)
    cores, quota = host_cores()
    done, t1 = 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        while time.perf_counter() - t1 < budget_s * 0.6:
            for b, _ in zip(bufs, ex.map(lambda b: orc.zl_deflate(b, level), bufs)):
                done += len(b)
    da = time.perf_counter() - t1
    return {"value": round(done / 2**20 / da, 2), "unit": "MiB/s", "cores": cores, "kind": "port",
            "single_core_value": round(single, 2),
            "sample": "oracle/de_deflate.c Zl.Def level %d, queue 4096: %d MiB over %d sampled buffers, one buffer per "
                      "thread on %d threads%s" % (level, done >> 20, len(bufs), cores, quota)}


def _copy_ms(torch, dst, src, dev, reps=3):
    """milliseconds of one dst.copy_(src) between a pinned host tensor and a device tensor (HIP events, best of reps)"""
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    return best


def host_path_inflate(eng, dev, blob, in_off, in_len, out_off, out_cap, out_bytes, d_out_ref, kernel_ms, reps=3):
    """the caller with HOST buffers (the reference's are bigarrays, lib/de.mli:93-106) - md_inflate_batch_host
    end to end on pinned buffers, beside the copies and the kernel timed alone.  d_out_ref: the device result of the
    resident path, to compare with."""
    import torch
    import decompress_amd
    h_in = eng.host_buffer(blob.nbytes)
    h_in[:] = blob
    h_out = eng.host_buffer(out_bytes)
    t_in, t_out = torch.from_numpy(h_in), torch.from_numpy(h_out)   # (pinned: md_host_alloc)
    d_a = torch.empty(blob.nbytes, dtype=torch.uint8, device=dev)
    h2d_ms = _copy_ms(torch, d_a, t_in, dev)
    d2h_ms = _copy_ms(torch, t_out, d_out_ref[:out_bytes], dev)
    del d_a
    h_out[:] = 0
    best, res = None, None
    for _ in range(reps + 1):   # (the first call grows the context's device copies)
        t0 = time.perf_counter()
        res = eng.inflate_batch_host(decompress_amd.FORMAT_ZLIB, h_in, in_off, in_len, h_out, out_off, out_cap)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    ok = bool((res[2] == 0).all()) and bool(torch.equal(torch.from_numpy(h_out).to(dev), d_out_ref[:out_bytes]))
    eng.set_option("host_pipeline_slices", 1)
    t0 = time.perf_counter()
    eng.inflate_batch_host(decompress_amd.FORMAT_ZLIB, h_in, in_off, in_len, h_out, out_off, out_cap)
    serial = (time.perf_counter() - t0) * 1e3
    eng.set_option("host_pipeline_slices", 16)
    eng.set_option("release_workspace", 0)
    return {"entry_point": "md_inflate_batch_host, pinned host buffers (md_host_alloc), slices of streams pipelined on three HIP streams",
            "end_to_end_ms": round(best, 3), "h2d_ms": round(h2d_ms, 3), "kernel_ms": round(kernel_ms, 3), "d2h_ms": round(d2h_ms, 3),
            "over_max_copy": round(best / max(h2d_ms, d2h_ms), 3), "one_slice_ms": round(serial, 3),
            "h2d_gbs": round(blob.nbytes / h2d_ms / 1e6, 1), "d2h_gbs": round(out_bytes / d2h_ms / 1e6, 1),
            "mib_per_s": round(out_bytes / 2**20 / (best * 1e-3), 1), "parity_ok": ok}



def host_path_deflate(eng, dev, d_in, n, nb, kernel_ms, d_out_ref, ref_len, ref_cap):
    """md_deflate_batch_host on pinned buffers for the C3 batch, beside the copies alone.  Output room per buffer: the
    input size + 8 KiB (the batch compresses to 0.83), so that the copy-out moves about what was produced."""
    import torch
    import decompress_amd
    cap = nb + 8192
    h_in = eng.host_buffer(n * nb)
    t_in = torch.from_numpy(h_in)
    t_in.copy_(d_in)
    h_out = eng.host_buffer(n * cap)
    t_out = torch.from_numpy(h_out)
    d_a = torch.empty(n * nb, dtype=torch.uint8, device=dev)
    h2d_ms = _copy_ms(torch, d_a, t_in, dev, reps=2)
    m = min(n * nb, n * cap)
    d2h_ms = _copy_ms(torch, t_out[:m], d_a[:m], dev, reps=2) * (n * cap) / m
    del d_a
    off = np.arange(n, dtype=np.uint64)
    best, res = None, None
    for _ in range(2):
        t0 = time.perf_counter()
        res = eng.deflate_batch_host(decompress_amd.FORMAT_ZLIB, h_in, off * nb, np.full(n, nb, dtype=np.uint64), h_out, off * cap,
                                     np.full(n, cap, dtype=np.uint64), level=6, queue=4096)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    ok = bool((res[1] == 0).all()) and bool((res[0] == ref_len.cpu().numpy().astype(np.uint64)).all())
    for k in range(0, n, max(1, n // 16)):  # bytes against the resident path's
        m = int(res[0][k])
        ok = ok and bool(torch.equal(torch.from_numpy(h_out[k * cap:k * cap + m].copy()).to(dev), d_out_ref[k * ref_cap:k * ref_cap + m]))
    eng.set_option("release_workspace", 0)
    return {"entry_point": "md_deflate_batch_host, pinned host buffers, output room = input + 8 KiB per buffer; slices of positions, "
                           "the next slice's input and the finished output columns as strided copies under the kernels",
            "end_to_end_ms": round(best, 3), "h2d_ms": round(h2d_ms, 3), "kernel_ms": round(kernel_ms, 3), "d2h_ms": round(d2h_ms, 3),
            "sum_ms": round(h2d_ms + kernel_ms + d2h_ms, 3), "over_sum": round(best / (h2d_ms + kernel_ms + d2h_ms), 3),
            "mib_per_s": round(n * nb / 2**20 / (best * 1e-3), 1), "parity_ok": ok}



def deflate_leg(args, eng, dev, rank, world, dist, fence):
    """BASELINE config 3: n x 1 MiB printable-ASCII buffers, level 6, queue 4096, Zl driver."""
    import torch
    import decompress_amd
    n, nb = args.deflate_streams, args.deflate_kib * 1024
    g = torch.Generator(device=dev)
    g.manual_seed(0xC3 + rank)
    d_in = torch.randint(0x20, 0x7f, (n * nb,), dtype=torch.uint8, device=dev, generator=g)  # uniform printable ASCII
    cap = nb + nb // 4 + 8192
    off = torch.arange(n, dtype=torch.int64, device=dev)
    d_off, d_len = off * nb, torch.full((n,), nb, dtype=torch.int64, device=dev)
    d_ooff, d_cap = off * cap, torch.full((n,), cap, dtype=torch.int64, device=dev)
    d_out = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=6, queue=4096,
                            total_in=n * nb)
    fence()
    eng.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.deflate_steps):
        res = eng.deflate_batch(decompress_amd.FORMAT_ZLIB, d_in, d_off, d_len, d_out, d_ooff, d_cap, level=6, queue=4096,
                                results=res, total_in=n * nb)
    kernel_ms = eng.timing_end() / max(1, args.deflate_steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=_xdev(dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    out_len, status, adler = res
    ok = bool((status == 0).all().item())
    comp_bytes = int(out_len.sum().item())
    sample = []
    round_trip = None
    if not args.no_verify:
        # EVERY buffer of the batch: the compressed streams are inflated on the device (one launch, outside the timed
        # region) and the result is compared with the input, byte for byte; the checksums are the ones inflate verifies
        d_back = torch.empty(n * nb, dtype=torch.uint8, device=dev)
        back = eng.inflate_batch(decompress_amd.FORMAT_ZLIB, d_out, d_ooff, out_len.to(torch.int64), d_back, d_off, d_len)
        fence()
        round_trip = bool((back[2] == 0).all().item()) and bool((back[0] == nb).all().item()) and bool(torch.equal(d_back, d_in))
        ok = ok and round_trip
        del d_back
        from tests import oracle_lib
        orc = oracle_lib.load()
        # bytes equal the oracle's (one buffer in 64); the CPU leg gets one buffer per host thread
        nsample = min(n, 64)
        ncheck = 0
        for k in range(0, n, max(1, n // nsample)):
            plain = d_in[k * nb:(k + 1) * nb].cpu().numpy().tobytes()
            ncheck += 1
            got = d_out[k * cap:k * cap + int(out_len[k].item())].cpu().numpy().tobytes()
            ok = ok and got == orc.zl_deflate(plain, 6) and zlib.decompress(got) == plain
            ok = ok and (int(adler[k].item()) & 0xffffffff) == zlib.adler32(plain)
            sample.append(plain)
    if world > 1:
        t = torch.tensor([int(ok), comp_bytes], dtype=torch.int64, device=_xdev(dev))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ok, comp_all = int(t[0].item()) == world, int(t[1].item())
    else:
        comp_all = comp_bytes
    if rank != 0:
        return None
    algo = n * nb + comp_bytes  # per launch: U read once + C written once
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    is_default = (n, nb) == (4096, 1 << 20)
    leg = {
        "metric": "MiB/s deflate over N buffers (uncompressed bytes / wall second)",
        "value": round(world * n * nb * args.deflate_steps / 2**20 / elapsed, 1), "unit": "MiB/s",
        "steps": args.deflate_steps, "ms_per_step": round(elapsed / args.deflate_steps * 1e3, 3), "parity_ok": ok,
        "parity": {"round_trip_all_buffers_on_device": round_trip, "oracle_byte_compare_buffers": (ncheck if not args.no_verify else 0)},
        "config": {"workload": "C3: %d x %d KiB printable-ASCII buffers per GPU, De.Lz77 + De.Def level 6, queue 4096, "
                               "Zl driver, dynamic blocks" % (n, args.deflate_kib),
                   "compressed_ratio": round(comp_all / (world * n * nb), 4),
                   "workspace_cap_mib": ("library default: device memory / 6 (a batch above it goes in slices of positions, same bytes)"
                                         if args.deflate_cap_mib is None else args.deflate_cap_mib)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": pmc_traffic(("deflate_link_kernel", "deflate_match_kernel", "deflate_kernel"), is_default, "deflate_calls"),
                     "traffic_source": pmc_source(),
                     "kernel": "md::defl::deflate_link_kernel + deflate_match_kernel + deflate_kernel (per step one launch of each, or "
                               "one per slice of positions where the workspace cap applies: 2 at the default cap)",
                     "kernel_ms": round(kernel_ms, 3),
                     "algorithmic_bytes_per_launch": algo},
    }
    if world == 1 and not args.no_host_path:
        try:
            leg["host_path"] = host_path_deflate(eng, dev, d_in, n, nb, kernel_ms, d_out, out_len, cap)
        except (MemoryError, RuntimeError) as e:
            leg["host_path"] = {"error": str(e)[:200]}
    if world == 1 and not args.no_cpu_baseline and sample:
        leg["cpu_baseline"] = cpu_baseline_deflate(sample, 6, args.cpu_seconds)
    return leg


def gzip_leg(args, eng, dev, rank, world, dist):
    """BASELINE config 4, strong scaling: ONE batch of args.gzip_members gzip members (32768 = the 15 files of the
    reference's test/corpus cycled, 6.64 GiB) sharded over the ranks in contiguous ranges balanced by bytes
    (shard_by_bytes); every rank runs Gz.Def level 4 (mtime 0, OS Unix, no name) then Gz.Inf on its shard; the
    compressed members of every rank are then gathered to rank 0 — the path's final gather: sizes by all_gather, bytes by
    exact-size send/recv (RCCL over xGMI).  Times are max over ranks; bytes checked against the oracle on one cycle."""
    import numpy as np
    import torch
    import decompress_amd
    from decompress_amd import shard, workloads
    n_total = args.gzip_members
    uniq = list(workloads.corpus().values())
    lengths = [len(uniq[i % len(uniq)]) for i in range(n_total)]
    spans = shard.shard_by_bytes(lengths, world)
    lo, hi = spans[rank]
    n = hi - lo
    ln = np.array(lengths[lo:hi], dtype=np.int64)
    off = np.zeros(n, dtype=np.int64)
    np.cumsum(((ln + 15) // 16 * 16)[:-1], out=off[1:])
    total = int(ln.sum())
    # the shard's input, built on the device from one copy of the 15 files
    d_files = [torch.from_numpy(np.frombuffer(u, dtype=np.uint8).copy()).to(dev) for u in uniq]
    d_in = torch.zeros(int(off[-1] + ln[-1]) + 64 if n else 64, dtype=torch.uint8, device=dev)
    for i in range(n):
        d_in[int(off[i]):int(off[i]) + int(ln[i])] = d_files[(lo + i) % len(uniq)]
    cap = (ln + 8192).astype(np.int64)
    ooff = np.zeros(n, dtype=np.int64)
    np.cumsum(((cap + 255) // 256 * 256)[:-1], out=ooff[1:])
    t = lambda a: torch.from_numpy(a).to(dev)
    d_off, d_len = t(off), t(ln)
    d_z = torch.empty(int(ooff[-1] + cap[-1]) if n else 1, dtype=torch.uint8, device=dev)
    d_zoff, d_zcap = t(ooff), t(cap)
    hdr = dict(mtime=0, os=3, hcrc=0, ascii=0, filename=None, comment=None)

    def tmax(ms):
        if world == 1:
            return ms
        tt = torch.tensor([ms], dtype=torch.float64, device=_xdev(dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    kw = dict(level=4, header=hdr, total_in=max(1, total))
    res = eng.deflate_batch(decompress_amd.FORMAT_GZIP, d_in, d_off, d_len, d_z, d_zoff, d_zcap, **kw)
    torch.cuda.synchronize(dev)
    eng.timing_begin()
    res = eng.deflate_batch(decompress_amd.FORMAT_GZIP, d_in, d_off, d_len, d_z, d_zoff, d_zcap, results=res, **kw)
    ms_def = tmax(eng.timing_end())
    z_len, z_status, _ = res
    ok = bool((z_status == 0).all().item())
    d_back = torch.zeros(d_in.numel(), dtype=torch.uint8, device=dev)
    r = eng.inflate_batch(decompress_amd.FORMAT_GZIP, d_z, d_zoff, z_len.to(torch.int64), d_back, d_off, d_len)
    torch.cuda.synchronize(dev)
    eng.timing_begin()
    for _ in range(2):
        r = eng.inflate_batch(decompress_amd.FORMAT_GZIP, d_z, d_zoff, z_len.to(torch.int64), d_back, d_off, d_len, results=r)
    ms_inf = tmax(eng.timing_end() / 2)
    out_len, consumed, status, crc = r
    ok = ok and bool((status == 0).all().item()) and bool((out_len == d_len).all().item())
    ok = ok and bool(torch.equal(d_back, d_in))
    del d_back
    # ---- the final gather: compressed members packed back to back, sizes, then the bytes to rank 0
    zl = z_len.to(torch.int64)
    zstart = torch.cumsum(zl, 0) - zl
    idx = torch.repeat_interleave(d_zoff - zstart, zl) + torch.arange(int(zl.sum().item()), device=dev)
    payload = d_z[idx]
    del idx
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    all_zl = torch.cat(shard.gather_varlen(dist, zl.to(_xdev(dev)), world))
    parts, sizes = shard.gather_payload(dist, payload.to(_xdev(dev)), world, rank, dst=0)
    torch.cuda.synchronize(dev)
    gather_ms = tmax((time.perf_counter() - t0) * 1e3)
    if rank == 0 and not args.no_verify:
        import gzip as _gz
        from tests import oracle_lib
        orc = oracle_lib.load()
        got_all = torch.cat(parts) if world > 1 else payload
        starts = (torch.cumsum(all_zl, 0) - all_zl).cpu().numpy()
        lens_all = all_zl.cpu().numpy()
        ok = ok and int(all_zl.numel()) == n_total and int(got_all.numel()) == int(lens_all.sum())
        for k in list(range(0, len(uniq), 4)) + [n_total - 1, n_total // 2]:  # members from every part of the gathered blob
            m = got_all[int(starts[k]):int(starts[k]) + int(lens_all[k])].cpu().numpy().tobytes()
            ok = ok and m == orc.gz_deflate(uniq[k % len(uniq)], level=4) and _gz.decompress(m) == uniq[k % len(uniq)]
    flags = shard.gather_results(dist, torch.tensor([int(ok), total, int(zl.sum().item())], dtype=torch.int64, device=_xdev(dev)), world)
    if rank != 0:
        return None
    ok = all(int(f[0].item()) for f in flags)
    tot_all, comp_all = float(sum(int(f[1].item()) for f in flags)), float(sum(int(f[2].item()) for f in flags))
    return {"workload": "C4: %d gzip members = the reference's 15 corpus files cycled (%d B), Gz.Def level 4 / Gz.Inf, one "
                        "batch sharded over %d GPU(s) by bytes" % (n_total, int(tot_all), world),
            "scaling": "strong", "members_total": n_total, "members_per_rank": [b - a for a, b in spans],
            "results_gathered": int(all_zl.numel()), "gathered_bytes": int(sum(sizes)), "gather_ms": round(gather_ms, 2),
            "deflate": {"value": round(tot_all / 2**20 / (ms_def * 1e-3), 1), "unit": "MiB/s", "ms": round(ms_def, 2),
                        "frac": round((tot_all + comp_all) / (ms_def * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 5)},
            "inflate": {"value": round(tot_all / 2**20 / (ms_inf * 1e-3), 1), "unit": "MiB/s", "ms": round(ms_inf, 3),
                        "frac": round((tot_all + comp_all) / (ms_inf * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 5)},
            "compressed_ratio": round(comp_all / tot_all, 4), "parity_ok": ok}


def def_ns_leg(args, eng, dev):
    """De.Def.Ns.deflate (the reference's default level 4) over 1024 x 256 KiB buffers (half word text,
    half slices of the reference's corpus), bytes checked against the oracle on a sample, every stream inflated back."""
    import ctypes
    import numpy as np
    import torch
    import decompress_amd
    from decompress_amd import workloads
    n, nb = 1024, 256 * 1024
    corpus = b"".join(workloads.corpus().values())
    uniq = [workloads.text(0xD5 + i, nb) if i % 2 == 0 else (corpus * 2)[(i * 40961) % len(corpus):][:nb] for i in range(64)]
    bufs = [uniq[i % len(uniq)] for i in range(n)]
    blob, off, ln = workloads.pack(bufs)
    cap = np.full(n, int(eng.lib.md_de_def_ns_compress_bound(nb)), dtype=np.int64)
    zoff = np.arange(n, dtype=np.int64) * ((int(cap[0]) + 255) // 256 * 256)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_off, d_len = t(blob), t(off), t(ln)
    d_z = torch.empty(int(zoff[-1] + cap[-1]) + 64, dtype=torch.uint8, device=dev)
    d_zoff, d_zcap = t(zoff), t(cap)
    out_len = torch.empty(n, dtype=torch.int64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())

    def run():
        eng._check(eng.lib.md_def_ns_batch_device(eng.ctx, decompress_amd.FORMAT_DEFLATE, 4, n * nb, n, p(d_in), p(d_off), p(d_len),
                                                  p(d_z), p(d_zoff), p(d_zcap), p(out_len), p(status), None))
    run()
    torch.cuda.synchronize(dev)
    eng.timing_begin()
    run()
    ms = eng.timing_end()
    ok = bool((status == 0).all().item())
    d_back = torch.zeros(int(blob.size) + 64, dtype=torch.uint8, device=dev)
    r = eng.inflate_batch(decompress_amd.FORMAT_DEFLATE, d_z, d_zoff, out_len, d_back, d_off, d_len)
    torch.cuda.synchronize(dev)
    ok = ok and bool((r[2] == 0).all().item()) and bool(torch.equal(d_back[:blob.size], d_in[:blob.size]))
    if not args.no_verify:
        from tests import oracle_lib
        orc = oracle_lib.load()
        zl = out_len.cpu().numpy()
        for k in range(0, len(uniq), 8):
            got = d_z[int(zoff[k]):int(zoff[k]) + int(zl[k])].cpu().numpy().tobytes()
            ok = ok and (0, got) == orc.def_ns(bufs[k], 4)
    total, comp = float(n * nb), float(out_len.sum().item())
    return {"workload": "De.Def.Ns.deflate level 4, 1024 x 256 KiB buffers (word text / corpus slices)",
            "value": round(total / 2**20 / (ms * 1e-3), 1), "unit": "MiB/s", "ms": round(ms, 2),
            "frac": round((total + comp) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "compressed_ratio": round(comp / total, 4),
            "parity_ok": ok}


def long_stream_leg(args, eng):
    """ONE long stream (N = 1 only): Zl.Higher.uncompress on 64 MiB of seeded word text (lib/zl.ml:650-666; the reference's
    tool on one big file) - decoded in pieces by the whole device, host to host from pinned buffers; beside it
    the one-pair-of-wavefronts path the same call took before round 6 (on the first 8 MiB of the same text)."""
    import ctypes
    from decompress_amd import workloads
    n = 64 << 20
    plain = workloads.text(0x51, n)
    z = zlib.compress(plain, 6)
    h_in, h_out = eng.host_buffer(len(z)), eng.host_buffer(n)
    h_in[:] = np.frombuffer(z, dtype=np.uint8)
    wrote = ctypes.c_size_t()

    def run(src_ptr, src_len, cap):
        t0 = time.perf_counter()
        st = eng.lib.md_zl_higher_uncompress(eng.ctx, ctypes.c_void_p(src_ptr), src_len, ctypes.c_void_p(h_out.ctypes.data), cap, ctypes.byref(wrote))
        return st, (time.perf_counter() - t0) * 1e3

    best = None
    for _ in range(4):
        st, ms = run(h_in.ctypes.data, len(z), n)
        best = ms if best is None or ms < best else best
    v = eng.lib.md_set_option(eng.ctx, b"inflate_parallel_last", 0)
    ok = st == 0 and wrote.value == n and h_out.tobytes() == plain
    # the serial path on a shorter stream of the same text (it runs at ~0.15 GiB/s: 8 MiB take 50 ms)
    m = 8 << 20
    z8 = zlib.compress(plain[:m], 6)
    h_in[:len(z8)] = np.frombuffer(z8, dtype=np.uint8)
    eng.set_option("inflate_parallel_min", 0)
    try:
        st8, ms8 = run(h_in.ctypes.data, len(z8), m)
    finally:
        eng.set_option("inflate_parallel_min", 96)
    ok = ok and st8 == 0 and h_out[:m].tobytes() == plain[:m]
    return {"workload": "Zl.Higher.uncompress on ONE zlib stream of 64 MiB of seeded word text (level 6, %d bytes), pinned host buffers in and out" % len(z),
            "ms": round(best, 3), "mib_per_s": round(n / 2**20 / (best * 1e-3), 1), "pieces": v & 0xffffff, "decode_rounds": v >> 24,
            "serial_path": {"what": "the same call with the pieces switched off (one pair of wavefronts), first 8 MiB of the text", "ms": round(ms8, 3),
                            "mib_per_s": round(m / 2**20 / (ms8 * 1e-3), 1)},
            "parity_ok": bool(ok)}


def lzo_leg(args, eng, dev):
    """BASELINE config 5: 8192 x 128 KiB buffers (half word text, half printable-ASCII noise; 128 distinct),
    Lzo.compress then Lzo.uncompress; bytes checked against the oracle on a sample."""
    import numpy as np
    import torch
    from decompress_amd import workloads, lzo
    n, nb = 8192, 128 * 1024
    uniq = [(workloads.text if i % 2 == 0 else workloads.ascii_uniform)(0xC5 + i, nb) for i in range(128)]
    bufs = [uniq[i % len(uniq)] for i in range(n)]
    blob, off, ln = workloads.pack(bufs, align=32)
    cap = np.full(n, lzo.max_compressed_length(nb), dtype=np.int64)
    zoff = np.arange(n, dtype=np.int64) * ((int(cap[0]) + 255) // 256 * 256)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_off, d_len = t(blob), t(off), t(ln)
    d_z = torch.empty(int(zoff[-1] + cap[-1]) + 64, dtype=torch.uint8, device=dev)
    d_zoff, d_zcap = t(zoff), t(cap)
    res = eng.lzo_batch(True, d_in, d_off, d_len, d_z, d_zoff, d_zcap)
    torch.cuda.synchronize(dev)
    eng.timing_begin()
    res = eng.lzo_batch(True, d_in, d_off, d_len, d_z, d_zoff, d_zcap, results=res)
    ms_c = eng.timing_end()
    z_len, z_st = res
    ok = bool((z_st == 0).all().item())
    d_back = torch.zeros(int(blob.size) + 64, dtype=torch.uint8, device=dev)
    r = eng.lzo_batch(False, d_z, d_zoff, z_len, d_back, d_off, d_len)
    torch.cuda.synchronize(dev)
    eng.timing_begin()
    for _ in range(3):
        r = eng.lzo_batch(False, d_z, d_zoff, z_len, d_back, d_off, d_len, results=r)
    ms_d = eng.timing_end() / 3
    ok = ok and bool((r[1] == 0).all().item()) and bool((r[0] == d_len).all().item())
    ok = ok and bool(torch.equal(d_back[:blob.size], d_in[:blob.size]))
    if not args.no_verify:
        from tests import oracle_lib
        orc = oracle_lib.load()
        zl = z_len.cpu().numpy()
        for k in range(0, len(uniq), 16):
            got = d_z[int(zoff[k]):int(zoff[k]) + int(zl[k])].cpu().numpy().tobytes()
            ok = ok and got == orc.lzo_compress(bufs[k])[1]
    total, comp = float(ln.sum()), float(z_len.sum().item())
    return {"workload": "C5: 8192 x 128 KiB buffers (half word text, half printable ASCII), Lzo.compress / Lzo.uncompress",
            "compress": {"value": round(total / 2**20 / (ms_c * 1e-3), 1), "unit": "MiB/s", "ms": round(ms_c, 2),
                         "frac": round((total + comp) / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "uncompress": {"value": round(total / 2**20 / (ms_d * 1e-3), 1), "unit": "MiB/s", "ms": round(ms_d, 3),
                           "frac": round((total + comp) / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "compressed_ratio": round(comp / total, 4), "parity_ok": ok}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        from decompress_amd import _lib
        cmd = respawn_command(args, sys.argv[1:], _lib.load().md_device_count())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        sys.exit("bench: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    dev_index = 0 if share_device() else local_rank
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if share_device():
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    import decompress_amd
    from decompress_amd import shard, workloads

    eng = decompress_amd.Engine(dev_index)
    if args.inflate_waves != 2:
        eng.set_option("inflate_waves", args.inflate_waves)
    if args.deflate_cap_mib is not None:
        eng.set_option("deflate_workspace_cap_mib", args.deflate_cap_mib)
    ranks_seen = 1
    if world > 1:
        t = torch.ones(1, dtype=torch.int64, device=_xdev(dev))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_seen = int(t.item())

    n = args.streams
    nbytes = args.stream_kib * 1024
    unique = args.unique or n
    # every rank owns different streams (seed offset by rank): weak scaling
    t_gen = time.perf_counter()
    streams = workloads.c2_streams(n, nbytes=nbytes, level=args.level, first=rank * n,
                                   unique=unique, workers=max(1, min(32, (os.cpu_count() or 8) // max(1, world))))
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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=_xdev(dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- verification (outside the timed region): every stream Ok, bytes exact
    out_len, consumed, status, checksum = results
    ok = bool((status == 0).all().item()) and bool((out_len == nbytes).all().item())
    ok = ok and bool((consumed == d_in_len).all().item())
    if not args.no_verify:
        # every output byte of every stream, compared on the device with libz's plaintexts; every Adler-32 the kernel
        # reports with libz's
        host_sum = checksum.cpu().numpy().view(np.uint32)
        plains = [zlib.decompress(z) for z in streams]
        want_sum = np.array([zlib.adler32(p_) for p_ in plains], dtype=np.uint32)
        ok = ok and all(len(p_) == nbytes for p_ in plains) and bool((host_sum == want_sum).all())
        if ok:
            expect = torch.from_numpy(np.frombuffer(b"".join(plains), dtype=np.uint8).copy()).to(dev)
            ok = bool(torch.equal(d_out[: n * nbytes], expect))
            del expect
        del plains
    # the path's only exchange: every rank's per-stream results (sizes, Adler-32 from the kernel), RCCL all_gather
    sums = checksum.to(torch.int64).bitwise_and(0xffffffff)
    all_len = torch.cat(shard.gather_varlen(dist, out_len.to(_xdev(dev)), world))
    all_sum = torch.cat(shard.gather_varlen(dist, sums.to(_xdev(dev)), world))
    flags = shard.gather_results(dist, torch.tensor([int(ok)], dtype=torch.int64, device=_xdev(dev)), world)
    ok = all(int(x.item()) for x in flags)
    ok = ok and all_len.numel() == world * n and bool((all_len == nbytes).all().item())
    digest = int(all_sum.sum().item()) & 0xffffffffffff
    if not ok:
        print("bench: PARITY FAILURE (status/bytes/checksum mismatch)", file=sys.stderr)

    line = None
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
            "ranks_seen": ranks_seen,
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
                "workload": "C2: %d x %d KiB zlib streams per GPU, dynamic Huffman (libz level %d; even streams = "
                            "slices of the reference's test/corpus, odd = order-2 Markov ASCII text, 64 symbols, seed 0xC2 + i, SURVEY 8(d)), "
                            "Zl.Inf.Ns semantics, one stream per pair of wavefronts (decoder + copier)" % (n, args.stream_kib, args.level),
                "streams_per_gpu": n, "stream_bytes": nbytes, "unique_streams": unique,
                "compressed_ratio": round(comp_bytes / (n * nbytes), 4), "gen_seconds": round(t_gen, 1),
                "results_gathered": int(all_len.numel()), "result_digest": digest,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic(("inflate_wave_kernel",), (n, nbytes) == (4096, 262144)),
                "traffic_source": pmc_source(),
                "kernel": "md::wv::inflate_wave_kernel", "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_inflate(streams, nbytes, args.cpu_seconds)
    if world == 1 and not args.no_host_path:
        # the same batch from HOST buffers (outside the timed region of `value`, which stays the resident-buffer figure)
        try:
            line["host_path"] = host_path_inflate(eng, dev, blob, in_off, in_len, out_off, out_cap, n * nbytes, d_out, kernel_ms)
        except (MemoryError, RuntimeError) as e:
            line["host_path"] = {"error": str(e)[:200]}
    if not args.no_text_leg and world == 1 and (n, nbytes) == (4096, 262144):
        # the same kernel on round 1's stand-in workload (every stream seeded word text, 512 distinct), so that
        # BENCH_r01's 10.99 ms/step has a like-for-like successor; outside the timed region of `value`
        ts = [workloads._c2_text_one((0xC2 + i, nbytes, args.level)) for i in range(512)]
        tblob, toff, tlen = workloads.pack([ts[i % 512] for i in range(n)], align=16)
        td_in, td_off, td_len = t(tblob), t(toff), t(tlen)
        r2 = eng.inflate_batch(decompress_amd.FORMAT_ZLIB, td_in, td_off, td_len, d_out, d_out_off, d_out_cap)
        torch.cuda.synchronize(dev)
        eng.timing_begin()
        for _ in range(5):
            r2 = eng.inflate_batch(decompress_amd.FORMAT_ZLIB, td_in, td_off, td_len, d_out, d_out_off, d_out_cap, r2)
        tms = eng.timing_end() / 5
        line["r01_workload"] = {"workload": "4096 x 256 KiB zlib streams of seeded word text (BENCH_r01's input)",
                                "kernel_ms": round(tms, 4), "parity_ok": bool((r2[2] == 0).all().item()) and
                                bool((r2[0] == nbytes).all().item()),
                                "frac": round((int(tlen.sum()) + n * nbytes) / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        del td_in
    del d_in, d_out
    torch.cuda.empty_cache()
    if not args.no_deflate:
        leg = deflate_leg(args, eng, dev, rank, world, dist, fence)
        if rank == 0:
            line["deflate"] = leg
    if not args.no_secondary:  # configs 4 (one batch sharded over the ranks, outputs gathered) and 5 (not part of `value`)
        torch.cuda.empty_cache()
        leg = gzip_leg(args, eng, dev, rank, world, dist)
        torch.cuda.empty_cache()
        lz = lzo_leg(args, eng, dev) if world == 1 else None
        torch.cuda.empty_cache()
        dn = def_ns_leg(args, eng, dev) if world == 1 else None
        ls = long_stream_leg(args, eng) if world == 1 else None
        if rank == 0:
            line["gzip"] = leg
            if lz:
                line["lzo"] = lz
            if dn:
                line["def_ns"] = dn
            if ls:
                line["long_stream"] = ls
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
