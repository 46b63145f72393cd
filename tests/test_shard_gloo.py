"""N>1 path on CPU: world_size=2 over gloo (the GPU run uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from decompress_amd import shard


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 11
    lo, hi = shard.shard_range(n_total, rank, world)
    # stand-in for the per-stream results a rank's kernels produce: out_len and adler of stream i
    out_len = torch.tensor([1000 + i for i in range(lo, hi)], dtype=torch.int64)
    adler = torch.tensor([(i * 2654435761) & 0xffffffff for i in range(lo, hi)], dtype=torch.int64)
    lens = torch.cat(shard.gather_varlen(dist, out_len, world))
    sums = torch.cat(shard.gather_varlen(dist, adler, world))
    digest = shard.gather_results(dist, torch.tensor([int(adler.sum())], dtype=torch.int64), world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
    if rank == 0:
        q.put((lens.tolist(), sums.tolist(), [int(d.item()) for d in digest], float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    lens, sums, digest, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert lens == [1000 + i for i in range(11)]
    assert sums == [(i * 2654435761) & 0xffffffff for i in range(11)]
    assert sum(digest) == sum(sums)
    assert tmax == 2.0
