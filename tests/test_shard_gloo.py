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


def _payload_worker(rank, world, port, q):
    """Every rank deflates-by-stand-in its byte-balanced shard of 13 buffers of different sizes (the bytes themselves
    stand in for the kernels' output) and the packed outputs are gathered to rank 0: sizes, then exact-size transfers."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [(i * 7919) % 1000 + (0 if i == 5 else 1) for i in range(13)]  # ragged, one empty
    spans = shard.shard_by_bytes(lengths, world)
    lo, hi = spans[rank]
    outs = [bytes((i * 31 + k) & 0xff for k in range(lengths[i])) for i in range(lo, hi)]
    payload = torch.frombuffer(bytearray(b"".join(outs)) or bytearray(1), dtype=torch.uint8)[:sum(len(o) for o in outs)]
    lens = torch.tensor([len(o) for o in outs], dtype=torch.int64)
    all_lens = torch.cat(shard.gather_varlen(dist, lens, world))
    parts, sizes = shard.gather_payload(dist, payload, world, rank, dst=0)
    if rank == 0:
        q.put((all_lens.tolist(), sizes, [bytes(p.numpy().tobytes()) for p in parts], spans))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_payload_gather_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_payload_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    all_lens, sizes, parts, spans = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = [(i * 7919) % 1000 + (0 if i == 5 else 1) for i in range(13)]
    assert all_lens == lengths  # stream order = rank order of contiguous shards
    want = [bytes((i * 31 + k) & 0xff for k in range(lengths[i])) for i in range(13)]
    assert b"".join(parts) == b"".join(want)
    assert sizes == [sum(lengths[a:b]) for a, b in spans] and len(spans) == world
