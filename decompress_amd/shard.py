"""Multi-GPU sharding of a batch of independent streams.

Streams never talk to each other, so the data path has no collective: rank r
owns the contiguous stream range shard_range(n, r, world) and runs the same
kernels on it.  The only exchange is the final gather of per-stream results
(sizes / status / checksums) — one all_gather over RCCL (backend "nccl" on
ROCm) or gloo (CPU tests)."""


def shard_range(n_total, rank, world):
    """Contiguous, balanced [lo, hi) of the n_total streams owned by `rank`."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_results(dist, local, world):
    """all_gather of one equal-length 1-D tensor per rank -> list ordered by rank."""
    if world == 1:
        return [local]
    out = [local.new_zeros(local.shape) for _ in range(world)]
    dist.all_gather(out, local)
    return out


def gather_varlen(dist, local, world):
    """Gather 1-D tensors of different lengths (sizes exchanged first, payload padded)."""
    import torch
    if world == 1:
        return [local]
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = max(int(s.item()) for s in sizes)
    pad = local.new_zeros(m)
    pad[: local.numel()] = local
    outs = [local.new_zeros(m) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[: int(s.item())] for o, s in zip(outs, sizes)]


def gather_payload(dist, payload, world, rank, dst=0):
    """The final gather of the path: every rank's output bytes, of different lengths, to rank `dst`.
    Sizes first (one all_gather of a word per rank), then the bytes as grouped point-to-point transfers of their exact
    sizes — no padding to the longest shard (RCCL send/recv over xGMI with backend "nccl"; gloo on CPU).
    payload: 1-D uint8 tensor (this rank's packed outputs).  Returns (list of per-rank uint8 tensors, sizes) on `dst`,
    (None, sizes) elsewhere."""
    import torch
    if world == 1:
        return [payload], [int(payload.numel())]
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if rank == dst:
        parts = [payload if r == dst else payload.new_empty(sizes[r]) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, parts[r], r) for r in range(world) if r != dst and sizes[r] > 0]
    else:
        parts = None
        ops = [dist.P2POp(dist.isend, payload, dst)] if sizes[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return parts, sizes


def shard_by_bytes(lengths, world):
    """Contiguous shards of a batch balanced by the sum of `lengths` ("contiguous ranges of the stream
    index balanced by the uncompressed bytes"): -> [(lo, hi)] per rank.  With equal lengths this is shard_range."""
    n = len(lengths)
    total = float(sum(lengths))
    spans, lo, acc = [], 0, 0.0
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        while hi < n and acc + lengths[hi] / 2.0 <= target:
            acc += lengths[hi]
            hi += 1
        if r == world - 1:
            hi = n
        spans.append((lo, hi))
        lo = hi
    return spans
