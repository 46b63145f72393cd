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
