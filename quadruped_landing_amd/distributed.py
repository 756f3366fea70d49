"""Multi-GPU: one process per GPU, problems sharded by contiguous batch ranges, no data-path
collective; a single gather of the per-problem results (f, c) to rank 0 at the end
(BASELINE.json north_star).  Jacobian values stay resident on the GPU that
produced them: at config 5 they are 6.2 GB per rank, ~18x the kernel time over xGMI.

Backend-agnostic (`nccl` = RCCL on the GPUs, `gloo` on CPU for the tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_problems: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of the global problem index owned by `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(n_problems, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_to_root(t, dst: int = 0, group=None):
    """Gather equally-sized 1-D tensors to `dst`; returns the list on dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    world = dist.get_world_size(group)
    if dist.get_rank(group) == dst:
        out = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, gather_list=out, dst=dst, group=group)
        return out
    dist.gather(t, gather_list=None, dst=dst, group=group)
    return None


def gather_results(f, c, dst: int = 0, group=None):
    """The end-of-job exchange: objective values and constraint vectors of every shard to rank 0."""
    return gather_to_root(f, dst, group), gather_to_root(c, dst, group)


def max_over_ranks(value: float, device=None, group=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
