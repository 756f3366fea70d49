"""Multi-GPU, one process per GPU: problems sharded by contiguous batch ranges, no data-path
collective; a single gather of the per-problem results (f, c) to rank 0 at the end
(BASELINE.json north_star).  Jacobian values stay resident on the GPU that
produced them: at config 5 they are 6.2 GB per rank, ~18x the kernel time over xGMI.

Backend-agnostic (`nccl` = RCCL on the GPUs, `gloo` on CPU for the tests).  The same exchange from a
host that has no torch.distributed (the reference's Julia, a C program) is include/qln_multi.h.

Shards are ragged in general: a rank's constraint buffer holds sum_b round_up(18N - k_trans(b) + 16, align)
doubles, which differs between ranks as soon as k_trans varies per problem (BASELINE.json configs[3]) or the
batch does not divide by the world size -- so the gather exchanges the lengths first.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(n_problems: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of the global problem index owned by `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(n_problems, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_sizes(numel: int, like, group=None) -> List[int]:
    """Every rank's element count, on every rank (one small all_gather)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine = torch.tensor([int(numel)], dtype=torch.int64, device=like.device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    return [int(s.item()) for s in sizes]


def gather_to_root(t, dst: int = 0, group=None) -> Optional[list]:
    """Gather 1-D tensors -- of possibly different lengths -- to group rank `dst`.
    Returns the list of per-rank tensors on dst, None elsewhere.

    Equal lengths: one `gather`.  Unequal: the root posts one receive per peer and every peer one send, as one
    batch (ncclGroupStart/End under RCCL -- exactly what ncclGather does inside), so nothing is padded or copied.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [t]
    t = t.contiguous().reshape(-1)
    sizes = gather_sizes(t.numel(), t, group)
    if all(s == sizes[0] for s in sizes):
        if rank == dst:
            out = [torch.empty_like(t) for _ in range(world)]
            dist.gather(t, gather_list=out, dst=_global(dst, group), group=group)
            return out
        dist.gather(t, gather_list=None, dst=_global(dst, group), group=group)
        return None
    if min(sizes) == 0:
        # A rank with an empty shard must still take part: under the nccl backend communicators (and their P2P channels)
        # are created lazily inside the first collective / batch, and a rank that sits the exchange out leaves its peers'
        # initialisation waiting.  So with an empty shard anywhere, every rank joins ONE equal-size gather of its data
        # padded to the longest shard, and the root cuts the padding off (an edge case: the copy does not matter).
        n = max(sizes)
        padded = t if t.numel() == n else torch.cat([t, torch.zeros(n - t.numel(), dtype=t.dtype, device=t.device)])
        if rank == dst:
            out = [torch.empty_like(padded) for _ in range(world)]
            dist.gather(padded, gather_list=out, dst=_global(dst, group), group=group)
            return [o[: sizes[r]] for r, o in enumerate(out)]
        dist.gather(padded, gather_list=None, dst=_global(dst, group), group=group)
        return None
    if rank == dst:
        out = [t if r == dst else torch.empty(sizes[r], dtype=t.dtype, device=t.device) for r in range(world)]
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, out[r], _global(r, group), group) for r in range(world) if r != dst]):
            w.wait()
        return out
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, _global(dst, group), group)]):
        w.wait()
    return None


def _global(group_rank: int, group) -> int:
    import torch.distributed as dist

    return group_rank if group is None else dist.get_global_rank(group, group_rank)


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
