"""Multi-GPU sharding of independent trains (one process per GPU, torch.distributed).

The hot path partitions by train: apply / hadamard / + are per-core independent and the chain
recurrences (dot, orthogonalize, tt_compress!) are sequential only WITHIN a train, so a batch of
trains shards across ranks with no data-path collective (SURVEY §8e "replicas / trains in flight").
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for barriers, the
max-over-ranks timing reduction and gathering small per-train results (ranks, norms).
Core-wise sharding of ONE long chain with neighbour hand-offs lives in pipeline.py.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def partition(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced partition of n_units over `world` ranks: [start, stop) of `rank`
    (strong scaling: the total is fixed).  The first n_units % world ranks get one extra unit."""
    assert 0 <= rank < world and n_units >= 0
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def weak_train_ids(rank: int, world: int, per_rank: int) -> List[int]:
    """Global train indices of `rank` under weak scaling (per_rank trains on every GPU)."""
    assert 0 <= rank < world and per_rank >= 0
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def max_over_ranks(value: float, dist=None, device="cpu") -> float:
    """MAX all-reduce of a Python float (timing reduction of the bench contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_lists(local: Sequence, dist=None) -> List:
    """Concatenate per-rank lists in rank order on every rank (small metadata only: ranks, norms)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local))
    flat = []
    for part in out:
        flat.extend(part)
    return flat


def cores_per_second(world: int, per_rank: int, d: int, seconds_per_step: float) -> float:
    """Whole-job throughput: all trains of all ranks times d cores, over the max-over-ranks step time."""
    return world * per_rank * d / seconds_per_step
