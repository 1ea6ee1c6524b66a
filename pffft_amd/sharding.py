"""Batch sharding across GPUs: the path has no exchange step, so ranks own contiguous
slices of the batch and never talk on the data path.  The only communication is the measurement bracket:
barrier, then MAX of the elapsed time and SUM of the units processed (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [start, start+count) of `total` vectors owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def timed_steps(step: Callable[[], None], steps: int, warmup: int, dist=None,
                sync: Optional[Callable[[], None]] = None) -> float:
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync on both
    sides.  Returns this rank's elapsed seconds."""
    def bracket():
        if dist is not None:
            dist.barrier()
        if sync is not None:
            sync()
    for _ in range(warmup):
        step()
    bracket()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if sync is not None:
        sync()
    elapsed = time.perf_counter() - t0
    bracket()
    return elapsed


def combine(elapsed: float, units: float, dist=None, device=None) -> Tuple[float, float]:
    """(max over ranks of elapsed, sum over ranks of units)."""
    if dist is None:
        return elapsed, units
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())
