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
    r = combine_stats(elapsed, units, dist, device)
    return r["elapsed_max"], r["units"]


def combine_stats(elapsed: float, units: float, dist=None, device=None) -> dict:
    """The measurement bracket's reductions: MAX and MIN of the ranks' elapsed times, SUM of their units, and
    `ranks_seen` = SUM of ones — the number of ranks that actually took part in the collective (RCCL on GPUs), which a
    reader can hold against `n_gpus`."""
    if dist is None:
        return {"elapsed_max": elapsed, "elapsed_min": elapsed, "units": units, "ranks_seen": 1}
    import torch
    t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([units, 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    t, c = t.cpu(), c.cpu()
    return {"elapsed_max": float(t[0]), "elapsed_min": float(-t[1]), "units": float(c[0]), "ranks_seen": int(round(float(c[1])))}


# ---- synthetic input addressed by GLOBAL element index (counter hash of (seed, global index)) ---------------
# A strong-scaling shard must hold the same data whatever the number of ranks, and a test must be able to rebuild any vector
# on the host from its global index alone: value(i) = (hash32(seed, i) >> 8) * 2^-23 - 1, uniform on [-1, 1) in steps of
# 2^-23 (exact in float and double).  The torch and numpy versions below give identical bits.
_M32 = 0xFFFFFFFF


def _mix_seed(seed: int) -> int:
    return (seed * 0x85EBCA6B + 0xC2B2AE35) & _M32


def global_uniform_np(first: int, count: int, seed: int, dtype):
    import numpy as np
    i = np.arange(first, first + count, dtype=np.uint64)
    x = (i & np.uint64(_M32)) ^ (((i >> np.uint64(32)) * np.uint64(0x9E3779B1)) & np.uint64(_M32)) ^ np.uint64(_mix_seed(seed))
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & np.uint64(_M32)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & np.uint64(_M32)
    x ^= x >> np.uint64(16)
    return ((x >> np.uint64(8)).astype(np.float64) * 2.0 ** -23 - 1.0).astype(dtype)


def global_uniform_(out, first: int, seed: int, chunk: int = 1 << 26):
    """Fill the torch tensor `out` (any device, float32 / float64, contiguous) with value(first + k) for its k-th element."""
    import torch
    flat = out.view(-1)
    n = flat.numel()
    sm = _mix_seed(seed)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        i = torch.arange(first + lo, first + lo + m, dtype=torch.int64, device=out.device)
        x = (i & _M32) ^ (((i >> 32) * 0x9E3779B1) & _M32) ^ sm
        del i
        x ^= x >> 16; x *= 0x7FEB352D; x &= _M32       # int64 products wrap modulo 2^64: the low 32 bits are exact
        x ^= x >> 15; x *= 0x846CA68B; x &= _M32
        x ^= x >> 16
        x >>= 8
        flat[lo:lo + m] = x.to(torch.float64).mul_(2.0 ** -23).sub_(1.0).to(out.dtype)
        del x
    return out
