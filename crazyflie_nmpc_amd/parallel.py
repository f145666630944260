"""Multi-GPU plumbing: NMPC instances are independent, so a batch shards
across ranks with NO data-path collective; torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only to aggregate the report."""
from __future__ import annotations

import numpy as np

BASE_SEED = 20200103


def shard_seed(rank: int) -> int:
    """Every rank draws its own shard of the synthetic fleet."""
    return BASE_SEED + int(rank)


def shard_range(total: int, rank: int, world: int):
    """Contiguous index range of `rank` when `total` instances are split over `world` ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_horizon(horizons, world: int):
    """Mixed-horizon batches (config C5): bucket by N (wave-homogeneous), then deal buckets out
    so that sum(N_i) -- the cost model -- is balanced.  Returns a list of index arrays per rank."""
    horizons = np.asarray(horizons)
    order = np.argsort(-horizons, kind="stable")
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += horizons[i]
    return [np.array(sorted(ix), dtype=np.int64) for ix in out]


def aggregate_report(elapsed: float, sums, dist=None, device=None):
    """max-over-ranks of the timed region and sum-over-ranks of additive statistics.
    `dist` is torch.distributed (initialised) or None for a single process."""
    sums = np.asarray(sums, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed), sums
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    s = torch.tensor(sums, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), s.cpu().numpy()
