"""Multi-GPU sharding of the candidate axis: one process per GPU, torch.distributed for
the exchange (backend "nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Every rank scores its slice [n*rank/world, n*(rank+1)/world) of every pair's candidate list
and reports, per pair, the exact first minimum inside the slice.  The per-shard best scores
are min-all-reduced (the global best score per pair) and the small per-pair records
(cost, index, angle, near-tie flag) are all-gathered; ``mm_merge_shards`` (host, C ABI) then
picks the reference's winner -- the first index of minimal cost over the whole axis,
process_utils.rs:72 -- identically on every rank.  Message size: 28 B x pairs x ranks
(2044 pairs x 8 ranks = 0.46 MB): latency-bound, not xGMI-bandwidth-bound.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import _native as N


def shard_bounds(n: int, rank: int, world: int):
    """This rank's share of a candidate list of length n."""
    return (n * rank) // world, (n * (rank + 1)) // world


def merge_shards(world: int, cost: np.ndarray, uniform: np.ndarray, angle: np.ndarray, idx: np.ndarray,
                 tol: Optional[np.ndarray]):
    """Host merge of [world, n] per-shard arrays (``mm_merge_shards``)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(world, -1)
    n = cost.shape[1]
    uniform = np.ascontiguousarray(uniform, dtype=np.int32).reshape(world, n)
    angle = np.ascontiguousarray(angle, dtype=np.float64).reshape(world, n)
    idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(world, n)
    tol = None if tol is None else np.ascontiguousarray(tol, dtype=np.float64)
    ok = np.zeros(n, dtype=np.uint8)
    out_angle = np.zeros(n, dtype=np.float64)
    out_idx = np.zeros(n, dtype=np.int32)
    out_cost = np.zeros(n, dtype=np.float64)
    N.check(N.lib().mm_merge_shards(world, n, N._ptr(cost), N._ptr(uniform), N._ptr(angle), N._ptr(idx), N._ptr(tol),
                                    N._ptr(ok), N._ptr(out_angle), N._ptr(out_idx), N._ptr(out_cost)),
            "mm_merge_shards")
    return ok, out_angle, out_idx, out_cost


def merge_level(local: Dict[str, np.ndarray], tol: Optional[np.ndarray], group=None):
    """Exchange one level's per-shard results between the ranks of `group` and merge them.
    Without an initialised process group (or world == 1) this is the single-rank merge."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return merge_shards(1, local["cost"], local["uniform"], local["angle"], local["idx"], tol)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n = local["cost"].shape[0]
    # one packed f64 record per pair: [cost, angle, idx, uniform] (idx/uniform are small ints: exact in f64)
    rec = torch.from_numpy(np.stack([local["cost"], local["angle"], local["idx"].astype(np.float64),
                                     local["uniform"].astype(np.float64)], axis=0)).to(dev)
    gathered = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(gathered, rec, group=group)
    # the per-shard best score, min-reduced over the ranks (RCCL all-reduce on the GPU path)
    gbest = rec[0].clone()
    dist.all_reduce(gbest, op=dist.ReduceOp.MIN, group=group)
    g = torch.stack(gathered, dim=0).cpu().numpy()     # [world, 4, n]
    ok, angle, idx, cost = merge_shards(world, g[:, 0, :], g[:, 3, :].astype(np.int32), g[:, 1, :],
                                        g[:, 2, :].astype(np.int32), tol)
    if not np.array_equal(cost, gbest.cpu().numpy()):
        raise RuntimeError("all-reduced best score disagrees with the gathered shards")
    return ok, angle, idx, cost
