"""Worker of test_distributed_gloo.py: world_size-2 (or more) `gloo` run of the candidate-axis
sharding on CPU.  Each rank takes its slice of every pair's candidate list, computes the
slice's exact first minimum from oracle costs (no GPU here, so the product's local search cannot run: the
real sharded search on the device is tests/test_gpu_sharded.py), exchanges through
multimoda_rs_amd.distributed.merge_level (all_gather) and checks the merged winner against the unsharded
first minimum; then the key encoding of the device-side exchange is reduced with two all_reduce(MIN) over
the same process group and must decode to the same winners and decided flags."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def local_from_costs(costs, angles, tol, rank, world, D):
    n = len(costs)
    lo, hi = D.shard_bounds(n, rank, world)
    if hi <= lo:
        return np.inf, 1, 0.0, -1
    sl = costs[lo:hi]
    k = int(np.argmin(sl))                      # first minimum inside the slice
    near = [lo + i for i in range(len(sl)) if sl[i] <= sl[k] + tol]
    uniform = int(all(angles[i].tobytes() == angles[near[0]].tobytes() for i in near))
    return float(sl[k]), uniform, float(angles[lo + k]), lo + k


def main():
    import torch.distributed as dist
    import multimoda_rs_amd as mm
    from multimoda_rs_amd import distributed as D
    from oracle import oracle as orc
    from helpers import blob

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(123)            # same data on every rank
    angles, _, _ = mm.search_angles(5.0, 180.0)  # 73 candidates, first and last are both -pi
    jobs = []
    for p in range(6):
        ref = blob(rng, 60)
        tgt = blob(rng, 60)
        jobs.append(orc.costs_over_angles(ref, tgt, angles, 4.5, 4.5))
    # crafted: best angle is +-pi -> duplicate candidates 0 and 72 tie exactly (different ranks)
    ref = blob(rng, 80)
    jobs.append(orc.costs_over_angles(ref, (ref - 4.5) * -1.0 + 4.5, angles, 4.5, 4.5))
    # crafted: two different angles tie within the tolerance -> undecided (ok == 0)
    c = jobs[0].copy()
    c[5] = c.min() - 1.0
    c[60] = c[5] + 1e-13
    jobs.append(c)
    # crafted: fewer candidates than ranks would leave empty slices
    short_angles = angles[:1]
    tol = np.full(len(jobs) + 1, 1e-11)

    n = len(jobs) + 1
    local = {"cost": np.zeros(n), "uniform": np.zeros(n, np.int32), "angle": np.zeros(n), "idx": np.zeros(n, np.int32)}
    for j, cj in enumerate(jobs):
        local["cost"][j], local["uniform"][j], local["angle"][j], local["idx"][j] = \
            local_from_costs(cj, angles, tol[j], rank, world, D)
    cs = np.array([0.25])
    local["cost"][n - 1], local["uniform"][n - 1], local["angle"][n - 1], local["idx"][n - 1] = \
        local_from_costs(cs, short_angles, tol[n - 1], rank, world, D)

    ok, angle, idx, cost = D.merge_level(local, tol)

    # The device-side exchange (multimoda_rs_amd.distributed.search_device) reduces the same records with two
    # all_reduce(MIN) calls; on the GPU the records are written by k_export_cost / k_export_keys and decoded by
    # mm_within_plan_level_commit_dev.  Here the same encoding is written in numpy and reduced over the real process
    # group: the reduced keys must give the winner and the decided / undecided flag of mm_merge_shards.
    import torch
    I64MAX, I64MIN = np.iinfo(np.int64).max, np.iinfo(np.int64).min
    g = torch.from_numpy(local["cost"].copy())
    dist.all_reduce(g, op=dist.ReduceOp.MIN)
    g = g.numpy()
    keys = np.full(3 * n, I64MAX, dtype=np.int64)
    abits = local["angle"].view(np.int64)
    for j in range(n):
        if local["idx"][j] < 0:
            continue
        if local["cost"][j] == g[j]:
            keys[j] = local["idx"][j]
        if local["cost"][j] <= g[j] + tol[j]:
            keys[n + j] = abits[j] if local["uniform"][j] else I64MIN
            keys[2 * n + j] = ~abits[j] if local["uniform"][j] else I64MIN
    k = torch.from_numpy(keys)
    dist.all_reduce(k, op=dist.ReduceOp.MIN)
    k = k.numpy()
    for j in range(n):
        assert g[j] == cost[j]
        assert k[j] == idx[j], (j, k[j], idx[j])
        assert bool(k[n + j] == ~k[2 * n + j]) == bool(ok[j]), j
        if ok[j]:
            assert k[n + j] == np.float64(angle[j]).view(np.int64)

    for j, cj in enumerate(jobs):
        k = int(np.argmin(cj))
        assert idx[j] == k, (j, idx[j], k)
        assert cost[j] == cj[k] and angle[j] == angles[k]
    assert list(ok[:6]) == [1] * 6
    assert ok[6] == 1 and idx[6] == 0 and angle[6] == -math.pi      # duplicates: lowest index, decided
    assert ok[7] == 0 and idx[7] == 5                               # genuine near-tie: undecided
    assert idx[n - 1] == 0 and cost[n - 1] == 0.25 and ok[n - 1] == 1
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("GLOO_SHARD_OK")


if __name__ == "__main__":
    main()
