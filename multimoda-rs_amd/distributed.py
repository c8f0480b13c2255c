"""Multi-GPU sharding of the candidate axis: one process per GPU, torch.distributed for
the exchange (backend "nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Every rank scores its slice [n*rank/world, n*(rank+1)/world) of every pair's candidate list
and holds, per pair, the exact first minimum inside the slice.  The reference's winner is the
first index of minimal cost over the whole axis (process_utils.rs:72).

A shard is a tile of the (frame pair x candidate) grid: ``shard_grid(world, n_jobs)`` gives the default
(pair_blocks, cand_slices); pair_blocks = 1 is the pure candidate-axis split.  The exchange is the same for every grid
(a rank exports +inf / INT64_MAX for what it does not own).

Three exchanges, same result (identical on every rank):

``rccl`` (default when the process group's backend is nccl)
    the ``device`` exchange below issued by the LIBRARY on its own RCCL communicator
    (``mm_within_plan_search_sharded``: ncclAllReduce(MIN) on the engine's stream between the export kernels; the
    path a non-Python host uses).  The communicator's id travels over the torch process group once.
``device`` (the checker of ``rccl``; default with a CPU backend such as gloo)
    per level two RCCL all-reduces on DEVICE buffers, stream-ordered behind the search kernels on
    the engine's stream -- no host round trip between the search and the collectives:
      1. all_reduce(MIN) of the per-shard best cost (f64 x pairs)
      2. all_reduce(MIN) of 3 x pairs int64 keys: the best index masked by cost == global min, and the
         (bits, ~bits) of the near-tie shards' angles (their min / max: equal <=> the step is decided)
    then ONE D2H copy of the reduced records (32 B x pairs) and the commit.  Message sizes for
    config3: 16 KB + 48 KB per rank -- latency-bound, not xGMI-bandwidth-bound.
``gather`` (MM_EXCHANGE=gather; the checker)
    the per-pair records (cost, index, angle, near-tie flag) go through the host, are all-gathered,
    and ``mm_merge_shards`` (host, C ABI) picks the winner.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import _native as N


def shard_bounds(n: int, rank: int, world: int):
    """This rank's share of a candidate list of length n."""
    return (n * rank) // world, (n * (rank + 1)) // world


def exchange_mode(group=None) -> str:
    """MM_EXCHANGE = rccl | device | gather; default: ``device`` (torch's all-reduces on the device records).

    ``rccl`` -- the library's own communicator, ``mm_within_plan_search_sharded`` -- is OPT-IN: it has run at world = 1, in
    lock-step in-process emulation and over gloo, never at world > 1 on hardware, so its parity there is UNPINNED
    (ADVICE r3).  ``bench.py`` opts in on an nccl group after running one case through all three exchanges and comparing
    them.  With ``rccl`` a non-OK return of ``search_sharded*`` on ANY rank means the job must be torn down: the peers of a
    rank that failed between the two collectives are blocked inside RCCL."""
    m = os.environ.get("MM_EXCHANGE", "") or "device"
    if m not in ("rccl", "device", "gather"):
        raise ValueError("MM_EXCHANGE must be 'rccl', 'device' or 'gather'")
    return m


def shard_grid(world: int, n_jobs: int):
    """Default tile shape (pair_blocks, cand_slices) of `world` ranks (``mm_shard_grid``); MM_SHARD_GRID=PxC overrides."""
    g = os.environ.get("MM_SHARD_GRID", "")
    if g:
        pb, cs = (int(v) for v in g.lower().split("x"))
        if pb * cs != world:
            raise ValueError(f"MM_SHARD_GRID={g} does not multiply to the world size {world}")
        return pb, cs
    return N.shard_grid(world, n_jobs)


_native_comms: Dict[object, "N.Comm"] = {}


def native_comm(group=None) -> "N.Comm":
    """The library's RCCL communicator over the ranks of a torch process group: rank 0 makes the id, it is
    broadcast over the group once, every rank joins (collective).  Cached per group."""
    import torch
    import torch.distributed as dist
    key = group if group is not None else "default"
    c = _native_comms.get(key)
    if c is not None:
        return c
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    on_dev = dist.get_backend(group) == "nccl"
    # Every step up to the collective join is decided by ALL ranks together: a rank that cannot load RCCL, or a rank 0
    # that cannot make the id, must not leave the others waiting in the broadcast.
    ok = torch.tensor([1 if N.lib().mm_comm_version() >= 0 else 0], dtype=torch.int32)
    ok = ok.to(dev) if on_dev else ok
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        raise RuntimeError("RCCL cannot be loaded on at least one rank (mm_comm_version() < 0)")
    uid, err = bytes(N.Comm.ID_BYTES), 0
    if rank == 0:
        try:
            uid = N.Comm.unique_id()
        except Exception:
            err = 1
    t = torch.frombuffer(bytearray(bytes([err]) + uid), dtype=torch.uint8)
    t = t.to(dev) if on_dev else t.clone()
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    raw = bytes(t.cpu().numpy().tobytes())
    if raw[0]:
        raise RuntimeError("rank 0 could not make the communicator id (mm_comm_unique_id)")
    c = N.Comm(raw[1:], rank, world, dev.index)
    _native_comms[key] = c
    return c


def close_native_comms():
    for c in _native_comms.values():
        c.close()
    _native_comms.clear()


def world_size(group=None) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


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
    """The ``gather`` exchange of one level: per-shard records through the host, all-gathered and merged.
    Without an initialised process group (or world == 1) this is the single-rank merge."""
    import torch
    import torch.distributed as dist

    world = world_size(group)
    if world == 1:
        return merge_shards(1, local["cost"], local["uniform"], local["angle"], local["idx"], tol)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    # one packed f64 record per pair: [cost, angle, idx, uniform, tol] (idx/uniform are small ints: exact in f64).
    # The tolerance travels with the record: a rank that staged only its pair block's frames knows the tolerance of its
    # own jobs (0 for the others, mm_within_plan_staged), the owners of a job agree on it -- every rank merges with the
    # largest.
    n = len(local["cost"])
    tol_row = np.zeros(n, dtype=np.float64) if tol is None else np.asarray(tol, dtype=np.float64)
    rec = torch.from_numpy(np.stack([local["cost"], local["angle"], local["idx"].astype(np.float64),
                                     local["uniform"].astype(np.float64), tol_row], axis=0)).to(dev)
    gathered = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(gathered, rec, group=group)
    g = torch.stack(gathered, dim=0).cpu().numpy()     # [world, 5, n]
    tol_all = None if tol is None else g[:, 4, :].max(axis=0)
    return merge_shards(world, g[:, 0, :], g[:, 3, :].astype(np.int32), g[:, 1, :], g[:, 2, :].astype(np.int32), tol_all)


# ------------------------------------------------------------------------------------------
# the exchange on the device
# ------------------------------------------------------------------------------------------
class _ExchangeBuffers:
    """Device buffers of one plan's exchange records (torch owns the memory, the C ABI gets the
    addresses) and the engine's stream as a torch stream, so that the collectives are enqueued
    stream-ordered behind the export kernels."""

    def __init__(self, plan, n_jobs: int):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        self.cost = torch.empty(max(n_jobs, 1), dtype=torch.float64, device=dev)
        self.keys = torch.empty(3 * max(n_jobs, 1), dtype=torch.int64, device=dev)
        self.stream = torch.cuda.ExternalStream(plan.engine.stream, device=dev)


def _buffers(plan, n_jobs):
    """One set of exchange buffers per engine and size (an engine runs one search at a time): consecutive cases on
    an engine reuse them, so nothing is allocated between a step's hand-over and its first launch."""
    b = getattr(plan, "_xbuf", None)
    if b is None:
        cache = plan.engine.__dict__.setdefault("_xbuf_cache", {})
        b = cache.get(n_jobs)
        if b is None:
            b = cache[n_jobs] = _ExchangeBuffers(plan, n_jobs)
        plan._xbuf = b
    return b


def _all_reduce_min(t, group, stream):
    """all_reduce(MIN) of a device tensor, ordered on `stream`.  RCCL reduces in place on the device; with a
    CPU backend (gloo: the tests, several ranks on one GPU) the record takes the detour through the host."""
    import torch
    import torch.distributed as dist
    with torch.cuda.stream(stream):
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        else:
            h = t.cpu()                      # synchronises `stream`: the export kernel has finished
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
            t.copy_(h)


def search_device(plan, group=None):
    """The search half of WithinPlan.run_sharded with the exchange on the device (module docstring)."""
    n_jobs, n_levels, _tol = plan.dims()
    b = None
    begun = bool(getattr(plan, "_begun", False))  # WithinPlan.search_begin has enqueued level 0 already
    plan._begun = False
    for l in range(n_levels):
        if not (begun and l == 0):
            plan.level_launch(l)                  # first thing: the device starts on the level while the host goes on
        if b is None:
            b = _buffers(plan, n_jobs)
        plan.level_export_cost(l, b.cost.data_ptr())
        _all_reduce_min(b.cost, group, b.stream)
        plan.level_export_keys(l, b.cost.data_ptr(), b.keys.data_ptr())
        _all_reduce_min(b.keys, group, b.stream)
        plan.level_commit_dev(l, b.cost.data_ptr(), b.keys.data_ptr())


_checked_groups = set()


def first_search_pending(group=None) -> bool:
    """True until the first sharded search of this process on `group` has been cross-checked (MM_EXCHANGE_CHECK=0: never)."""
    return os.environ.get("MM_EXCHANGE_CHECK", "1") != "0" and (group if group is not None else "default") not in _checked_groups


def search_checked(plan, group=None, mode="rccl"):
    """The FIRST sharded search of a process group runs the chosen on-device exchange (``rccl``: the library's
    communicator, ``device``: torch's all-reduces) AND the gather exchange on every level, and compares them job by job
    -- reduced cost, first index of minimal cost, decided flag -- before it commits: a transport that reduces wrongly
    (in-place MIN on int64 / f64 records, stream ordering) fails here, loudly, not in a later alignment.
    Costs one all_gather per level, once."""
    import torch
    n_jobs, n_levels, tol = plan.dims()
    b = _buffers(plan, n_jobs)
    comm = native_comm(group) if mode == "rccl" else None
    begun = bool(getattr(plan, "_begun", False))
    plan._begun = False
    for l in range(n_levels):
        if not (begun and l == 0):
            plan.level_launch(l)
        local = plan.level_collect(l, n_jobs)
        ok_g, _angle_g, idx_g, cost_g = merge_level(local, tol, group)
        plan.level_export_cost(l, b.cost.data_ptr())
        if comm is not None:
            comm.all_reduce_min_f64(b.cost.data_ptr(), n_jobs, plan.engine.stream)
        else:
            _all_reduce_min(b.cost, group, b.stream)
        plan.level_export_keys(l, b.cost.data_ptr(), b.keys.data_ptr())
        if comm is not None:
            comm.all_reduce_min_i64(b.keys.data_ptr(), 3 * n_jobs, plan.engine.stream)
        else:
            _all_reduce_min(b.keys, group, b.stream)
        plan.engine.synchronize()
        torch.cuda.synchronize()
        cost_d = b.cost[:n_jobs].cpu().numpy()
        keys_d = b.keys[:3 * n_jobs].cpu().numpy()
        act = local["active"] != 0
        fin = act & np.isfinite(cost_g)
        lo, hi = keys_d[n_jobs:2 * n_jobs], ~keys_d[2 * n_jobs:3 * n_jobs]
        if not (np.array_equal(cost_d[act], cost_g[act]) and np.array_equal(keys_d[:n_jobs][fin], idx_g[fin].astype(np.int64))
                and np.array_equal((lo == hi)[fin], ok_g[fin] != 0)):
            raise RuntimeError(f"sharded search: the '{mode}' exchange and the gather exchange disagree on level {l} "
                               f"(first search of this process group; set MM_EXCHANGE=gather to run without the device exchange)")
        plan.level_commit_dev(l, b.cost.data_ptr(), b.keys.data_ptr())
    _checked_groups.add(group if group is not None else "default")


def search_inprocess(plans: Sequence):
    """`world` shard plans of ONE process driven in lockstep with the device exchange, the all-reduces
    replaced by element-wise minima over the plans' device records (tests: the kernels, the key encoding
    and the commit are the ones a multi-rank run uses; only the transport differs)."""
    import torch
    n_jobs, n_levels, _tol = plans[0].dims()
    bufs = [_ExchangeBuffers(p, n_jobs) for p in plans]      # the plans share one engine here: one record each
    for l in range(n_levels):
        for p, b in zip(plans, bufs):
            p.level_launch(l)
            p.level_export_cost(l, b.cost.data_ptr())
        for p in plans:
            p.engine.synchronize()
        g = bufs[0].cost.clone()
        for b in bufs[1:]:
            g = torch.minimum(g, b.cost)
        torch.cuda.synchronize()
        for p, b in zip(plans, bufs):
            b.cost.copy_(g)
        torch.cuda.synchronize()
        for p, b in zip(plans, bufs):
            p.level_export_keys(l, b.cost.data_ptr(), b.keys.data_ptr())
        for p in plans:
            p.engine.synchronize()
        k = bufs[0].keys.clone()
        for b in bufs[1:]:
            k = torch.minimum(k, b.keys)
        torch.cuda.synchronize()
        for p, b in zip(plans, bufs):
            b.keys.copy_(k)
        torch.cuda.synchronize()
        for p, b in zip(plans, bufs):
            p.level_commit_dev(l, b.cost.data_ptr(), b.keys.data_ptr())


# ------------------------------------------------------------------------------------------
# the finish of a sharded alignment, sharded too
# ------------------------------------------------------------------------------------------
# After a sharded search every rank holds every winner.  The chain walks (host work, one pullback independent of the other:
# the reference's four crossbeam threads, entry.rs:140-203) and the between alignments (2 || + 2 ||, entry.rs:206-277) used
# to be repeated on every rank -- at N = 8 that finish (2.8 ms) was longer than the rank's launch (2.4 ms).  Here pullback g
# is walked on rank g mod world alone and pair k of a between batch is aligned on rank k mod world alone; what they changed
# -- logs, coordinates, the rotation -- goes to the other ranks in ONE broadcast per pullback from its owner.
def _mutable_arrays(g):
    """the arrays of a FlatGeometry that a chain walk or a between alignment rewrites"""
    return [a for a in (g.centroids, g.lumen, g.cath, g.extra, g.ref, getattr(g, "lumen_centroids", None)) if a is not None]


def broadcast_state(arrays, src: int, group=None, comm: "N.Comm" = None, stream: int = 0):
    """One broadcast of a list of numpy arrays / ctypes buffers from rank `src`, in place.  comm: the library's
    communicator (``mm_comm_broadcast`` on a device staging buffer); else torch.distributed on `group` (gloo: host tensors)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    views = [np.frombuffer(a, dtype=np.uint8) if not isinstance(a, np.ndarray) else a.reshape(-1).view(np.uint8) for a in arrays]
    sizes = [v.size for v in views]
    rank = comm.rank if comm is not None else dist.get_rank(group)
    flat = np.concatenate(views) if rank == src else np.empty(sum(sizes), dtype=np.uint8)
    if comm is not None:
        t = torch.from_numpy(flat).cuda()
        comm.broadcast(t.data_ptr(), flat.size, src, stream)
        torch.cuda.synchronize()
        flat = t.cpu().numpy()
    else:
        on_dev = dist.get_backend(group) == "nccl"
        t = torch.from_numpy(flat)
        t = t.cuda() if on_dev else t
        dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        flat = t.cpu().numpy() if on_dev else flat
    if rank != src:
        at = 0
        for v, n in zip(views, sizes):
            v[:] = flat[at:at + n]
            at += n


def walk_sharded(plan, group=None, rank: Optional[int] = None, world: Optional[int] = None, exchange: bool = True,
                 comm: "N.Comm" = None):
    """The chain walks of a searched plan, pullback g on rank g mod world, then every pullback's logs and coordinates
    broadcast from its owner: every rank returns what ``plan.walk()`` returns on one GPU.  (rank, world, exchange=False:
    a single process playing one rank of a larger job for timing -- nothing is exchanged, the result is not an alignment.
    comm: the library's communicator for the broadcasts (``mm_comm_broadcast``); default: torch.distributed on `group`.)"""
    import torch
    import torch.distributed as dist
    if world is None:
        world, rank = world_size(group), (dist.get_rank(group) if world_size(group) > 1 else 0)
    G = len(plan.geoms)
    take = [g % world == rank for g in range(G)]
    logs, evals, unresolved = plan.walk(take=take)
    if world > 1 and exchange:
        for g, geom in enumerate(plan.geoms):
            broadcast_state(_mutable_arrays(geom) + [logs[g]._buf], g % world, group, comm, plan.engine.stream)
        t = torch.tensor([evals, unresolved], dtype=torch.int64)
        on_dev = dist.get_backend(group) == "nccl"
        t = t.cuda() if on_dev else t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        evals, unresolved = int(t[0].item()), int(t[1].item())
    return logs, evals, unresolved


def align_between_sharded(engine, pairs, rot_deg: float, step_rot_deg: float, sample_size: int, precision: int, group=None,
                          rank: Optional[int] = None, world: Optional[int] = None, exchange: bool = True, comm: "N.Comm" = None):
    """``geometry.align_between`` with pair k aligned on rank k mod world alone; the moved geometry (b of every pair) and
    its rotation are broadcast from the owner.  Returns (rotations, pose_evals) like align_between, on every rank."""
    import torch
    import torch.distributed as dist
    from .geometry import align_between
    if world is None:
        world, rank = world_size(group), (dist.get_rank(group) if world_size(group) > 1 else 0)
    mine = [k for k in range(len(pairs)) if k % world == rank]
    rot = np.zeros(len(pairs), dtype=np.float64)
    evals = 0
    if mine:
        r, evals = align_between(engine, [pairs[k] for k in mine], rot_deg, step_rot_deg, sample_size, precision)
        rot[mine] = r
    if world > 1 and exchange:
        for k, (_, b) in enumerate(pairs):
            broadcast_state(_mutable_arrays(b) + [rot[k:k + 1]], k % world, group, comm, engine.stream)
        t = torch.tensor([evals], dtype=torch.int64)
        on_dev = dist.get_backend(group) == "nccl"
        t = t.cuda() if on_dev else t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        evals = int(t[0].item())
    return rot, evals
