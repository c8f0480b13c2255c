"""The real multi-rank WithinPlan path on the GPU.

(a) In one process: `world` WithinPlans on copies of the same geometries, each owning its share
    [n*r/world, n*(r+1)/world) of every candidate list (set_shard), driven level by level exactly as
    `world` ranks would be: level_local on every plan -> merge of the per-shard records
    (mm_merge_shards, what every rank computes after the exchange) -> level_commit into every plan ->
    walk.  Logs, coordinates and the number of re-searched steps of EVERY "rank" must equal the
    single-rank plan.run() and the oracle's sequential chain.
(b) The same through torch.distributed: a world = 2 `gloo` child (torch.distributed.run) on the one
    GPU drives plan.run_sharded() end to end (tests/_shard_worker.py).
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import geoms_equal, to_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def drive_sharded(mm, engine, base, world, step, rng_deg, bruteforce, ss, precision, exchange="gather", grid=None,
                  on_tile=False, moved=False):
    """Run `world` shard plans in lockstep; returns per rank (geoms, logs, evals, unresolved).
    grid = (pair_blocks, cand_slices) with pair_blocks * cand_slices == world: tiles of the (frame pair x candidate)
    grid (mm_within_plan_set_shard_grid); None = the pure candidate-axis split (set_shard).
    on_tile: the plans are CREATED on their tiles (mm_within_plan_create_grid), which stages only the frames of the tile's
    pair block; moved: created on the NEXT rank's tile and then moved to their own (the frames are staged again)."""
    from multimoda_rs_amd import distributed as D
    cases = [[g.copy() for g in base] for _ in range(world)]
    if on_tile:
        assert grid is not None and grid[0] * grid[1] == world
        plans = [mm.WithinPlan(engine, cases[r], step, rng_deg, bruteforce, ss, precision=precision,
                               shard=((r + 1) % world if moved else r, grid[0], grid[1])) for r in range(world)]
        full = sum(int(g.lumen_off[-1]) + (int(g.cath_off[-1]) if g.cath_off is not None else 0) for g in base)
        if moved:
            for r, p in enumerate(plans):
                p.set_shard_grid(r, grid[0], grid[1])
        raw = [p.staged()[0] for p in plans]
        if grid[0] > 1:
            assert max(raw) < full                         # nobody staged every frame ...
            assert sum(raw[::grid[1]]) >= full             # ... and the pair blocks cover them (with their halo frames)
        else:
            assert raw == [full] * world
    else:
        plans = [mm.WithinPlan(engine, cases[r], step, rng_deg, bruteforce, ss, precision=precision) for r in range(world)]
        for r, p in enumerate(plans):
            if grid is None:
                p.set_shard(r, world)
            else:
                assert grid[0] * grid[1] == world
                p.set_shard_grid(r, grid[0], grid[1])
    n_jobs, n_levels, tol = plans[0].dims()
    for p in plans[1:]:                        # a job's tolerance is its owners' (0 from a plan that did not stage it)
        tol = np.maximum(tol, p.dims()[2])
    assert (tol > 0).all()
    if exchange == "gather":
        for l in range(n_levels):
            loc = [p.level_local(l, n_jobs) for p in plans]
            stack = lambda k: np.stack([x[k] for x in loc], axis=0)
            for x in loc[1:]:
                assert np.array_equal(x["active"], loc[0]["active"])      # every rank searches the same jobs
            ok, angle, _idx, _cost = D.merge_shards(world, stack("cost"), stack("uniform"), stack("angle"), stack("idx"), tol)
            for p in plans:
                p.level_commit(l, ok, angle)
    else:
        D.search_inprocess(plans)             # device-side exchange, all_reduce(MIN) emulated over the plans
    out = []
    for r, p in enumerate(plans):
        logs, evals, unres = p.walk()
        out.append((cases[r], logs, evals, unres))
        p.close()
    return out


CASES = [
    # bruteforce, step, range, sample_size
    (True, 1.0, 180.0, 501),          # 361 candidates; first and last are both -pi (duplicates on different shards)
    (False, 0.05, 45.0, 200),         # ladder 1 deg -> 0.1 deg -> 0.05 deg: per-job lists from level 1 on
    (True, 45.0, 90.0, 64),           # 5 candidates: with world = 8 some shards own nothing
]


@pytest.mark.parametrize("exchange", ["gather", "device"])
@pytest.mark.parametrize("precision", [2, 3, 4])
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", CASES)
def test_sharded_within_plan_equals_single_rank_and_oracle(engine, oracle, mm, bruteforce, step, rng_deg, ss, world,
                                                           precision, exchange):
    engine.set_bound_min_candidates(0)        # bound rounds on every batch (shards prune against their own best)
    try:
        base = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((9, 6, 9, 7))]
        single = [g.copy() for g in base]
        wp = mm.WithinPlan(engine, single, step, rng_deg, bruteforce, ss, precision=precision)
        slogs, sevals, sunres = wp.run()
        wp.close()
        ogeoms = [to_oracle(oracle, g) for g in base]
        ologs = [oracle.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=8) for o in ogeoms]
        for geoms, logs, evals, unres in drive_sharded(mm, engine, base, world, step, rng_deg, bruteforce, ss, precision,
                                                       exchange):
            assert evals == sevals and unres == sunres
            for k in range(len(base)):
                assert logs[k] == slogs[k] == ologs[k]
                assert geoms_equal(geoms[k], ogeoms[k]) and geoms_equal(single[k], ogeoms[k])
    finally:
        engine.set_bound_min_candidates(16384)


@pytest.mark.parametrize("exchange", ["gather", "device"])
@pytest.mark.parametrize("precision", [2, 3, 4])
@pytest.mark.parametrize("grid", [(2, 1), (3, 1), (2, 2), (2, 4), (4, 2), (8, 1), (27, 1), (32, 1)])
@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", CASES)
def test_grid_sharded_within_plan_equals_single_rank_and_oracle(engine, oracle, mm, bruteforce, step, rng_deg, ss, grid,
                                                                precision, exchange):
    """Tiles of the (frame pair x candidate) grid (include/mm_hausdorff.h, "multi-GPU"): pair blocks x candidate slices.
    27 frame pairs here: (27, 1) gives every rank one pair, (32, 1) leaves five ranks without any."""
    engine.set_bound_min_candidates(0)
    try:
        world = grid[0] * grid[1]
        base = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((9, 6, 9, 7))]
        ogeoms = [to_oracle(oracle, g) for g in base]
        ologs = [oracle.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=8) for o in ogeoms]
        evals0 = None
        for geoms, logs, evals, unres in drive_sharded(mm, engine, base, world, step, rng_deg, bruteforce, ss, precision,
                                                       exchange, grid=grid):
            evals0 = evals if evals0 is None else evals0
            assert evals == evals0 and unres == 0
            for k in range(len(base)):
                assert logs[k] == ologs[k]
                assert geoms_equal(geoms[k], ogeoms[k])
    finally:
        engine.set_bound_min_candidates(16384)


@pytest.mark.parametrize("exchange", ["gather", "device"])
@pytest.mark.parametrize("precision", [2, 4])
@pytest.mark.parametrize("grid,moved", [((1, 2), False), ((2, 1), False), ((2, 2), False), ((4, 2), False), ((8, 1), False),
                                        ((27, 1), False), ((32, 1), False), ((3, 2), True), ((27, 1), True)])
@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", CASES[:2])
def test_plans_created_on_their_tile_stage_their_frames_only(engine, oracle, mm, bruteforce, step, rng_deg, ss, grid, moved,
                                                             precision, exchange):
    """mm_within_plan_create_grid with several pair blocks: each plan copies and builds only the frames its pair block
    reads (its jobs' frames i-1 and i), the other sets are empty -- same logs and coordinates as the oracle on every
    rank.  The pullbacks sit away from the origin, so the tolerance (derived from the coordinates' magnitude) of a plan
    that has not seen frame 0 matters."""
    engine.set_bound_min_candidates(0)
    try:
        world = grid[0] * grid[1]
        base = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((9, 6, 9, 7))]
        for k, g in enumerate(base):
            shift = np.array([37.5 * (k + 1), -12.25 * (k + 2), 0.0])
            g.lumen[:] += shift; g.centroids[:] += shift
            if g.cath is not None:
                g.cath[:] += shift
            if g.ref is not None:
                g.ref[:] += shift
        ogeoms = [to_oracle(oracle, g) for g in base]
        ologs = [oracle.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=8) for o in ogeoms]
        for geoms, logs, evals, unres in drive_sharded(mm, engine, base, world, step, rng_deg, bruteforce, ss, precision,
                                                       exchange, grid=grid, on_tile=True, moved=moved):
            assert unres == 0
            for k in range(len(base)):
                assert logs[k] == ologs[k]
                assert geoms_equal(geoms[k], ogeoms[k])
    finally:
        engine.set_bound_min_candidates(16384)


def test_default_shard_grid():
    from multimoda_rs_amd import _native as N
    assert N.shard_grid(8, 2044) == (8, 1)       # config3: frame pairs only
    assert N.shard_grid(8, 508) == (4, 2)        # config2
    assert N.shard_grid(8, 80) == (1, 8)         # a few pairs with long candidate lists: the candidate axis
    assert N.shard_grid(1, 5) == (1, 1) and N.shard_grid(6, 1000) == (6, 1) and N.shard_grid(6, 200) == (3, 2)


@pytest.mark.parametrize("precision", [2, 3, 4])
@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", CASES)
def test_native_rccl_search_world1(engine, oracle, mm, bruteforce, step, rng_deg, ss, precision):
    """mm_within_plan_run_sharded's search half on a world = 1 RCCL communicator owned by the library (mm_comm_*):
    the launch path a multi-GPU job takes -- export kernels, ncclAllReduce(MIN) x 2 on the engine's stream, commit --
    without a peer.  Same logs and coordinates as the oracle."""
    comm = mm.Comm(mm.Comm.unique_id(), 0, 1)
    try:
        assert comm.rank == 0 and comm.world == 1
        base = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((9, 6, 9, 7))]
        ogeoms = [to_oracle(oracle, g) for g in base]
        ologs = [oracle.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=8) for o in ogeoms]
        plan = mm.WithinPlan(engine, base, step, rng_deg, bruteforce, ss, precision=precision)
        plan.search_sharded(comm)
        logs, _evals, unres = plan.walk()
        plan.close()
        assert unres == 0
        for k in range(len(base)):
            assert logs[k] == ologs[k] and geoms_equal(base[k], ogeoms[k])
        # a plan made for another world is refused
        plan = mm.WithinPlan(engine, [g.copy() for g in base], step, rng_deg, bruteforce, ss, precision=precision, shard=(0, 2))
        with pytest.raises(RuntimeError, match="not the communicator"):
            plan.search_sharded(comm)
        plan.close()
    finally:
        comm.close()


@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", CASES)
def test_chained_cases_over_two_engines_native_comm(oracle, mm, bruteforce, step, rng_deg, ss):
    """The driver's back-to-back flow over the library's communicator (bench.py at N > 1): case k's whole exchange is
    enqueued (search_sharded_begin), case k+1's launch is ordered behind it on another engine (Engine.wait_exchange +
    level_launch) BEFORE case k is collected.  Five cases over two engines; every case ends with the oracle's result."""
    comm = mm.Comm(mm.Comm.unique_id(), 0, 1)
    engs = [mm.Engine(), mm.Engine()]
    try:
        cases = [[mm.synthetic_pullback(f, 501, pullback_id=i, seed=k) for i, f in enumerate((9, 6, 7))] for k in range(5)]
        want = []
        for c in cases:
            og = [to_oracle(oracle, g) for g in c]
            want.append(([oracle.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=8) for o in og], og))
        plans = [None] * len(cases)
        mk = lambda k: mm.WithinPlan(engs[k % 2], cases[k], step, rng_deg, bruteforce, ss, precision=mm.MM_PRECISION_F32_FAST)
        plans[0] = mk(0)
        plans[0].level_launch(0)
        for k in range(len(cases)):
            plans[k].search_sharded_begin(comm)
            if k + 1 < len(cases):
                plans[k + 1] = mk(k + 1)
                engs[(k + 1) % 2].wait_exchange(engs[k % 2])
                plans[k + 1].level_launch(0)
            plans[k].search_sharded(comm)
            logs, _ev, unres = plans[k].walk()
            plans[k].close()
            assert unres == 0
            for g, lg, ol, o in zip(cases[k], logs, want[k][0], want[k][1]):
                assert lg == ol and geoms_equal(g, o), k
    finally:
        for e in engs:
            e.close()
        comm.close()


@pytest.mark.parametrize("exchange", ["gather", "device"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cross_shard_ties_fall_back_to_chain_state(engine, oracle, mm, world, exchange):
    """Perfect circles: every candidate of every shard ties up to rounding, so no shard is uniform, the merge
    cannot decide (ok == 0) and every step is searched again on the chain state during the walk."""
    t = np.arange(120) * (2 * math.pi / 120)
    lum = [np.stack([4.5 + 0.01 * k + 2 * np.cos(t), 4.4 + 2 * np.sin(t), np.full_like(t, 0.5 * k)], 1) for k in range(5)]
    g = mm.FlatGeometry.from_frames(lum, ref_points={0: (6.5, 4.4, 0.0)})
    og = to_oracle(oracle, g)
    ol = oracle.align_within_chain(og, 2.0, 60.0, True, 120)
    for geoms, logs, _evals, unres in drive_sharded(mm, engine, [g], world, 2.0, 60.0, True, 120, 2, exchange):
        assert logs[0] == ol and geoms_equal(geoms[0], og)
        assert unres == 4


@pytest.mark.parametrize("exchange", ["gather", "device"])
def test_sharded_near_tie_across_two_shards(engine, oracle, mm, exchange):
    """A two-fold symmetric frame pair: the candidates theta and theta + pi cost the same up to rounding and
    lie on different shards (world = 2 splits +-180 deg in the middle).  Each shard alone is uniform; only
    the merge sees two different angles within the tolerance -> undecided -> re-searched on the chain
    state, with the oracle's result."""
    t = np.arange(200) * (2 * math.pi / 200)
    def ell(k, rot):
        x, y = 2.4 * np.cos(t), 1.5 * np.sin(t)                          # centrally symmetric: R(pi) maps it onto itself
        c, s = math.cos(rot), math.sin(rot)
        return np.stack([4.5 + c * x - s * y, 4.5 + s * x + c * y, np.full_like(t, 0.5 * k)], 1)
    g = mm.FlatGeometry.from_frames([ell(0, 0.0), ell(1, math.radians(30.0)), ell(2, math.radians(50.0))],
                                    ref_points={0: (6.9, 4.5, 0.0)})
    og = to_oracle(oracle, g)
    ol = oracle.align_within_chain(og, 1.0, 180.0, True, 200)
    single = g.copy()
    wp = mm.WithinPlan(engine, [single], 1.0, 180.0, True, 200, precision=2)
    slogs, _e, sunres = wp.run()
    wp.close()
    assert slogs[0] == ol and geoms_equal(single, og)
    for geoms, logs, _evals, unres in drive_sharded(mm, engine, [g], 2, 1.0, 180.0, True, 200, 2, exchange):
        assert logs[0] == ol and geoms_equal(geoms[0], og)
        assert unres == sunres


@pytest.mark.parametrize("exchange", ["gather", "device"])
def test_run_sharded_world2_gloo_on_the_gpu(exchange):
    """plan.run_sharded() end to end over torch.distributed (`gloo`, two ranks sharing the one GPU)."""
    import __graft_entry__ as ge
    ge.build()
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "4"
    env["MM_EXCHANGE"] = exchange
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + (os.getpid() % 200)),
           os.path.join(ROOT, "tests", "_shard_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SHARD_WORKER_OK" in r.stdout


def test_sharded_finish_world2_gloo():
    """the FINISH of a sharded alignment sharded too -- pullback g walked on rank g mod world, between pair k
    aligned on rank k mod world, what they changed broadcast from the owner (distributed.walk_sharded /
    align_between_sharded over mm_within_plan_walk_geoms): two `gloo` ranks on the one GPU, every rank against the oracle's
    whole 4-phase alignment (logs, rotations, every coordinate)."""
    import __graft_entry__ as ge
    ge.build()
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", MM_EXCHANGE="device")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29300 + (os.getpid() % 200)),
           os.path.join(ROOT, "tests", "_finish_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "FINISH_WORKER_OK" in r.stdout


def test_walk_of_some_pullbacks_leaves_the_others_untouched(engine, oracle, mm):
    """mm_within_plan_walk_geoms: the taken pullbacks come out as the oracle's chain, the others exactly as they went in."""
    geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((9, 6, 11))]
    before = [g.copy() for g in geoms]
    og = [to_oracle(oracle, g) for g in geoms]
    plan = mm.WithinPlan(engine, geoms, 1.0, 180.0, True, 501, precision=mm.MM_PRECISION_F32_MATRIX)
    plan.search()
    logs, evals, unres = plan.walk(take=[True, False, True])
    plan.close()
    assert unres == 0 and evals == (8 + 10) * 361
    for k in (0, 2):
        assert logs[k] == oracle.align_within_chain(og[k], 1.0, 180.0, True, 501, n_threads=4) and geoms_equal(geoms[k], og[k])
    assert np.array_equal(geoms[1].lumen, before[1].lumen) and np.array_equal(geoms[1].centroids, before[1].centroids)


# ---------------------------------------------------------------------------------------
# search sets built on the device (k_build_sets) vs the host-side helper (mm_catheter_lumen_vec)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("sample_size", [7, 64, 200, 501, 600])
def test_device_built_search_sets_equal_host_construction(engine, mm, sample_size):
    """Ragged contours (33..501 points per frame), with and without catheter: every set staged in HBM must hold
    exactly downsample(lumen, S) ++ downsample(catheter, ceil(n_cath * S / len_lumen0)) minus the frame centroid
    (align_within.rs:45-59,173-191, contour.rs:47-58), its f32 copy the once-rounded value, rho the largest
    distance from the centre."""
    rng = np.random.default_rng(5)
    lens = [501, 33, 200, 64, 500, 77]
    def contour(n, k):
        t = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = 2.0 + 0.3 * np.cos(3 * t + k)
        return np.stack([4.4 + 0.05 * k + r * np.cos(t), 4.6 + 0.8 * r * np.sin(t), np.full(n, 0.5 * k)], 1)
    lum = [contour(n, k) for k, n in enumerate(lens)]
    from multimoda_rs_amd.geometry import catheter_points
    cath = [catheter_points(0.5 * k) for k in range(len(lens))]
    g1 = mm.FlatGeometry.from_frames(lum, catheters=cath, ref_points={0: lum[0][0]})
    g2 = mm.FlatGeometry.from_frames(lum[::-1], ref_points={0: lum[-1][0]})          # no catheter
    plan = mm.WithinPlan(engine, [g1, g2], 5.0, 30.0, True, sample_size)
    s = 0
    for g in (g1, g2):
        for i in range(g.n_frames):
            want = mm.search_set(g, i, sample_size) - g.centroids[i, :2]
            got64, got32, rho = plan.fetch_set(s)
            assert got64.shape == want.shape
            assert np.array_equal(got64, want)
            assert np.array_equal(got32, want.astype(np.float32))
            r = float(np.sqrt(np.max(want[:, 0] * want[:, 0] + want[:, 1] * want[:, 1])))
            assert r <= rho <= r * (1 + 1e-11)
            s += 1
    plan.close()
