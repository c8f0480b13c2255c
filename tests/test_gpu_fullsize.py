"""BASELINE-sized parity inside the driver-run GPU suite.

The whole 4-phase alignment (entry.rs:140-277: four within-pullback chains, then AB | CD and
AC | BD between-alignments) of the bench workloads

  config2   4 pullbacks x 128 frames x 501 pts, 1 deg   x +-180 deg  (361 candidates / search)
  config3   4 pullbacks x 512 frames x 501 pts, 0.5 deg x +-180 deg  (721 candidates / search)

through the product's decoupled path (one launch over all frame pairs, host chain walk) in the
FAST and BOUNDED precisions with the engine's shipped settings, compared bit for bit -- logs,
between rotations, every output coordinate -- with the CPU oracle's sequential chain
(align_within.rs:24-134) and its align_between (align_between.rs:11-68).  The oracle runs once
per workload (about 4 s / 35 s on the GPU box's 16 host threads).  This is what
``bench.py --check`` does; here it runs on every `pytest -m gpu`.

Also one slice of the rotation x frame-shift EXTENSION grid at full length (512 frames, shifts
-2..2) against oracle searches.
"""
import os

import numpy as np
import pytest

from helpers import geoms_equal, to_oracle

pytestmark = pytest.mark.gpu

CONFIGS = {
    "config2": dict(frames=128, points=501, step_deg=1.0, range_deg=180.0, sample_size=501),
    "config3": dict(frames=512, points=501, step_deg=0.5, range_deg=180.0, sample_size=501),
}
BETWEEN_PAIRS = ((0, 1), (2, 3), (0, 2), (1, 3))      # AB | CD, then AC | BD (entry.rs:206-277)


def _threads():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        return 8


@pytest.fixture(scope="module")
def fresh_engine(mm):
    """An engine with the shipped settings (test_gpu_parity.py switches its shared engine's bound
    threshold to 0; the full-size cases must run what a caller gets by default)."""
    import __graft_entry__ as ge
    ge.build()
    eng = mm.Engine()
    yield eng
    eng.close()


_oracle_cache = {}


def _oracle_alignment(oracle, mm, name):
    """(logs per pullback, between rotations, final oracle geometries) of one workload."""
    if name not in _oracle_cache:
        cfg = CONFIGS[name]
        base = mm.synthetic_case(cfg["frames"], cfg["points"])
        og = [to_oracle(oracle, g) for g in base]
        th = _threads()
        logs = [oracle.align_within_chain(o, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"], n_threads=th)
                for o in og]
        rot = [oracle.align_between(og[i], og[j], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], n_threads=th)
               for i, j in BETWEEN_PAIRS]
        _oracle_cache[name] = (logs, rot, og)
    return _oracle_cache[name]


@pytest.mark.parametrize("precision", ["fast", "bounded", "matrix"])
@pytest.mark.parametrize("name", ["config2", "config3"])
def test_full_four_phase_alignment_equals_oracle(fresh_engine, oracle, mm, name, precision):
    cfg = CONFIGS[name]
    prec = {"fast": mm.MM_PRECISION_F32_FAST, "bounded": mm.MM_PRECISION_F32_BOUNDED, "matrix": mm.MM_PRECISION_F32_MATRIX}[precision]
    ologs, orot, ogeoms = _oracle_alignment(oracle, mm, name)
    geoms = mm.synthetic_case(cfg["frames"], cfg["points"])
    plan = mm.WithinPlan(fresh_engine, geoms, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"], precision=prec)
    logs, evals, unresolved = plan.run()
    plan.close()
    n_ang = len(mm.search_angles(cfg["step_deg"], cfg["range_deg"])[0])
    assert evals == 4 * (cfg["frames"] - 1) * n_ang
    assert unresolved == 0                      # generic data: every step decided by the one-shot search
    for k in range(4):
        assert logs[k] == ologs[k], f"within logs of pullback {k} differ"
    a, b, c, d = geoms
    r1, _ = mm.align_between(fresh_engine, [(a, b), (c, d)], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], prec)
    r2, _ = mm.align_between(fresh_engine, [(a, c), (b, d)], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], prec)
    assert list(r1) + list(r2) == orot
    for k, (g, og) in enumerate(zip(geoms, ogeoms)):
        assert geoms_equal(g, og), f"coordinates of pullback {k} differ"


@pytest.mark.parametrize("precision", ["fast", "matrix"])
@pytest.mark.parametrize("grid", [(1, 2), (2, 1), (1, 8), (8, 1), (2, 4)])
def test_config4_sharded_full_size_equals_oracle(fresh_engine, oracle, mm, grid, precision):
    """BASELINE config 4 at its workload: the config3 alignment (4 x 512 frames, 721 candidates) with the
    (frame pair x candidate) grid sharded over 2 and 8 ranks -- the pure candidate-axis split (1, W), the pure pair
    split (W, 1) and a mixed tile (2, 4) -- every "rank" a shard plan of this process, driven level by level with the
    device exchange (export kernels, key encoding and commit are the ones a multi-rank run uses; the all-reduces are
    element-wise minima over the plans' device records).  EVERY rank must end with the oracle's logs and coordinates."""
    from multimoda_rs_amd import distributed as D
    cfg = CONFIGS["config3"]
    ologs, _orot, ogeoms = _oracle_alignment(oracle, mm, "config3")
    # the oracle's geometries have been through the between stage as well: compare with the within result
    world = grid[0] * grid[1]
    cases = [mm.synthetic_case(cfg["frames"], cfg["points"]) for _ in range(world)]
    prec = {"fast": mm.MM_PRECISION_F32_FAST, "matrix": mm.MM_PRECISION_F32_MATRIX}[precision]
    plans = [mm.WithinPlan(fresh_engine, cases[r], cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"],
                           precision=prec, shard=(r, grid[0], grid[1])) for r in range(world)]
    D.search_inprocess(plans)
    single = mm.synthetic_case(cfg["frames"], cfg["points"])
    wp = mm.WithinPlan(fresh_engine, single, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"],
                       precision=prec)
    slogs, sevals, sunres = wp.run()
    wp.close()
    for r, p in enumerate(plans):
        logs, evals, unres = p.walk()
        p.close()
        assert evals == sevals == 4 * 511 * 721 and unres == sunres == 0
        for k in range(4):
            assert logs[k] == ologs[k] == slogs[k], f"rank {r}: within logs of pullback {k} differ"
            assert geoms_equal(cases[r][k], single[k]), f"rank {r}: coordinates of pullback {k} differ"


@pytest.mark.parametrize("precision", ["fast", "matrix"])
@pytest.mark.parametrize("step,rng_deg", [(0.5, 90.0), (0.05, 90.0)])
def test_default_ladder_full_size_equals_oracle(fresh_engine, oracle, mm, step, rng_deg, precision):
    """The reference's DEFAULT mode (bruteforce=False, align_within.rs:193-247) at the config3 shape: 4 x 512 frames,
    0.5 deg x +-90 deg (two dependent levels, 202 evaluations per frame pair) and 0.05 deg (three levels, 303) -- the
    decoupled path's logs, evaluation count and coordinates against the oracle's sequential ladder chain."""
    geoms = mm.synthetic_case(512, 501)
    og = [to_oracle(oracle, g) for g in geoms]
    th = _threads()
    ologs = [oracle.align_within_chain(o, step, rng_deg, False, 501, n_threads=th) for o in og]
    plan = mm.WithinPlan(fresh_engine, geoms, step, rng_deg, False, 501,
                         precision={"fast": mm.MM_PRECISION_F32_FAST, "matrix": mm.MM_PRECISION_F32_MATRIX}[precision])
    logs, evals, unresolved = plan.run()
    plan.close()
    # evaluations per search: 202 / 303 unless a finer level's window is clipped at +-range (limes);
    # oracle.count_evals counts the fully clipped case (its dummy cost makes the first candidate, -range, win)
    full = {0.5: 202, 0.05: 303}[step]
    assert 4 * 511 * oracle.count_evals(step, rng_deg, False) <= evals <= 4 * 511 * full
    assert evals > 4 * 511 * (full - 1)               # generic data: hardly any window touches the limit
    for k in range(4):
        assert logs[k] == ologs[k], f"ladder logs of pullback {k} differ"
        assert geoms_equal(geoms[k], og[k]), f"coordinates of pullback {k} differ"


@pytest.mark.parametrize("bruteforce", [True, False])
@pytest.mark.parametrize("precision", ["matrix", "bounded"])
def test_oct_single_benchmark_shape_equals_oracle(fresh_engine, oracle, mm, bruteforce, precision):
    """The reference's SECOND published benchmark at full size (benchmarks/benchmark_cpu_scaling.py:32-80): 280 frames,
    0.01 deg x +-6 deg (1201 candidates brute force; 1 deg / 0.1 deg / 0.01 deg ladder = 135 evaluations otherwise),
    sample_size 200, n_points 40 -> 223 points per set (7 x 7 tiles of the matrix-pipe screen): the whole chain's logs and
    coordinates against the oracle's sequential chain, and `mm.from_array_single` end to end on the same arrays."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    O = bench.OCT
    g = bench.oct_pullback(mm)
    og = to_oracle(oracle, g)
    ologs = oracle.align_within_chain(og, O["step_deg"], O["range_deg"], bruteforce, O["sample_size"], n_threads=_threads())
    prec = {"matrix": mm.MM_PRECISION_F32_MATRIX, "bounded": mm.MM_PRECISION_F32_BOUNDED}[precision]
    before = fresh_engine.screen_stats()
    plan = mm.WithinPlan(fresh_engine, [g], O["step_deg"], O["range_deg"], bruteforce, O["sample_size"], precision=prec)
    logs, evals, unresolved = plan.run()
    plan.close()
    after = fresh_engine.screen_stats()
    assert logs[0] == ologs and geoms_equal(g, og)
    if bruteforce:
        assert evals == 279 * 1201
    else:
        assert 279 * oracle.count_evals(O["step_deg"], O["range_deg"], False) <= evals <= 279 * 135
    assert after["packed_fma"] == before["packed_fma"] and after["direct_f32"] == before["direct_f32"]
    if precision == "matrix":
        assert after["matrix"] - before["matrix"] == evals          # every candidate screened on the matrix pipe
    elif not bruteforce:
        # the bounded search: the ladder's first and last level (279 x 13 and 279 x 21 candidates) are small batches, screened
        # outright on the matrix pipe; the middle level (279 x 101) runs the bound rounds
        assert after["matrix"] - before["matrix"] >= 279 * 13
    # the entry point itself: same logs
    g_api, logs_api = mm.from_array_single(bench.oct_input_data(mm, bench.oct_pullback(mm)), step_rotation_deg=O["step_deg"],
                                           range_rotation_deg=O["range_deg"], sample_size=O["sample_size"],
                                           image_center=O["image_center"], n_points=O["n_points"], write_obj=False, smooth=False,
                                           bruteforce=bruteforce, engine=fresh_engine)
    assert list(logs_api) == ologs


def test_bench_two_ranks_check_config2():
    """`bench.py --gpus 2 --workload config2 --check` as the driver starts it (torch.distributed.run, here two `gloo`
    ranks sharing the one GPU): the sharded step pipeline end to end, rank 0's result compared with the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MM_BENCH_BACKEND="gloo", OMP_NUM_THREADS="8")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "config2", "--steps", "3",
           "--warmup", "1", "--check", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["config"]["all_timed_steps_identical"]
    assert out["check"]["within_logs_identical"] and out["check"]["between_rotations_identical"]
    assert out["check"]["all_coordinates_identical"]


def test_full_size_chain_mode_equals_oracle_config2(fresh_engine, oracle, mm):
    """The faithful per-step chain (mode 0) at config2 size: 127 dependent batched searches."""
    cfg = CONFIGS["config2"]
    ologs, _orot, _og = _oracle_alignment(oracle, mm, "config2")
    geoms = mm.synthetic_case(cfg["frames"], cfg["points"])
    logs, evals = mm.align_within(fresh_engine, geoms, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"],
                                  precision=mm.MM_PRECISION_F32_FAST, mode=0)
    for k in range(4):
        assert logs[k] == ologs[k]
    assert evals == 4 * 127 * 361


@pytest.mark.parametrize("precision", ["fast", "bounded"])
def test_extension_grid_full_length_slice_vs_oracle(fresh_engine, oracle, mm, precision):
    """EXTENSION axis (absent from the reference's 4-phase path): one 512-frame pullback, shifts -2..2,
    721 rotations = 2 041 searches.  Every 8th search is re-done by the oracle (brute force over the same
    candidate list on the same centred sets); winners and exact costs must agree bit for bit, and the
    bounded search must agree with the full screen everywhere."""
    prec = {"fast": mm.MM_PRECISION_F32_FAST, "bounded": mm.MM_PRECISION_F32_BOUNDED}[precision]
    g = mm.synthetic_pullback(512, 501, pullback_id=2)
    srs = mm.ShiftRotationSearch(fresh_engine, [g], -2, 2, 0.5, 180.0, 501, precision=prec)
    res = srs.run()
    angles = srs.angles
    assert len(angles) == 721 and len(srs.meta) == 4 * 512 - 7   # refs i-3, i-2, i-1, i+1 inside the pullback
    sets = {}

    def centred(i):
        if i not in sets:
            sets[i] = mm.search_set(g, int(i), 501) - g.centroids[i, :2]
        return sets[i]

    th = _threads()
    for p in range(0, len(srs.meta), 8):
        _gi, i, _sh, j = srs.meta[p]
        # the oracle's search_range over the same list (process_utils.rs:33-75), candidates in parallel
        best = oracle.bruteforce_rotation(centred(j), centred(i), 0.5, 180.0, 0.0, 0.0, n_threads=th)
        assert res["best_angle"][p] == best and angles[res["best_idx"][p]] == best
        assert res["best_cost"][p] == oracle.cost_within(centred(j), centred(i), best, 0.0, 0.0)
    srs.close()
    key = "ext_slice_result"
    if key in _oracle_cache:                     # second precision: identical to the first everywhere
        prev = _oracle_cache[key]
        assert np.array_equal(prev["best_idx"], res["best_idx"]) and np.array_equal(prev["best_cost"], res["best_cost"])
        assert prev["winners"] == res["winners"]
    _oracle_cache[key] = res
