"""Randomised parity of the search primitive on the device (hypothesis, derandomised so that the GPU box runs the
same examples every time): one brute-force rotation search through the C ABI at every precision against the oracle's
threaded `bruteforce_rotation` (winner angle) and `cost_within` (its f64 cost), over shape families that stress the
tie rules -- circles (every candidate ties), two-fold symmetric shapes (exact ties half a turn apart), duplicated and
collinear points, single points, point sets far smaller than their coordinates (exact costs round to ties the
screen does not see) -- and set sizes around the kernels' row-block / LDS-tile boundaries."""
import math
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 15, 16, 17, 33, 63, 64, 65, 144, 160, 208, 224, 225, 287, 288, 289, 320, 352, 500, 520, 521, 528]


def shape(rng, kind, n):
    t = np.sort(rng.uniform(0, 2 * math.pi, n)) if kind != "regular" else np.arange(n) * (2 * math.pi / max(n, 1))
    if kind == "blob":
        r = 2.5 * (1 + 0.15 * np.cos(2 * t + rng.uniform(0, 6)) + 0.07 * np.sin(3 * t)) + rng.normal(0, 0.05, n)
        return np.stack([4.5 + r * np.cos(t), 4.5 + 0.8 * r * np.sin(t)], 1)
    if kind in ("circle", "regular"):
        return np.stack([4.5 + 2.0 * np.cos(t), 4.5 + 2.0 * np.sin(t)], 1)
    if kind == "ellipse2":                       # symmetric under a half turn: exact ties pi apart
        h = np.arange((n + 1) // 2) * (2 * math.pi / max(n, 1))
        p = np.stack([3.0 * np.cos(h), 1.2 * np.sin(h)], 1)
        return (np.concatenate([p, -p])[:n]) + 4.5
    if kind == "dups":                           # few distinct points, many repeats
        base = rng.normal(4.5, 1.0, size=(max(1, min(5, n)), 2))
        return base[rng.integers(0, base.shape[0], n)]
    if kind == "line":
        s = rng.uniform(-2, 2, n)
        return np.stack([4.5 + s, 4.5 + 0.5 * s], 1)
    if kind == "point":                          # every point the same: all exact costs are 0.0, the first candidate wins
        return np.tile(rng.normal(4.5, 1.0, size=(1, 2)), (n, 1))
    if kind == "holes":                          # a contour with a few NaN / inf coordinates: the metric skips those points
        p = shape(rng, "blob", n)
        bad = rng.integers(0, n, size=max(1, n // 50))
        p[bad, rng.integers(0, 2, size=bad.size)] = rng.choice([np.nan, np.inf, -np.inf], size=bad.size)
        return p
    if kind == "speck":                          # a cloud of a few ulps / of 1e-9 around a far point
        return rng.normal(4.5, 1.0, size=(1, 2)) + rng.normal(0, float(rng.choice([4e-16, 1e-12, 1e-9])), size=(n, 2))
    raise AssertionError(kind)


@settings(max_examples=150 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), na=st.sampled_from(SIZES), nb=st.sampled_from(SIZES),
       kind_a=st.sampled_from(["blob", "circle", "regular", "ellipse2", "dups", "line", "point", "speck", "holes"]),
       same=st.booleans(), step=st.sampled_from([0.5, 1.0, 2.5, 7.0]), rng_deg=st.sampled_from([3.0, 45.0, 90.0, 180.0]),
       twist=st.sampled_from([0.0, 7.3, 90.0, 180.0, -33.0]))
def test_one_search_all_precisions_match_the_oracle(engine, oracle, mm, seed, na, nb, kind_a, same, step, rng_deg, twist):
    rng = np.random.default_rng(seed)
    ref = shape(rng, kind_a, na)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = np.nanmean(np.where(np.isfinite(ref), ref, np.nan), axis=0) if kind_a == "holes" else ref.mean(axis=0)
    if not np.all(np.isfinite(c)):
        c = np.array([4.5, 4.5])
    if same and na == nb:                         # the target is the reference turned by `twist` (exactly at 0 / 180)
        th = math.radians(twist)
        rot = np.array([[math.cos(th), math.sin(th)], [-math.sin(th), math.cos(th)]])
        with np.errstate(all="ignore"):
            tgt = ref.copy() if twist == 0.0 else (2 * c - ref if twist == 180.0 else (ref - c) @ rot + c)
    else:
        tgt = shape(rng, str(rng.choice(["blob", "circle", "dups"])), nb) if kind_a not in ("point", "speck") else \
            ref[rng.integers(0, na, nb)] + rng.normal(0, float(rng.choice([0.0, 4e-16, 1e-10])), size=(nb, 2))
    centre = (float(c[0]), float(c[1]))
    angles, degenerate, _ = mm.search_angles(step, rng_deg)
    assert not degenerate
    o_angle = oracle.bruteforce_rotation(ref, tgt, step, rng_deg, centre[0], centre[1], n_threads=8)
    o_cost = oracle.cost_within(ref, tgt, o_angle, centre[0], centre[1])
    # every candidate in exact f64: the winner is the FIRST index of minimal cost (process_utils.rs:72)
    bi64, ba, bc, costs = engine.best_rotation(ref, tgt, angles, centre, skip_zero=True, precision=mm.MM_PRECISION_F64,
                                               return_costs=True)
    assert bi64 == int(np.argmin(costs)) and ba == o_angle and bc == o_cost == costs[bi64], (bi64, ba, o_angle, bc, o_cost)
    for prec in (mm.MM_PRECISION_F32, mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED, mm.MM_PRECISION_F32_MATRIX):
        bi, ba, bc = engine.best_rotation(ref, tgt, angles, centre, skip_zero=True, precision=prec)
        assert bi == bi64 and ba == o_angle and bc == o_cost, (prec, bi, bi64, ba, o_angle, bc, o_cost)


@settings(max_examples=40 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), n_frames=st.integers(2, 9), n_points=st.sampled_from([6, 20, 64, 120, 200, 501]),
       ss=st.sampled_from([5, 64, 200, 500, 501]), bruteforce=st.booleans(),
       step=st.sampled_from([0.005, 0.05, 0.5, 1.0, 3.0]), rng_deg=st.sampled_from([6.0, 20.0, 45.0, 90.0, 180.0]),
       prec=st.sampled_from([0, 1, 2, 3]), mode=st.sampled_from([0, 1]), sigma=st.sampled_from([0.5, 3.0, 25.0]),
       circular=st.booleans())
def test_chain_matches_the_oracle_chain(engine, oracle, mm, seed, n_frames, n_points, ss, bruteforce, step, rng_deg, prec, mode,
                                        sigma, circular):
    """align_frames_in_geometry lines 24-134: logs and every coordinate of the faithful chain (mode 0) and of the
    decoupled plan (mode 1) equal the oracle's sequential chain, over random pullback sizes, sample sizes, grids,
    ladders, precisions, torsion, and near-circular frames (flat cost curves: many near-ties)."""
    from helpers import geoms_equal, to_oracle
    if bruteforce and step < 0.5:
        step = 0.5                                    # keep a brute-force grid of at most 721 candidates
    g = mm.synthetic_pullback(n_frames, n_points, pullback_id=seed % 4, seed=seed % 1000, torsion_sigma_deg=sigma)
    if circular:
        c = g.centroids[np.repeat(np.arange(g.n_frames), np.diff(g.lumen_off)), :2]
        d = g.lumen[:, :2] - c
        r = np.hypot(d[:, 0], d[:, 1])[:, None]
        g.lumen[:, :2] = c + d * (0.15 + 0.85 * (2.0 / r))
    og = to_oracle(oracle, g)
    logs, _ = mm.align_within(engine, [g], step, rng_deg, bruteforce, ss, precision=prec, mode=mode)
    ol = oracle.align_within_chain(og, step, rng_deg, bruteforce, ss, n_threads=8)
    assert logs[0] == ol
    assert geoms_equal(g, og)


@settings(max_examples=30 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), world=st.sampled_from([2, 3, 5, 8]), exchange=st.sampled_from(["gather", "device"]),
       n_frames=st.integers(2, 7), n_points=st.sampled_from([20, 64, 200, 501]), ss=st.sampled_from([16, 200, 501]),
       bruteforce=st.booleans(), step=st.sampled_from([0.05, 0.5, 1.0, 9.0, 45.0]),
       rng_deg=st.sampled_from([20.0, 90.0, 180.0]), prec=st.sampled_from([2, 3]), circular=st.booleans(),
       always_bound=st.booleans())
def test_sharded_plan_equals_the_oracle_chain(engine, oracle, mm, seed, world, exchange, n_frames, n_points, ss, bruteforce,
                                              step, rng_deg, prec, circular, always_bound):
    """The candidate axis split over `world` plans (what `world` ranks run), either exchange: every rank ends with
    the oracle chain's logs and coordinates -- random worlds (also more ranks than candidates), grids, ladders,
    near-circular frames whose near-ties straddle shard borders."""
    from helpers import geoms_equal, to_oracle
    from test_gpu_sharded import drive_sharded
    if bruteforce and step < 0.5:
        step = 0.5
    g = mm.synthetic_pullback(n_frames, n_points, pullback_id=seed % 4, seed=seed % 1000, torsion_sigma_deg=3.0)
    if circular:
        c = g.centroids[np.repeat(np.arange(g.n_frames), np.diff(g.lumen_off)), :2]
        d = g.lumen[:, :2] - c
        r = np.hypot(d[:, 0], d[:, 1])[:, None]
        g.lumen[:, :2] = c + d * (0.15 + 0.85 * (2.0 / r))
    og = to_oracle(oracle, g)
    ol = oracle.align_within_chain(og, step, rng_deg, bruteforce, ss, n_threads=8)
    if always_bound:
        engine.set_bound_min_candidates(0)
    try:
        for geoms, logs, _evals, _unres in drive_sharded(mm, engine, [g], world, step, rng_deg, bruteforce, ss, prec, exchange):
            assert logs[0] == ol
            assert geoms_equal(geoms[0], og)
    finally:
        engine.set_bound_min_candidates(16384)


@settings(max_examples=40 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), fa=st.integers(1, 24), fb=st.integers(1, 24), n_points=st.sampled_from([6, 40, 120, 501]),
       ss=st.sampled_from([1, 50, 500, 640]), step=st.sampled_from([0.005, 0.05, 0.5, 1.0, 3.0]),
       rng_deg=st.sampled_from([6.0, 30.0, 90.0, 180.0]), twist=st.sampled_from([0.0, 15.0, -70.0, 180.0]),
       prec=st.sampled_from([0, 1, 2, 3]))
def test_between_alignment_matches_the_oracle(engine, oracle, mm, seed, fa, fb, n_points, ss, step, rng_deg, twist, prec):
    """align_between_geometries (align_between.rs:11-68): sampling of both pullbacks, the global centroid, the
    hierarchical search without the angle-0 shortcut, rotation and translation of the target -- best angle and every
    coordinate of both geometries equal the oracle's, for pullbacks of different lengths and point counts."""
    from helpers import geoms_equal, to_oracle
    a = mm.synthetic_pullback(fa, n_points, pullback_id=0, seed=seed % 1000)
    b = mm.synthetic_pullback(fb, n_points, pullback_id=1, seed=(seed // 7) % 1000)
    if twist:
        th = math.radians(twist)
        c = b.lumen[:, :2].mean(axis=0)
        d = b.lumen[:, :2] - c
        b.lumen[:, 0] = c[0] + d[:, 0] * math.cos(th) - d[:, 1] * math.sin(th)
        b.lumen[:, 1] = c[1] + d[:, 0] * math.sin(th) + d[:, 1] * math.cos(th)
    oa, ob = to_oracle(oracle, a), to_oracle(oracle, b)
    best, _ = mm.align_between(engine, [(a, b)], rng_deg, step, ss, precision=prec)
    obest = oracle.align_between(oa, ob, rng_deg, step, ss, n_threads=8)
    assert best[0] == obest
    assert geoms_equal(a, oa) and geoms_equal(b, ob)


@settings(max_examples=20 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), n_frames=st.integers(3, 18), n_points=st.sampled_from([16, 64, 200]),
       n_ccta=st.sampled_from([60, 900, 6000]), idx_range=st.integers(0, 3), rng_deg=st.sampled_from([0.0, 3.0, 8.0]),
       step_deg=st.sampled_from([0.5, 1.0, 4.0]), clutter=st.sampled_from([0.0, 0.1]), true_index=st.integers(0, 12),
       start_shift=st.integers(-3, 3))
def test_refine_grid_matches_the_oracle(engine, oracle, mm, seed, n_frames, n_points, n_ccta, idx_range, rng_deg, step_deg,
                                        clutter, true_index, start_shift):
    """refine_alignment_hausdorff (align_algorithms.rs:339-451): every candidate's cost, the candidate order and the
    strict-< first minimum equal the oracle's, for random grids (also the one-candidate grid), start indices near
    the ends of the centerline (skipped candidates) and clouds of any size; the winner-only path agrees."""
    from oracle import oracle_cl as ocl
    from helpers import to_oracle, to_oracle_cl
    ocl.lib()
    case = mm.synth.synthetic_centerline_case(n_frames=n_frames, n_points=n_points, n_ccta=n_ccta, seed=seed % 997,
                                              true_rotation_deg=float(seed % 360) - 180.0, true_index=true_index,
                                              clutter_frac=clutter)
    g = case["geometry"]
    aligned, _, _ = mm.align_three_point(case["centerline"], g, case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"],
                                         angle_step_deg=2.0)
    rcl, _ = mm.preprocess_centerline(case["centerline"], g)
    idx0 = max(0, rcl.find_reference_cl_point_idx(case["main_ref_pt"]) + start_shift)
    og, orcl = to_oracle(oracle, aligned), to_oracle_cl(ocl, rcl)
    a, s = math.radians(rng_deg), math.radians(step_deg)
    r = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.0, case["points"], a, s, idx_range)
    o = ocl.refine_alignment_hausdorff([og], orcl, idx0, 0.0, case["points"], a, s, idx_range)
    assert np.array_equal(r[3], o[3]) and r[:3] == o[:3]
    w = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.0, case["points"], a, s, idx_range,
                                                 return_costs=False)
    assert w[:3] == o[:3]


@settings(max_examples=40 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), na=st.sampled_from([1, 2, 255, 256, 257, 511, 512, 513, 1500, 4096, 4097, 9000]),
       nb=st.sampled_from([1, 2, 511, 512, 513, 1024, 1025, 4095, 4096, 7000]), shape=st.sampled_from(["cloud", "tube", "line", "dups"]),
       offset=st.sampled_from([0.0, 0.3, 25.0]))
def test_nearest_neighbour_minima_match_the_oracle(engine, mm, seed, na, nb, shape, offset):
    """k_nn3_min (the CCTA diameter search's kernel): per-query squared nearest-neighbour distances in 3-D, bit for
    bit, around the 512-point chunk / 4096-point slab-order boundaries, for thin tubes (what the pruning is built
    for), lines, duplicates and clouds that do not overlap at all."""
    from oracle import oracle_ccta as occ
    occ.lib()
    rng = np.random.default_rng(seed)

    def cloud(n):
        if shape == "cloud":
            return rng.normal(0, 4, size=(n, 3))
        if shape == "tube":
            s = rng.uniform(0, 60, n); t = rng.uniform(0, 2 * math.pi, n)
            return np.stack([2 * np.cos(t) + 5 * np.sin(s / 9), 2 * np.sin(t), s], 1) + rng.normal(0, 0.02, (n, 3))
        if shape == "line":
            s = rng.uniform(-30, 30, n)
            return np.stack([s, 0.5 * s, -s], 1)
        base = rng.normal(0, 3, size=(max(1, min(7, n)), 3))
        return base[rng.integers(0, base.shape[0], n)]
    origin = np.array([10.0, -190.0, 1700.0])
    a, b = cloud(na) + origin, cloud(nb) + origin + offset
    assert np.array_equal(mm.ccta.nn_min_sq(a, b, engine=engine), occ.nn_min_sq(a, b))
    assert mm.ccta.symmetric_nn_distance(a, b, engine=engine) == occ.symmetric_nn_distance(a, b)


@settings(max_examples=15 * int(os.environ.get("MM_HYP_SCALE", "1")), deadline=None, derandomize=True, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), n_frames=st.integers(4, 14), n_points=st.sampled_from([24, 60, 120]),
       hole=st.booleans(), with_eem=st.booleans(), thick=st.sampled_from([None, 0.7, 1.1]), smooth=st.booleans(),
       post=st.booleans(), bruteforce=st.booleans(), step=st.sampled_from([0.5, 1.0, 3.0]),
       mode=st.sampled_from(["single", "singlepair", "full"]))
def test_entry_points_native_bookkeeping_equals_the_python_checkers(engine, mm, seed, n_frames, n_points, hole, with_eem, thick,
                                                                    smooth, post, bruteforce, step, mode):
    """from_array_single / _singlepair / _full end to end with the searches on the device: the native builder, post-steps
    and pair post-processing (the product path) against the same calls with every host piece switched to its Python
    checker (MM_PY_BUILDER, MM_PY_POSTPROC) -- logs, frame counts and every array of every returned geometry, over
    pullbacks with missing frames, EEM contours, measured thicknesses (anomalous coronaries), with and without
    smoothing and post-processing."""
    from test_golden_and_api import _array_input
    from test_native_frames import assert_same

    def inputs():
        out = []
        for q in range({"single": 1, "singlepair": 2, "full": 4}[mode]):
            drop = (2 + (seed + q) % max(n_frames - 3, 1),) if hole else ()
            d = _array_input(mm, n_frames=n_frames, n_points=n_points, drop=drop, with_eem=with_eem, thickness=thick,
                             seed=(seed + 17 * q) % 1000)
            d.diastole = q % 2 == 0
            out.append(d)
        return out

    kw = dict(step_rotation_deg=step, range_rotation_deg=30.0, bruteforce=bruteforce, smooth=smooth, write_obj=False,
              engine=engine)
    fn = {"single": mm.from_array_single, "singlepair": mm.from_array_singlepair, "full": mm.from_array_full}[mode]
    if mode != "single":
        kw["postprocessing"] = post
    mp = pytest.MonkeyPatch()
    try:
        got = fn(*inputs(), **kw)
        mp.setenv("MM_PY_BUILDER", "1"); mp.setenv("MM_PY_POSTPROC", "1")
        want = fn(*inputs(), **kw)
    finally:
        mp.undo()

    def geoms(r):
        if mode == "single":
            return [r[0]], r[1]
        pairs = r[:-1] if mode == "full" else [r[0]]
        return [g for p in pairs for g in (p.geom_a, p.geom_b)], r[-1]
    ga, la = geoms(got)
    gb, lb = geoms(want)
    assert la == lb and len(ga) == len(gb)
    for x, y in zip(ga, gb):
        assert_same(x, y, mode)
