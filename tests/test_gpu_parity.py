"""GPU parity tests: every call goes through the C ABI (libmm_hausdorff.so) and is compared
with the CPU oracle on the same seeded inputs.  Bar: bit-exact for costs scored in f64,
for winners (index, angle) and for the geometry the chain produces; the f32 screening costs
must stay within the stated bound delta = 24 * 2^-24 * (rho_ref + rho_tgt)."""
import math

import numpy as np
import pytest

import refgeom
from helpers import blob, geoms_equal, to_oracle

pytestmark = pytest.mark.gpu

U24 = 2.0 ** -24


def P(*xy):
    return np.array(xy, dtype=np.float64).reshape(-1, 2)


# ---------------------------------------------------------------------------------------
# the metric (process_utils.rs:78-121) -- reference KATs through the device path
# ---------------------------------------------------------------------------------------
def test_hausdorff_kats(engine):
    pts = P((0, 0), (1, 0), (0, 1))
    assert engine.hausdorff(pts, pts) == 0.0
    assert engine.hausdorff(P((0, 0), (1, 0)), P((2, 0), (3, 0))) == 2.0
    assert engine.hausdorff(P((0, 0), (3, 0)), P((1, 0), (2, 0), (4, 0))) == 1.0
    e = np.zeros((0, 2))
    assert engine.hausdorff(e, P((1, 1))) == 0.0 and engine.hausdorff(P((1, 1)), e) == 0.0
    assert engine.hausdorff(e, e) == 0.0
    i = np.arange(100, dtype=np.float64)
    assert engine.hausdorff(np.stack([i, 0 * i], 1), np.stack([i + 0.5, 0 * i], 1)) == 0.5
    sq, di = P((0, 0), (2, 0), (2, 2), (0, 2)), P((1, 0), (2, 1), (1, 2), (0, 1))
    assert engine.hausdorff(sq, di) == 1.0


@pytest.mark.parametrize("na,nb", [(1, 1), (1, 7), (6, 6), (17, 33), (64, 65), (100, 90), (208, 208), (224, 225),
                                   (272, 288), (289, 300), (520, 520), (521, 521), (544, 512), (545, 521),
                                   (600, 1000), (1100, 40), (33, 2049), (2000, 3000)])
def test_hausdorff_random_bit_exact(engine, oracle, na, nb):
    rng = np.random.default_rng(na * 10007 + nb)
    a = rng.normal(4.5, 1.5, size=(na, 2))
    b = rng.normal(4.6, 1.4, size=(nb, 2))
    assert engine.hausdorff(a, b) == oracle.hausdorff(a, b)
    assert engine.hausdorff(b, a) == oracle.hausdorff(b, a)


def test_search_target_set_beyond_lds_budget_is_an_error(engine, mm):
    """A rotation SEARCH stages the rotated target in LDS (f64 re-score: 4080 points); beyond that
    it fails loudly.  The plain metric has no such limit (streaming kernel, next tests)."""
    a = np.zeros((8, 2))
    b = np.zeros((9000, 2))
    with pytest.raises(RuntimeError, match="LDS budget"):
        engine.best_rotation(a, b, np.array([0.0, 0.1]), (0.0, 0.0))


# ---------------------------------------------------------------------------------------
# one search: costs and winner
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,step,rng_deg,between", [(6, 1.0, 30.0, False), (120, 1.0, 180.0, False),
                                                    (208, 0.5, 90.0, True), (521, 1.0, 180.0, False),
                                                    (521, 5.0, 180.0, True), (505, 2.0, 90.0, True)])
def test_search_f64_costs_bit_exact(engine, oracle, mm, n, step, rng_deg, between):
    rng = np.random.default_rng(n + int(step * 10))
    ref = blob(rng, n)
    th = math.radians(7.3)
    c = ref.mean(axis=0)
    tgt = (ref - c) @ np.array([[math.cos(th), math.sin(th)], [-math.sin(th), math.cos(th)]]) + c
    tgt = tgt + rng.normal(0, 0.01, tgt.shape)
    angles, deg, _ = mm.search_angles(step, rng_deg)
    assert not deg
    centre = (float(c[0]), float(c[1]))
    bi, ba, bc, costs = engine.best_rotation(ref, tgt, angles, centre, skip_zero=not between,
                                             precision=mm.MM_PRECISION_F64, return_costs=True)
    ocosts = oracle.costs_over_angles(ref, tgt, angles, centre[0], centre[1], between=between)
    assert np.array_equal(costs, ocosts)
    obi = int(np.argmin(ocosts))  # numpy argmin == first minimum
    assert bi == obi and ba == angles[obi] and bc == ocosts[obi]


@pytest.mark.parametrize("n,step,rng_deg", [(6, 1.0, 30.0), (200, 0.5, 180.0), (521, 0.5, 180.0), (520, 1.0, 90.0)])
def test_search_f32_screen_winner_bit_exact_and_bound(engine, oracle, mm, n, step, rng_deg):
    rng = np.random.default_rng(1000 + n)
    ref = blob(rng, n)
    tgt = blob(rng, n) + rng.normal(0, 0.02, (n, 2))
    c = ref.mean(axis=0)
    centre = (float(c[0]), float(c[1]))
    angles, _, _ = mm.search_angles(step, rng_deg)
    bi, ba, bc, costs = engine.best_rotation(ref, tgt, angles, centre, skip_zero=True,
                                             precision=mm.MM_PRECISION_F32, return_costs=True)
    ocosts = oracle.costs_over_angles(ref, tgt, angles, centre[0], centre[1])
    obi = int(np.argmin(ocosts))
    assert bi == obi and ba == angles[obi]
    assert bc == ocosts[obi]                      # the winner is re-scored in f64: identical bits
    rho = np.hypot(*(ref - c).T).max() + np.hypot(*(tgt - c).T).max()
    delta = 24 * U24 * rho
    err = np.abs(costs - ocosts).max()
    assert err <= delta, (err, delta)             # stated f32 tolerance on the Hausdorff values
    assert err <= 0.5 * delta                     # and it is not a tight squeeze


def test_duplicate_candidates_first_index_wins(engine, oracle, mm):
    """range == limes == 180 deg: first and last candidate both wrap to -pi."""
    rng = np.random.default_rng(5)
    ref = blob(rng, 150)
    c = ref.mean(axis=0)
    tgt = (ref - c) @ np.array([[-1.0, 0.0], [0.0, -1.0]]) + c  # rotated by pi: best angle is +-pi
    angles, _, _ = mm.search_angles(1.0, 180.0)
    assert angles[0] == angles[-1] == -math.pi
    for prec in (mm.MM_PRECISION_F64, mm.MM_PRECISION_F32):
        bi, ba, bc = engine.best_rotation(ref, tgt, angles, (float(c[0]), float(c[1])), precision=prec)
        ocosts = oracle.costs_over_angles(ref, tgt, angles, float(c[0]), float(c[1]))
        assert bi == int(np.argmin(ocosts)) == 0 and bc == ocosts[0]


def test_point_sets_far_smaller_than_their_coordinates(engine, oracle, mm):
    """All points of both sets equal (found by tests/test_gpu_property.py): every exact cost is 0.0 -- the reference's
    rotation ends with `+ cx` and rounds at the coordinates' magnitude -- so the FIRST candidate wins
    (process_utils.rs:72), while the screen, working relative to the rotation centre, sees 1e-17 for every angle but
    0.  The screening bound carries the reference's own f64 rounding for exactly this case."""
    p = np.array([[5.554607710218, 3.975742059671]] * 3)
    c = p.mean(axis=0)                                        # one ulp off the point in y
    angles, _, _ = mm.search_angles(0.5, 3.0)
    ocosts = oracle.costs_over_angles(p, p, angles, float(c[0]), float(c[1]))
    assert not ocosts.any()
    for q in (p, p + np.array([[0.0, 0.0], [3e-16, 0.0], [0.0, 5e-16]])):      # and a cloud of a few ulps
        oc = oracle.costs_over_angles(p, q, angles, float(c[0]), float(c[1]))
        for prec in (mm.MM_PRECISION_F64, mm.MM_PRECISION_F32, mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED, mm.MM_PRECISION_F32_MATRIX):
            bi, ba, bc = engine.best_rotation(p, q, angles, (float(c[0]), float(c[1])), skip_zero=True, precision=prec)
            assert bi == int(np.argmin(oc)) and ba == angles[bi] and bc == oc[bi], prec


@pytest.mark.parametrize("n", [200, 521, 1200])
def test_non_finite_coordinates_follow_the_reference(engine, oracle, mm, n):
    """process_utils.rs:103-114: `if d2 < min_sq` never selects a NaN, and `if min_sq.is_finite() && min_sq > local_max`
    leaves a point out whose distances are all inf / NaN -- so with NaN, inf or overflowing coordinates the reference
    still returns a finite distance (0.0 if nothing is left) and a winner (the first candidate if all costs tie at 0).
    The exact kernels apply the same filter; a set with such coordinates makes its level skip the f32 screens (they
    cannot bound non-finite values) and score every candidate exactly.  Metric and search at all four precisions
    against the oracle."""
    rng = np.random.default_rng(n)
    ref, tgt = blob(rng, n), blob(rng, n)
    angles, _, _ = mm.search_angles(1.0, 30.0)
    c = tgt.mean(axis=0)
    centre = (float(c[0]), float(c[1]))
    idx = np.arange(n)[:, None]
    cases = [(ref, np.where(idx == 7, np.nan, tgt)), (np.where(idx == 3, np.inf, ref), tgt),
             (np.where(idx % 50 == 3, np.inf, ref), np.where(idx % 40 == 7, np.nan, tgt)),
             (ref * np.nan, tgt * np.nan), (ref * 1e200, tgt * 1e200), (ref, np.where(idx == 0, -np.inf, tgt))]
    for r, t in cases:
        assert engine.hausdorff(r, t) == oracle.hausdorff(r, t)
        o_angle = oracle.bruteforce_rotation(r, t, 1.0, 30.0, centre[0], centre[1], n_threads=8)
        o_cost = oracle.cost_within(r, t, o_angle, centre[0], centre[1])
        assert math.isfinite(o_cost)
        for prec in (mm.MM_PRECISION_F64, mm.MM_PRECISION_F32, mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED, mm.MM_PRECISION_F32_MATRIX):
            bi, ba, bc = engine.best_rotation(r, t, angles, centre, skip_zero=True, precision=prec)
            assert bi >= 0 and ba == o_angle and bc == o_cost, prec


@pytest.mark.parametrize("scale", [1e17, 3e18, 1e19, 1e25, 1e30, 1e38, 1e-14, 1e-16, 1e-20, 1e-30])
def test_coordinates_beyond_the_f32_range_of_the_screens(engine, oracle, mm, scale):
    """Finite coordinates whose squared distances overflow (or underflow) f32: a screen must not drop an
    overflowed row (that would understate the candidate and could shortlist past the true first minimum); sets beyond
    what f32 holds take the exact f64 kernel.  One far outlier in an otherwise mm-sized set as well.  Winner, angle
    and cost against the oracle at all four precisions."""
    rng = np.random.default_rng(11)
    n = 300
    ref, tgt = blob(rng, n), blob(rng, n)
    angles, _, _ = mm.search_angles(1.0, 45.0)
    idx = np.arange(n)[:, None]
    cases = [(ref * scale, tgt * scale)]
    if scale > 1:
        cases.append((np.where(idx == 5, ref * scale, ref), tgt))
        cases.append((ref, np.where(idx == 9, tgt * scale, tgt)))
    for r, t in cases:
        c = t.mean(axis=0) if np.isfinite(t.mean(axis=0)).all() else np.zeros(2)
        centre = (float(c[0]), float(c[1]))
        oc = oracle.costs_over_angles(r, t, angles, centre[0], centre[1])
        assert np.isfinite(oc).all()
        want = int(np.argmin(oc))
        assert engine.hausdorff(r, t) == oracle.hausdorff(r, t)
        for prec in (mm.MM_PRECISION_F64, mm.MM_PRECISION_F32, mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED, mm.MM_PRECISION_F32_MATRIX):
            bi, ba, bc = engine.best_rotation(r, t, angles, centre, skip_zero=True, precision=prec)
            assert bi == want and ba == angles[want] and bc == oc[want], (prec, scale)


@pytest.mark.parametrize("what", [np.nan, np.inf])
@pytest.mark.parametrize("mode", [0, 1])
def test_chain_with_non_finite_points_equals_the_oracle_chain(engine, oracle, mm, mode, what):
    """A pullback with a few NaN / inf lumen coordinates: the chain (faithful and decoupled, every precision) gives the
    oracle chain's logs and coordinates -- NaNs at the same places."""
    for prec in (mm.MM_PRECISION_F64, mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED):
        g = mm.synthetic_pullback(8, 120, pullback_id=1, seed=5)
        g.lumen[g.lumen_off[3] + 18, 0] = what
        g.lumen[g.lumen_off[5] + 2, 1] = -what
        og = to_oracle(oracle, g)
        logs, _ = mm.align_within(engine, [g], 1.0, 30.0, True, 100, precision=prec, mode=mode)
        assert logs[0] == oracle.align_within_chain(og, 1.0, 30.0, True, 100, n_threads=8)
        assert geoms_equal(g, og)


def test_all_candidates_tie_circle(engine, oracle, mm):
    """A perfectly symmetric target: costs tie to within rounding; the f32 screen must hand
    every near-tie to the exact re-score and still return the reference's first minimum."""
    t = np.arange(360) * (2 * math.pi / 360)
    ref = np.stack([4.5 + 2 * np.cos(t), 4.5 + 2 * np.sin(t)], 1)
    tgt = ref.copy()
    angles, _, _ = mm.search_angles(1.0, 180.0)
    ocosts = oracle.costs_over_angles(ref, tgt, angles, 4.5, 4.5)
    out = engine.best_rotation(ref, tgt, angles, (4.5, 4.5), precision=mm.MM_PRECISION_F32)
    assert out[0] == int(np.argmin(ocosts)) and out[2] == ocosts.min()


# ---------------------------------------------------------------------------------------
# batches: ragged sets, empty sets, per-pair candidate lists and flags
# ---------------------------------------------------------------------------------------
def test_batch_ragged_and_empty(engine, oracle, mm):
    rng = np.random.default_rng(77)
    sizes = [(6, 6), (40, 0), (0, 12), (521, 521), (100, 333), (17, 1), (300, 299), (64, 64)]
    refs = [blob(rng, a) if a else np.zeros((0, 2)) for a, _ in sizes]
    tgts = [blob(rng, b) + 0.05 if b else np.zeros((0, 2)) for _, b in sizes]
    lists = [mm.search_angles(s, r, c, lim)[0] for s, r, c, lim in
             [(1.0, 30.0, None, 30.0), (2.0, 20.0, None, 20.0), (2.0, 20.0, None, 20.0), (1.0, 180.0, None, 180.0),
              (0.5, 5.0, 0.3, 90.0), (3.0, 90.0, None, 90.0), (0.1, 5.0, -0.4, 45.0), (10.0, 180.0, None, 180.0)]]
    lists[5] = np.array([0.0])  # a single candidate, angle == 0.0
    centres = [(4.5, 4.5)] * len(sizes)
    flags = [1, 1, 1, 1, 0, 1, 0, 1]
    batch = mm.Batch(refs, tgts, lists, centres, flags)
    for prec in (mm.MM_PRECISION_F64, mm.MM_PRECISION_F32):
        out = engine.best_rotation_batch(batch, precision=prec, return_costs=True)
        for p, (r, t, al, fl) in enumerate(zip(refs, tgts, lists, flags)):
            oc = oracle.costs_over_angles(r, t, al, 4.5, 4.5, between=(fl == 0))
            obi = int(np.argmin(oc))
            assert out["best_idx"][p] == obi, (p, prec)
            assert out["best_angle"][p] == al[obi]
            assert out["best_cost"][p] == oc[obi]
            got = out["costs"][batch.ang_off[p]:batch.ang_off[p + 1]]
            if prec == mm.MM_PRECISION_F64 or len(r) == 0 or len(t) == 0:
                assert np.array_equal(got, oc)
            else:
                assert np.abs(got - oc).max() <= 24 * U24 * 12.0
        if prec == mm.MM_PRECISION_F32:
            assert (out["n_rescored"][[0, 3, 4, 6, 7]] >= 1).all()
            assert out["n_rescored"][3] < len(lists[3])   # the screen prunes almost everything


@pytest.mark.parametrize("prec", [1, 2, 3])
def test_plan_slices_partition_the_candidate_axis(engine, oracle, mm, prec):
    """Sharding the candidate axis (what each rank does at N GPUs): min over slices of
    (cost, index) == the unsharded winner.  prec 3 = the bounded screen: every slice prunes against
    its own best."""
    rng = np.random.default_rng(9)
    refs = [blob(rng, 260) for _ in range(5)]
    tgts = [blob(rng, 260) for _ in range(5)]
    angles, _, _ = mm.search_angles(1.0, 180.0)
    batch = mm.Batch(refs, tgts, [angles] * 5, [(4.5, 4.5)] * 5, [1] * 5)
    full = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32)
    world = 4
    per = (len(angles) + world - 1) // world
    best_cost = np.full(5, np.inf)
    best_idx = np.full(5, 2**31 - 1, dtype=np.int64)
    for r in range(world):
        plan = engine.plan(batch, prec, r * per, min((r + 1) * per, len(angles)))
        plan.run()
        res = plan.fetch()
        plan.close()
        for p in range(5):
            c, i = res["best_cost"][p], res["best_idx"][p]
            if i >= 0 and (c < best_cost[p] or (c == best_cost[p] and i < best_idx[p])):
                best_cost[p], best_idx[p] = c, i
    assert np.array_equal(best_idx, full["best_idx"]) and np.array_equal(best_cost, full["best_cost"])


# ---------------------------------------------------------------------------------------
# the within-pullback chain (align_within.rs:24-134)
# ---------------------------------------------------------------------------------------
def test_chain_dummy_geometry_reference_expectation(engine, oracle, mm):
    """align_within.rs:791-830: rot = -15 +- 1e-6, t = (-i, -i); and identical to the oracle."""
    g = mm.FlatGeometry.from_frames(**refgeom.to_arrays(refgeom.dummy_frames()))
    og = to_oracle(oracle, g)
    logs, _ = mm.align_within(engine, [g], 0.01, 30.0, False, 6)
    ologs = oracle.align_within_chain(og, 0.01, 30.0, False, 6)
    assert logs[0] == ologs
    for i, (_, _, rot, tx, ty, _, _) in enumerate(logs[0]):
        assert rot == pytest.approx(-15.0, abs=1e-6)
        assert tx == pytest.approx(-(i + 1.0), abs=1e-6) and ty == pytest.approx(-(i + 1.0), abs=1e-6)
    assert geoms_equal(g, og)


@pytest.mark.parametrize("bruteforce,step,rng_deg,ss,prec", [
    (True, 1.0, 180.0, 501, 1), (True, 2.0, 90.0, 500, 0), (False, 0.5, 90.0, 500, 1),
    (False, 0.05, 45.0, 200, 1), (False, 0.005, 20.0, 501, 1), (True, 0.5, 180.0, 501, 1)])
def test_chain_synthetic_bit_identical(engine, oracle, mm, bruteforce, step, rng_deg, ss, prec):
    geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((7, 5, 7, 6))]
    ogeoms = [to_oracle(oracle, g) for g in geoms]
    logs, evals = mm.align_within(engine, geoms, step, rng_deg, bruteforce, ss, precision=prec)
    for g, og, lg in zip(geoms, ogeoms, logs):
        ol = oracle.align_within_chain(og, step, rng_deg, bruteforce, ss, n_threads=8)
        assert lg == ol
        assert geoms_equal(g, og)
    assert evals > 0


@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", [
    (True, 1.0, 180.0, 501), (True, 0.5, 180.0, 500), (False, 0.5, 90.0, 500), (False, 0.05, 45.0, 200),
    (False, 0.005, 20.0, 501), (True, 3.0, 45.0, 64)])
def test_decoupled_mode_bit_identical_to_chain_oracle(engine, oracle, mm, bruteforce, step, rng_deg, ss):
    """mode 1: all frame pairs screened in one launch on the ORIGINAL frames, then the exact
    chain walk.  Logs and geometry must equal the oracle's sequential chain bit for bit."""
    geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((9, 6, 9, 7))]
    ogeoms = [to_oracle(oracle, g) for g in geoms]
    wp = mm.WithinPlan(engine, geoms, step, rng_deg, bruteforce, ss)
    logs, evals, unresolved = wp.run()
    wp.close()
    total_ref_evals = 0
    for g, og, lg in zip(geoms, ogeoms, logs):
        ol = oracle.align_within_chain(og, step, rng_deg, bruteforce, ss, n_threads=8)
        assert lg == ol
        assert geoms_equal(g, og)
    assert unresolved <= 2            # generic data: (almost) every step is resolved by the one-shot screen
    if bruteforce:
        n_ang = len(mm.search_angles(step, rng_deg)[0])
        assert evals == n_ang * sum(g.n_frames - 1 for g in geoms)


def test_decoupled_mode_symmetric_shapes_fall_back_to_chain_state(engine, oracle, mm):
    """Perfect circles: every candidate ties up to rounding, so the one-shot screen cannot
    decide and each step must be re-searched on the chain state -- still bit-identical."""
    t = np.arange(120) * (2 * math.pi / 120)
    lum = [np.stack([4.5 + 0.01 * k + 2 * np.cos(t), 4.4 + 2 * np.sin(t), np.full_like(t, 0.5 * k)], 1) for k in range(5)]
    g = mm.FlatGeometry.from_frames(lum, ref_points={0: (6.5, 4.4, 0.0)})
    og = to_oracle(oracle, g)
    logs, _ = mm.align_within(engine, [g], 2.0, 60.0, True, 120, mode=1)
    assert logs[0] == oracle.align_within_chain(og, 2.0, 60.0, True, 120)
    assert geoms_equal(g, og)
    # the reference's own dummy geometry through the decoupled path (align_within.rs:791-830)
    g2 = mm.FlatGeometry.from_frames(**refgeom.to_arrays(refgeom.dummy_frames()))
    og2 = to_oracle(oracle, g2)
    logs2, _ = mm.align_within(engine, [g2], 0.01, 30.0, False, 6, mode=1)
    assert logs2[0] == oracle.align_within_chain(og2, 0.01, 30.0, False, 6)
    assert geoms_equal(g2, og2)
    for (_, _, rot, *_r) in logs2[0]:
        assert rot == pytest.approx(-15.0, abs=1e-6)


def test_chain_validation_errors(engine, mm):
    g = mm.synthetic_pullback(3, 32)
    with pytest.raises(RuntimeError, match="sample_size must be > 0"):
        mm.align_within(engine, [g], 1.0, 30.0, False, 0)


# ---------------------------------------------------------------------------------------
# between pullbacks (align_between.rs:11-68)
# ---------------------------------------------------------------------------------------
def test_between_dummy_reference_expectation(engine, oracle, mm):
    """align_between.rs:280-303: B = A rotated by 15 deg; afterwards all points equal within 1e-6."""
    fa = refgeom.dummy_aligned_long_frames()
    fb = refgeom.dummy_aligned_long_frames()
    refgeom.rotate_geometry(fb, math.radians(15.0))
    ga = mm.FlatGeometry.from_frames(**refgeom.to_arrays(fa))
    gb = mm.FlatGeometry.from_frames(**refgeom.to_arrays(fb))
    oa, ob = to_oracle(oracle, ga), to_oracle(oracle, gb)
    best, _ = mm.align_between(engine, [(ga, gb)], 30.0, 0.01, 6)
    obest = oracle.align_between(oa, ob, 30.0, 0.01, 6)
    assert best[0] == obest
    assert geoms_equal(gb, ob) and geoms_equal(ga, oa)
    np.testing.assert_allclose(ga.lumen, gb.lumen, atol=1e-6)


def test_between_synthetic_bit_identical(engine, oracle, mm):
    geoms = mm.synthetic_case(10, 501)
    mm.align_within(engine, geoms, 1.0, 60.0, False, 500)
    og = [to_oracle(oracle, g) for g in geoms]
    pairs = [(geoms[0], geoms[1]), (geoms[2], geoms[3])]
    best, evals = mm.align_between(engine, pairs, 90.0, 0.5, 500)
    ob0 = oracle.align_between(og[0], og[1], 90.0, 0.5, 500, n_threads=8)
    ob1 = oracle.align_between(og[2], og[3], 90.0, 0.5, 500, n_threads=8)
    assert best[0] == ob0 and best[1] == ob1
    for g, o in zip(geoms, og):
        assert geoms_equal(g, o)
    assert evals > 0


# ---------------------------------------------------------------------------------------
# full-size properties (BASELINE configs: N = 521 pts/set, 361 / 721 candidates)
# ---------------------------------------------------------------------------------------
def test_fullsize_known_rotation_recovered(engine, mm):
    """target = reference rotated by -theta_k (theta_k on the candidate grid) about the
    centre => the search returns exactly candidate k with a ~0 cost (size-independent property)."""
    rng = np.random.default_rng(4)
    angles, _, _ = mm.search_angles(0.5, 180.0)
    assert len(angles) == 721
    refs, tgts, want = [], [], []
    for p in range(64):
        ref = blob(rng, 521)
        k = int(rng.integers(1, 720))
        c, s = math.cos(-angles[k]), math.sin(-angles[k])
        rel = ref - 4.5
        tgt = np.stack([rel[:, 0] * c - rel[:, 1] * s, rel[:, 0] * s + rel[:, 1] * c], 1) + 4.5
        refs.append(ref); tgts.append(tgt); want.append(k)
    batch = mm.Batch(refs, tgts, [angles] * 64, [(4.5, 4.5)] * 64, [1] * 64)
    out = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32)
    assert list(out["best_idx"]) == want
    assert out["best_cost"].max() < 1e-12


def test_fullsize_f32_and_f64_paths_agree(engine, mm):
    """Same batch through the screening path and the all-f64 path: identical winners and
    identical winning costs (the claim behind MM_PRECISION_F32)."""
    geoms = mm.synthetic_case(17, 501)
    refs, tgts, cs = [], [], []
    for g in geoms:
        for i in range(1, g.n_frames):
            r = mm.search_set(g, i - 1, 501) - g.centroids[i - 1, :2]
            t = mm.search_set(g, i, 501) - g.centroids[i, :2]
            refs.append(r); tgts.append(t); cs.append((0.0, 0.0))
    angles, _, _ = mm.search_angles(1.0, 180.0)
    batch = mm.Batch(refs, tgts, [angles] * len(refs), cs, [1] * len(refs))
    a = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32)
    b = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F64)
    assert np.array_equal(a["best_idx"], b["best_idx"])
    assert np.array_equal(a["best_cost"], b["best_cost"])
    assert a["n_rescored"].sum() < 0.05 * len(refs) * len(angles)


# ---------------------------------------------------------------------------------------
# third call site: refine_alignment_hausdorff's candidate grid (align_algorithms.rs:369-441)
# ---------------------------------------------------------------------------------------
def test_refine_grid_costs_and_selection(engine, oracle, mm):
    """(index shift x accumulated angle) candidates: for each, hausdorff(filtered CCTA points,
    placed + downsampled frames) on x,y only, strict `<` keeps the first minimum.  The placement
    here is a plain 2-D stand-in (the reference's nalgebra placement is host code outside this
    path); oracle and device get identical sets."""
    rng = np.random.default_rng(21)
    n_frames, m = 12, 200
    t = np.arange(m) * (2 * math.pi / m)
    frames = [np.stack([1.5 * np.cos(t) * (1 + 0.1 * np.sin(3 * t + k)), 1.2 * np.sin(t), np.full(m, 0.5 * k)], 1)
              for k in range(n_frames)]
    clutter = rng.uniform(7.0, 12.0, size=(4000, 3)) * rng.choice([-1.0, 1.0], size=(4000, 3))
    cloud = np.concatenate([f + rng.normal(0, 0.01, f.shape) for f in frames] + [clutter])  # wall points + far clutter
    angles = mm.refine_angles(0.0, math.radians(8.0), math.radians(1.0))
    cands, sets = [], []
    for delta in range(-2, 3):
        start = np.array([0.1 * delta, 0.0, 0.0]); end = np.array([0.1 * delta, 0.0, 0.5 * (n_frames - 1)])
        idx = mm.filter_points_in_region(cloud, start, end)
        filtered = cloud[idx]
        nd = mm.refine_downsample_count(len(filtered), m, n_frames)
        for ang in angles:
            c, s = math.cos(ang), math.sin(ang)
            placed = []
            for f in frames:
                g = np.stack([f[:, 0] * c - f[:, 1] * s + 0.1 * delta, f[:, 0] * s + f[:, 1] * c, f[:, 2]], 1)
                placed.append(oracle.downsample(g, nd) if nd < m else g)
            sets.append((filtered, np.concatenate(placed)))
            cands.append((delta, ang))
    costs, first = engine.hausdorff_batch(sets)
    ocosts, ofirst = oracle.refine_select(sets)
    assert np.array_equal(costs, ocosts)
    assert first == ofirst
    assert len(sets[0][0]) == n_frames * m              # the box filter dropped exactly the clutter
    assert cands[first][0] == 0 and abs(cands[first][1]) < 1e-9 and costs[first] < 0.1


def test_hausdorff_batch_large_and_swapped_sets(engine, oracle):
    rng = np.random.default_rng(8)
    pairs = [(rng.normal(size=(9000, 2)), rng.normal(size=(300, 2))),     # many rows, few columns
             (rng.normal(size=(300, 2)), rng.normal(size=(9000, 2))),     # columns exceed the LDS budget -> roles swapped
             (rng.normal(size=(50, 2)), rng.normal(size=(60, 2)))]
    costs, first = engine.hausdorff_batch(pairs)
    ref = np.array([oracle.hausdorff(a, b) for a, b in pairs])
    assert np.array_equal(costs, ref) and first == int(np.argmin(ref))


def test_hausdorff_both_sets_beyond_lds_budget_streaming_kernel(engine, oracle):
    """Both sets larger than the LDS budget (CCTA-cloud sized, align_algorithms.rs:400-431): the
    streaming kernel splits the rows of one pair over many workgroups and chunks the columns."""
    rng = np.random.default_rng(12)
    pairs = [(rng.normal(4.5, 2.0, size=(5000, 2)), rng.normal(4.6, 2.1, size=(4100, 2))),
             (rng.normal(size=(12345, 2)), rng.normal(size=(10001, 2)) * 1.1),
             (rng.normal(size=(40, 2)), rng.normal(size=(33, 2)))]
    costs, first = engine.hausdorff_batch(pairs)
    ref = np.array([oracle.hausdorff(a, b) for a, b in pairs])
    assert np.array_equal(costs, ref) and first == int(np.argmin(ref))
    assert engine.hausdorff(pairs[1][1], pairs[1][0]) == ref[1]


# ---------------------------------------------------------------------------------------
# MM_PRECISION_F32_FAST: expanded-form screening kernel, same exactness contract
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,step,rng_deg", [(6, 1.0, 30.0), (100, 1.0, 180.0), (208, 0.5, 90.0), (320, 1.0, 180.0),
                                            (416, 2.0, 180.0), (521, 0.5, 180.0), (528, 1.0, 90.0), (600, 2.0, 90.0)])
def test_fast_screen_winner_bit_exact_and_interval(engine, oracle, mm, n, step, rng_deg):
    rng = np.random.default_rng(3000 + n)
    ref = blob(rng, n)
    tgt = blob(rng, n) + rng.normal(0, 0.02, (n, 2))
    c = ref.mean(axis=0)
    centre = (float(c[0]), float(c[1]))
    angles, _, _ = mm.search_angles(step, rng_deg)
    bi, ba, bc, costs = engine.best_rotation(ref, tgt, angles, centre, skip_zero=True,
                                             precision=mm.MM_PRECISION_F32_FAST, return_costs=True)
    ocosts = oracle.costs_over_angles(ref, tgt, angles, centre[0], centre[1])
    obi = int(np.argmin(ocosts))
    assert bi == obi and ba == angles[obi] and bc == ocosts[obi]
    rho = np.hypot(*(ref - c).T).max() + np.hypot(*(tgt - c).T).max()
    delta = 24 * U24 * rho
    e2 = (8 * U24 * rho * rho) if n <= 528 else 0.0          # > 528 rows: the direct-form kernel is used
    s = costs ** 2
    lo = np.sqrt(np.maximum(s - e2, 0.0)) - delta
    hi = np.sqrt(s + e2) + delta
    assert np.all(lo <= ocosts * (1 + 1e-12)) and np.all(ocosts <= hi * (1 + 1e-12))   # stated tolerance of the fast screen
    assert np.abs(costs - ocosts).max() < 5e-3


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", [(True, 1.0, 180.0, 501), (False, 0.05, 45.0, 200), (True, 0.5, 180.0, 500)])
def test_fast_precision_chain_bit_identical(engine, oracle, mm, mode, bruteforce, step, rng_deg, ss):
    geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((8, 6, 7, 5))]
    ogeoms = [to_oracle(oracle, g) for g in geoms]
    logs, _ = mm.align_within(engine, geoms, step, rng_deg, bruteforce, ss, precision=mm.MM_PRECISION_F32_FAST, mode=mode)
    for g, og, lg in zip(geoms, ogeoms, logs):
        assert lg == oracle.align_within_chain(og, step, rng_deg, bruteforce, ss, n_threads=8)
        assert geoms_equal(g, og)


def test_fast_precision_ties_and_identical_sets(engine, oracle, mm):
    """Cost ~ 0 (identical sets) is where the expanded form is least accurate: d^2 of order 1e-5 can
    round below zero.  Winners must still be the reference's."""
    t = np.arange(360) * (2 * math.pi / 360)
    ref = np.stack([4.5 + 2 * np.cos(t) * (1 + 0.2 * np.cos(3 * t)), 4.5 + 2 * np.sin(t)], 1)
    angles, _, _ = mm.search_angles(1.0, 180.0)
    for tgt in (ref.copy(), np.roll(ref, 7, axis=0)):
        oc = oracle.costs_over_angles(ref, tgt, angles, 4.5, 4.5)
        out = engine.best_rotation(ref, tgt, angles, (4.5, 4.5), precision=mm.MM_PRECISION_F32_FAST)
        assert out[0] == int(np.argmin(oc)) and out[2] == oc.min()
    circ = np.stack([4.5 + 2 * np.cos(t), 4.5 + 2 * np.sin(t)], 1)
    oc = oracle.costs_over_angles(circ, circ, angles, 4.5, 4.5)
    out = engine.best_rotation(circ, circ, angles, (4.5, 4.5), precision=mm.MM_PRECISION_F32_FAST)
    assert out[0] == int(np.argmin(oc)) and out[2] == oc.min()


@pytest.mark.parametrize("precision", [0, 1, 2, 3])
def test_near_ties_between_neighbouring_candidates(engine, oracle, mm, precision):
    """Adversarial for the screen-then-exact scheme: the true rotation sits (almost) exactly between
    two grid candidates, so their costs differ by ~1e-9 .. 1e-5 -- far below the f32 screening error.
    Every precision mode must return the oracle's first minimum and its exact cost."""
    rng = np.random.default_rng(77)
    angles, _, _ = mm.search_angles(0.5, 180.0)
    refs, tgts, cs = [], [], []
    for p in range(48):
        ref = blob(rng, 300 + 7 * p)
        k = int(rng.integers(5, 700))
        eps = [0.0, 1e-9, -1e-9, 1e-6, -1e-6, 1e-4][p % 6]
        th = -(0.5 * (angles[k] + angles[k + 1]) + eps)
        c, s_ = math.cos(th), math.sin(th)
        rel = ref - 4.5
        tgt = np.stack([rel[:, 0] * c - rel[:, 1] * s_, rel[:, 0] * s_ + rel[:, 1] * c], 1) + 4.5
        refs.append(ref); tgts.append(tgt); cs.append((4.5, 4.5))
    batch = mm.Batch(refs, tgts, [angles] * len(refs), cs, [1] * len(refs))
    out = engine.best_rotation_batch(batch, precision=precision)
    n_close = 0
    for p, (r, t) in enumerate(zip(refs, tgts)):
        oc = oracle.costs_over_angles(r, t, angles, 4.5, 4.5)
        k = int(np.argmin(oc))
        assert out["best_idx"][p] == k and out["best_cost"][p] == oc[k]
        srt = np.sort(oc)
        n_close += int(srt[1] - srt[0] < 1e-6)
    assert n_close >= 8     # the scenario really produces near-ties


# ---------------------------------------------------------------------------------------
# MM_PRECISION_F32_BOUNDED: a lower bound rules candidates out before the screen; winners and costs
# must still be the oracle's, bit for bit
# ---------------------------------------------------------------------------------------
@pytest.fixture(autouse=True)
def _bound_rounds_on_every_batch(engine):
    """The bound rounds are skipped on small batches by default; here every batch must take them."""
    engine.set_bound_min_candidates(0)
    yield
    engine.set_bound_min_candidates(16384)


@pytest.mark.parametrize("n,step,rng_deg", [(64, 1.0, 180.0), (100, 1.0, 180.0), (208, 0.5, 90.0), (320, 1.0, 180.0),
                                            (521, 0.5, 180.0), (528, 1.0, 90.0)])
def test_bounded_screen_winner_bit_exact_and_prunes(engine, oracle, mm, n, step, rng_deg):
    rng = np.random.default_rng(4000 + n)
    ref = blob(rng, n)
    tgt = blob(rng, n) + rng.normal(0, 0.02, (n, 2))
    c = ref.mean(axis=0)
    centre = (float(c[0]), float(c[1]))
    angles, _, _ = mm.search_angles(step, rng_deg)
    batch = mm.Batch([ref], [tgt], [angles], [centre], [1])
    out = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_BOUNDED)
    full = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_FAST)
    ocosts = oracle.costs_over_angles(ref, tgt, angles, centre[0], centre[1])
    obi = int(np.argmin(ocosts))
    assert out["best_idx"][0] == obi and out["best_angle"][0] == angles[obi] and out["best_cost"][0] == ocosts[obi]
    # the bound never widens the exact re-score set
    assert 1 <= out["n_rescored"][0] <= full["n_rescored"][0]


def test_bounded_screen_mixed_batch(engine, oracle, mm):
    """Pairs of very different sizes, an empty set, different candidate lists and unordered point sets
    (any subset gives a valid bound; order only affects how tight it is) in one batch."""
    rng = np.random.default_rng(4100)
    a1, _, _ = mm.search_angles(1.0, 180.0)
    a2, _, _ = mm.search_angles(0.5, 45.0)
    refs, tgts, angs, cs = [], [], [], []
    for n, m, ang, shuffle in [(521, 521, a1, False), (70, 400, a2, False), (400, 70, a1, False), (300, 300, a2, True),
                               (128, 0, a1, False), (9, 12, a2, False), (528, 100, a1, True), (5, 300, a2, False),
                               (300, 3, a1, False), (1, 1, a2, False)]:
        r = blob(rng, n)
        t = blob(rng, m) if m else np.zeros((0, 2))
        if shuffle:
            rng.shuffle(r); rng.shuffle(t)
        refs.append(r); tgts.append(t); angs.append(ang); cs.append((4.5, 4.5))
    batch = mm.Batch(refs, tgts, angs, cs, [1] * len(refs))
    out = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_BOUNDED)
    for p, (r, t, ang) in enumerate(zip(refs, tgts, angs)):
        oc = oracle.costs_over_angles(r, t, ang, 4.5, 4.5)
        k = int(np.argmin(oc))
        assert out["best_idx"][p] == k and out["best_cost"][p] == oc[k], p


def test_bounded_screen_with_costs_falls_back_to_full_screen(engine, oracle, mm):
    rng = np.random.default_rng(4200)
    ref, tgt = blob(rng, 200), blob(rng, 200)
    angles, _, _ = mm.search_angles(2.0, 180.0)
    bi, ba, bc, costs = engine.best_rotation(ref, tgt, angles, (4.5, 4.5), precision=mm.MM_PRECISION_F32_BOUNDED,
                                             return_costs=True)
    oc = oracle.costs_over_angles(ref, tgt, angles, 4.5, 4.5)
    assert bi == int(np.argmin(oc)) and bc == oc.min()
    assert np.all(np.isfinite(costs)) and np.abs(costs - oc).max() < 5e-3


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("bruteforce,step,rng_deg,ss", [(True, 1.0, 180.0, 501), (False, 0.05, 45.0, 200), (True, 0.5, 180.0, 500)])
def test_bounded_precision_chain_bit_identical(engine, oracle, mm, mode, bruteforce, step, rng_deg, ss):
    geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((8, 6, 7, 5))]
    ogeoms = [to_oracle(oracle, g) for g in geoms]
    logs, _ = mm.align_within(engine, geoms, step, rng_deg, bruteforce, ss, precision=mm.MM_PRECISION_F32_BOUNDED, mode=mode)
    for g, og, lg in zip(geoms, ogeoms, logs):
        assert lg == oracle.align_within_chain(og, step, rng_deg, bruteforce, ss, n_threads=8)
        assert geoms_equal(g, og)


def test_bounded_precision_ties_and_identical_sets(engine, oracle, mm):
    t = np.arange(360) * (2 * math.pi / 360)
    ref = np.stack([4.5 + 2 * np.cos(t) * (1 + 0.2 * np.cos(3 * t)), 4.5 + 2 * np.sin(t)], 1)
    angles, _, _ = mm.search_angles(1.0, 180.0)
    for tgt in (ref.copy(), np.roll(ref, 7, axis=0)):
        oc = oracle.costs_over_angles(ref, tgt, angles, 4.5, 4.5)
        out = engine.best_rotation(ref, tgt, angles, (4.5, 4.5), precision=mm.MM_PRECISION_F32_BOUNDED)
        assert out[0] == int(np.argmin(oc)) and out[2] == oc.min()
    # a circle against itself: every candidate costs (about) the same, nothing can be ruled out
    circ = np.stack([4.5 + 2 * np.cos(t), 4.5 + 2 * np.sin(t)], 1)
    oc = oracle.costs_over_angles(circ, circ, angles, 4.5, 4.5)
    out = engine.best_rotation(circ, circ, angles, (4.5, 4.5), precision=mm.MM_PRECISION_F32_BOUNDED)
    assert out[0] == int(np.argmin(oc)) and out[2] == oc.min()


@pytest.mark.parametrize("seed", range(10))
def test_bounded_vs_full_screen_randomised(engine, mm, seed):
    """Differential test on random cases (sizes, steps, torsion, near-circular frames that flatten the cost
    curve): the bounded search and the full expanded-form screen -- itself checked against the oracle above --
    must agree on every log entry and every output coordinate."""
    rng = np.random.default_rng(9000 + seed)
    n_frames = int(rng.integers(6, 40))
    n_points = int(rng.choice([64, 120, 200, 333, 501]))
    step = float(rng.choice([0.5, 1.0, 2.0]))
    rng_deg = float(rng.choice([45.0, 90.0, 180.0]))
    ss = int(rng.choice([n_points, max(60, n_points // 2), 501]))
    sigma = float(rng.choice([0.5, 3.0, 25.0]))
    outs = []
    for prec in (mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED):
        geoms = [mm.synthetic_pullback(n_frames, n_points, pullback_id=i, seed=500 + seed, torsion_sigma_deg=sigma)
                 for i in range(2)]
        if seed % 3 == 0:   # flatten frame shapes towards circles: many near-ties
            for g in geoms:
                c = g.centroids[np.repeat(np.arange(g.n_frames), np.diff(g.lumen_off)), :2]
                d = g.lumen[:, :2] - c
                r = np.hypot(d[:, 0], d[:, 1])[:, None]
                g.lumen[:, :2] = c + d * (0.15 + 0.85 * (2.0 / r))
        logs, _ = mm.align_within(engine, geoms, step, rng_deg, True, ss, precision=prec, mode=1)
        outs.append((logs, geoms))
    (la, ga), (lb, gb) = outs
    for x, y in zip(la, lb):
        assert x == y
    for x, y in zip(ga, gb):
        assert np.array_equal(x.lumen, y.lumen) and np.array_equal(x.cath, y.cath) and np.array_equal(x.centroids, y.centroids)


def test_four_concurrent_callers_each_with_its_own_engine(oracle, mm):
    """the library has to serve >= 4 concurrent callers (the reference's crossbeam scopes).  The
    model is one engine per host thread; every entry point selects the engine's device for the calling thread.
    Four threads run searches, chains and between alignments at once and each must get the oracle's result."""
    import threading
    geoms_by_thread, results, errors = {}, {}, []
    angles, _, _ = mm.search_angles(1.0, 180.0)

    def work(t):
        try:
            eng = mm.Engine()
            rng = np.random.default_rng(700 + t)
            out = []
            for rep in range(3):
                ref, tgt = blob(rng, 200 + 40 * t), blob(rng, 180 + 30 * t)
                prec = (mm.MM_PRECISION_F32, mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED, mm.MM_PRECISION_F64, mm.MM_PRECISION_F32_MATRIX)[(t + rep) % 5]
                out.append((ref, tgt, eng.best_rotation(ref, tgt, angles, (4.5, 4.5), precision=prec)))
            geoms = [mm.synthetic_pullback(6 + t, 160, pullback_id=t), mm.synthetic_pullback(7, 160, pullback_id=t + 1)]
            logs, _ = mm.align_within(eng, geoms, 1.0, 90.0, True, 160, mode=t % 2)
            rot, _ = mm.align_between(eng, [(geoms[0], geoms[1])], 90.0, 1.0, 160)
            results[t] = (out, logs, rot)
            geoms_by_thread[t] = geoms
            eng.close()
        except BaseException as ex:
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        out, logs, rot = results[t]
        for ref, tgt, got in out:
            oc = oracle.costs_over_angles(ref, tgt, angles, 4.5, 4.5)
            assert got[0] == int(np.argmin(oc)) and got[2] == oc.min()
        fresh = [mm.synthetic_pullback(6 + t, 160, pullback_id=t), mm.synthetic_pullback(7, 160, pullback_id=t + 1)]
        og = [to_oracle(oracle, g) for g in fresh]
        for lg, o in zip(logs, og):
            assert lg == oracle.align_within_chain(o, 1.0, 90.0, True, 160, n_threads=2)
        orot = oracle.align_between(og[0], og[1], 90.0, 1.0, 160, n_threads=2)
        assert rot[0] == orot
        for g, o in zip(geoms_by_thread[t], og):
            assert geoms_equal(g, o)


# ---------------------------------------------------------------------------------------
# EXTENSION (not in the reference's 4-phase path): rotation x frame-shift grid
# ---------------------------------------------------------------------------------------
def test_shift_rotation_extension_vs_oracle(engine, oracle, mm):
    geoms = [mm.synthetic_pullback(9, 501, pullback_id=i) for i in range(2)]
    srs = mm.ShiftRotationSearch(engine, geoms, -2, 3, 2.0, 180.0, 501, want_costs=True)
    res = srs.run()
    angles = srs.angles
    assert srs.pose_evals == len(srs.meta) * len(angles)
    best = {}
    for p, (gi, i, sh, j) in enumerate(srs.meta):
        g = geoms[gi]
        ref = mm.search_set(g, int(j), 501) - g.centroids[j, :2]
        tgt = mm.search_set(g, int(i), 501) - g.centroids[i, :2]
        oc = oracle.costs_over_angles(ref, tgt, angles, 0.0, 0.0)
        k = int(np.argmin(oc))
        assert res["best_idx"][p] == k and res["best_cost"][p] == oc[k] and res["best_angle"][p] == angles[k]
        cur = best.get((gi, i))
        if cur is None or oc[k] < cur[2]:
            best[(gi, i)] = (int(sh), float(angles[k]), float(oc[k]))
    assert len(res["winners"]) == len(best)
    for (gi, i, sh, ang, c) in res["winners"]:
        assert best[(gi, i)] == (sh, ang, c)
    # at shift 0 the pair is the reference's chain step (decoupled): same winner as the chain's first step
    og = to_oracle(oracle, geoms[0])
    logs = oracle.align_within_chain(og, 2.0, 180.0, True, 501)
    p0 = [p for p, (gi, i, sh, j) in enumerate(srs.meta) if gi == 0 and i == 1 and sh == 0][0]
    assert math.degrees(res["best_angle"][p0]) == logs[0][2]
    srs.close()


def test_shift_rotation_extension_bounded_equals_full_screen(engine, mm):
    """The extension grid through the bounded screen (winner only, no per-candidate costs): same best
    candidate, cost and winners as the expanded-form screen of every candidate."""
    geoms = [mm.synthetic_pullback(12, 501, pullback_id=i) for i in range(2)]
    out = []
    for prec in (mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED):
        srs = mm.ShiftRotationSearch(engine, geoms, -3, 3, 1.0, 180.0, 501, precision=prec)
        out.append(srs.run())
        srs.close()
    a, b = out
    assert np.array_equal(a["best_idx"], b["best_idx"]) and np.array_equal(a["best_cost"], b["best_cost"])
    assert np.array_equal(a["best_angle"], b["best_angle"]) and a["winners"] == b["winners"]


# ---------------------------------------------------------------------------------------
# MM_PRECISION_F32_MATRIX: the screen on the f16 matrix pipe (k_screen_mx)
# ---------------------------------------------------------------------------------------
def _matrix_search_case(engine, oracle, mm, na, nb, scale, offset, step=1.0, want_kernel="matrix"):
    rng = np.random.default_rng(na * 1000 + nb)
    ref = (blob(rng, na) - 4.5) * scale + np.array(offset)
    tgt = (blob(rng, nb) - 4.5) * scale + np.array(offset)
    c = tgt.mean(axis=0)
    centre = (float(c[0]), float(c[1]))
    angles, _, _ = mm.search_angles(step, 180.0)
    oc = oracle.costs_over_angles(ref, tgt, angles, centre[0], centre[1])
    want = int(np.argmin(oc))
    before = engine.screen_stats()
    bi, ba, bc, costs = engine.best_rotation(ref, tgt, angles, centre, skip_zero=True, precision=mm.MM_PRECISION_F32_MATRIX,
                                             return_costs=True)
    after = engine.screen_stats()
    took = {k: after[k] - before[k] for k in after}
    assert took[want_kernel] == len(angles) and sum(took.values()) == len(angles), took     # the kernel asked for, nothing else
    assert bi == want and ba == angles[want] and bc == oc[want]
    ra, rb = np.sqrt(((ref - c) ** 2).sum(1)).max(), np.sqrt(((tgt - c) ** 2).sum(1)).max()
    rho = ra + rb
    e2 = 2.0 ** -24 * (47 * rho * rho + 6 * ra * ra + 27 * rb * rb)      # mx_e2 (csrc/mm_engine.cpp)
    delta = 24 * 2.0 ** -24 * rho + 2.0 ** -49 * (abs(c).sum() + rho) + 1e-300
    S = np.asarray(costs) ** 2
    lo = np.sqrt(np.maximum(0.0, S - e2)) - delta
    hi = np.sqrt(S + e2) + delta
    exact = np.isclose(costs, oc, rtol=0, atol=0)                 # re-scored candidates carry the exact cost
    assert ((oc >= lo) & (oc <= hi))[~exact].all()
    err2 = np.abs(S - oc ** 2)[~exact]
    assert err2.size and err2.max() < 0.25 * e2                   # the bound is not tight: a factor 4 in hand


@pytest.mark.parametrize("na,nb", [(521, 521), (544, 544), (449, 544), (500, 470), (543, 449)])
@pytest.mark.parametrize("scale,offset", [(1.0, (4.5, 4.5)), (37.0, (-900.0, 120.0)), (1e-3, (0.0, 0.0)), (2.0 ** 20, (1e7, -3e6))])
def test_matrix_screen_winner_cost_and_interval(engine, oracle, mm, na, nb, scale, offset):
    """One search through the matrix-pipe screen at the bench's set sizes (15 .. 17 tiles a side), at coordinate scales far
    from mm: the winner, its angle and its cost are the oracle's (exact re-score), and EVERY candidate's exact cost lies
    in the interval the screened value promises, [sqrt(max(0, S - e2)) - delta, sqrt(S + e2) + delta] with
    e2 = 2^-24 * (47 (rho_a + rho_b)^2 + 6 rho_a^2 + 27 rho_b^2) -- observed errors stay far inside it."""
    _matrix_search_case(engine, oracle, mm, na, nb, scale, offset)


# (reference set, target set): the reference's parameter space -- sample_size and n_points are user kwargs
# (binding/functions.rs:144-167): Rust-side default 200 + 8 catheter points = 208, the OCT benchmark's 216 .. 240 -- every
# column-tile count 2 .. 17 (one instantiation each), odd and even row-tile counts incl. a single loop-free pass (64 rows:
# row tile 0 + the tail), ragged pairs, and above 544 target points the column blocks (600 -> 2 x 10 tiles, 1042 -> 2 x 17,
# 1100 -> 3 x 12, 2048 -> 4 x 16) against 2 .. 64 row tiles
MX_SIZES = [(64, 64), (208, 208), (216, 240), (240, 216), (320, 320), (448, 448), (97, 65), (65, 130), (100, 161), (521, 100),
            (130, 200), (161, 250), (240, 290), (521, 330), (330, 360), (64, 390), (400, 420), (512, 470), (33 * 32, 500),
            (2048, 544), (1042, 521)]
MX_SIZES_BLOCKS = [(600, 600), (1042, 1042), (521, 1100), (64, 2048), (2048, 2048), (545, 545), (100, 577)]


@pytest.mark.parametrize("na,nb", MX_SIZES)
def test_matrix_screen_at_every_tile_count(engine, oracle, mm, na, nb):
    """the matrix-pipe screen covers the reference's parameter space, not the bench size alone."""
    _matrix_search_case(engine, oracle, mm, na, nb, 1.0, (4.5, 4.5), step=2.0 if na * nb > 600000 else 1.0)
    if (na, nb) in ((208, 208), (240, 216), (1042, 521)):
        _matrix_search_case(engine, oracle, mm, na, nb, 2.0 ** 20, (1e7, -3e6), step=2.0)


@pytest.mark.parametrize("na,nb", MX_SIZES_BLOCKS)
def test_matrix_screen_with_the_target_set_in_column_blocks(engine, oracle, mm, na, nb):
    """More than 544 target points: equal blocks of <= 17 column tiles, the row minima carried from block to block through
    the wave's row store (the `carry` form of the asm block)."""
    _matrix_search_case(engine, oracle, mm, na, nb, 1.0, (4.5, 4.5), step=4.0 if na * nb > 600000 else 2.0, want_kernel="matrix_blocks")
    if (na, nb) == (600, 600):
        _matrix_search_case(engine, oracle, mm, na, nb, 1e-3, (0.0, 0.0), step=2.0, want_kernel="matrix_blocks")


def test_matrix_screen_chooses_per_pair(engine, oracle, mm):
    """One batch, pairs of seven shapes: each goes to its own variant of the matrix kernel; a pair with a set of fewer than 64
    points is not screened at all (every candidate scored exactly), one with more than 2048 takes the direct-form screen --
    and no pair sends the others anywhere (`use_mx = false; break`).  Winners and costs are the oracle's."""
    rng = np.random.default_rng(5)
    angles, _, _ = mm.search_angles(2.0, 90.0)
    shapes = [(200, 200), (448, 521), (600, 521), (521, 30), (521, 600), (63, 64), (521, 521), (240, 216), (2049, 300), (200, 200)]
    refs = [blob(rng, a) for a, _ in shapes]
    tgts = [blob(rng, b) for _, b in shapes]
    cs = [t.mean(axis=0) for t in tgts]
    before = engine.screen_stats()
    out = engine.best_rotation_batch(mm.Batch(refs, tgts, [angles] * len(shapes), [(float(c[0]), float(c[1])) for c in cs]),
                                     precision=mm.MM_PRECISION_F32_MATRIX)
    after = engine.screen_stats()
    took = {k: after[k] - before[k] for k in after}
    n = len(angles)
    assert took == {"direct_f32": n, "packed_fma": 0, "matrix": 6 * n, "matrix_blocks": n, "exact_f64": 2 * n}, took
    for p in range(len(shapes)):
        oc = oracle.costs_over_angles(refs[p], tgts[p], angles, float(cs[p][0]), float(cs[p][1]))
        assert out["best_idx"][p] == int(np.argmin(oc)) and out["best_cost"][p] == oc[out["best_idx"][p]]
    # the same without the 2049-point set: no packed-FMA or direct-form launch at all
    keep = [0, 1, 3, 4, 5, 6, 7]
    before = engine.screen_stats()
    engine.best_rotation_batch(mm.Batch([refs[i] for i in keep], [tgts[i] for i in keep], [angles] * len(keep),
                                        [(float(cs[i][0]), float(cs[i][1])) for i in keep]), precision=mm.MM_PRECISION_F32_MATRIX)
    after = engine.screen_stats()
    took = {k: after[k] - before[k] for k in after}
    assert took == {"direct_f32": 0, "packed_fma": 0, "matrix": 4 * n, "matrix_blocks": n, "exact_f64": 2 * n}, took


@pytest.mark.parametrize("matrix", [False, True])
@pytest.mark.parametrize("na,nb", [(64, 64), (223, 223), (521, 521), (300, 521), (521, 97), (528, 528)])
def test_lower_bounds_never_exceed_the_exact_cost(engine, oracle, mm, na, nb, matrix):
    """The bound kernels themselves (test hook mm_lower_bounds: the first round's bound for EVERY candidate of one search),
    packed-FMA (k_screen_lb) and matrix pipe (k_bound_mx): a bound never exceeds the exact squared cost by more than the
    kernel's e2 and delta allow -- whatever subset of points the queries are -- and, with the numpy bound of the same
    queries in hand, it IS that bound up to the same error: a bound that is merely valid (0.0, as the first k_bound_mx
    returned: minima folded behind an MFMA without its wait states) would pass the first check and fail this one."""
    rng = np.random.default_rng(77 + na * 3 + nb)
    ref, tgt = blob(rng, na), blob(rng, nb) + rng.normal(0, 0.03, (nb, 2))
    c = tgt.mean(axis=0)
    ref, tgt = ref - c, tgt - c
    angles, _, _ = mm.search_angles(3.0, 180.0)
    lb, e2, delta, stride = engine.lower_bounds(ref, tgt, angles, (0.0, 0.0), matrix=matrix)
    oc = oracle.costs_over_angles(ref, tgt, angles, 0.0, 0.0)
    assert np.all(np.sqrt(np.maximum(lb - e2, 0.0)) - delta <= oc), "a lower bound above the exact cost"
    qa, want = ref[::stride], []
    for th in angles:
        cs, sn = np.cos(th), np.sin(th)
        rt = np.stack([tgt[:, 0] * cs - tgt[:, 1] * sn, tgt[:, 0] * sn + tgt[:, 1] * cs], 1)
        d1 = ((qa[:, None, :] - rt[None, :, :]) ** 2).sum(2).min(1).max()
        d2 = ((rt[::stride][:, None, :] - ref[None, :, :]) ** 2).sum(2).min(1).max()
        want.append(max(d1, d2))
    rho = np.hypot(ref[:, 0], ref[:, 1]).max() + np.hypot(tgt[:, 0], tgt[:, 1]).max()
    assert np.abs(lb - np.array(want)).max() <= e2 + 2 * delta * rho + 1e-12, "not the bound of its queries"
    assert (lb / np.maximum(oc ** 2, 1e-30)).mean() > 0.5          # every stride-th point: most of H^2 on these noisy blobs (0.74 - 0.99)


@pytest.mark.parametrize("na,nb", [(64, 64), (65, 97), (223, 223), (240, 200), (521, 521), (300, 521), (521, 97), (528, 528), (512, 449)])
def test_first_pick_leaves_row_and_column_minima(engine, oracle, mm, na, nb):
    """The `emit` form of the generated matrix block (k_screen_mx_emit, test hook mm_pick_minima): the squared distance from
    every reference point to its nearest rotated target point and from every target point to its nearest reference point,
    against numpy in f64, within the kernel's error bound e2; their maximum is the screened value and brackets the oracle's
    Hausdorff distance.  (k_lb_topk only ranks them -- a wrong minimum would cost tightness, not validity -- so nothing
    else in the suite would notice.)"""
    rng = np.random.default_rng(5 + 7 * na + nb)
    ref, tgt = blob(rng, na), blob(rng, nb) + rng.normal(0, 0.03, (nb, 2))
    c = tgt.mean(axis=0)
    ref, tgt = ref - c, tgt - c
    for th in (0.0, 0.37, -2.9):
        rows, cols, val, e2 = engine.pick_minima(ref, tgt, th, (0.0, 0.0), skip_zero=False)
        cs, sn = np.cos(th), np.sin(th)
        rt = np.stack([tgt[:, 0] * cs - tgt[:, 1] * sn, tgt[:, 0] * sn + tgt[:, 1] * cs], 1)
        d2 = ((ref[:, None, :] - rt[None, :, :]) ** 2).sum(2)
        assert rows.shape == (na,) and cols.shape == (nb,)
        assert np.abs(rows - d2.min(1)).max() <= e2 + 1e-12, "row minima"
        assert np.abs(cols - d2.min(0)).max() <= e2 + 1e-12, "column minima"
        assert val == max(rows.max(), cols.max())
        h = oracle.costs_over_angles(ref, tgt, np.array([th]), 0.0, 0.0)[0]
        assert abs(val - h * h) <= e2 + 1e-12


def test_matrix_screen_small_batches_fill_the_workgroup(engine, oracle, mm):
    """a single search of 361 candidates used to get one candidate per 256-thread workgroup (three of four waves
    idle); now at least four.  Same result at 1, 3, 4, 5 and 361 candidates."""
    rng = np.random.default_rng(11)
    ref, tgt = blob(rng, 521), blob(rng, 521)
    c = tgt.mean(axis=0)
    for n in (1, 3, 4, 5, 361):
        angles = np.linspace(-0.4, 0.4, n)
        oc = oracle.costs_over_angles(ref, tgt, angles, float(c[0]), float(c[1]))
        bi, ba, bc = engine.best_rotation(ref, tgt, angles, (float(c[0]), float(c[1])), skip_zero=True, precision=mm.MM_PRECISION_F32_MATRIX)
        assert bi == int(np.argmin(oc)) and bc == oc[bi]


@pytest.mark.parametrize("mode", [0, 1])
def test_matrix_screen_chain_equals_oracle(engine, oracle, mm, mode):
    g = [mm.synthetic_pullback(9, 501, pullback_id=i) for i in range(2)]
    og = [to_oracle(oracle, x) for x in g]
    logs, _ = mm.align_within(engine, g, 1.0, 180.0, True, 501, precision=mm.MM_PRECISION_F32_MATRIX, mode=mode)
    for x, o, lg in zip(g, og, logs):
        assert lg == oracle.align_within_chain(o, 1.0, 180.0, True, 501, n_threads=8)
        assert geoms_equal(x, o)


def test_matrix_screen_against_the_packed_fma_screen_on_a_large_batch(engine, oracle, mm):
    """108 300 candidates (300 pairs x 361 rotations, 8 candidates per workgroup): every screened cost of the matrix-pipe
    screen within 1e-4 of the packed-FMA screen's, winners and exact costs identical.  Regression test: the compiler
    once fused the rotation's fma into ONE of the two f32 -> f16 conversions of a coordinate (v_fma_mixlo_f16 against
    v_cvt_pk_f16_f32), so that at f16 ties the hi and lo pieces of a column no longer summed to the coordinate -- 18 of
    these candidates were off by up to 1e-2 while every smaller test passed (mx_split makes the value opaque now)."""
    rng = np.random.default_rng(1)
    P = 300
    refs = [blob(rng, 521) for _ in range(P)]
    tgts = [blob(rng, 521) for _ in range(P)]
    angles, _, _ = mm.search_angles(1.0, 180.0)
    cs = [t.mean(axis=0) for t in tgts]
    batch = mm.Batch(refs, tgts, [angles] * P, [(float(c[0]), float(c[1])) for c in cs])
    a = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_FAST, return_costs=True)
    b = engine.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_MATRIX, return_costs=True)
    assert np.array_equal(a["best_idx"], b["best_idx"]) and np.array_equal(a["best_cost"], b["best_cost"])
    assert np.abs(a["costs"] - b["costs"]).max() < 1e-4
    for p in range(0, P, 37):                                # and the winners are the oracle's
        o = oracle.bruteforce_rotation(refs[p], tgts[p], 1.0, 180.0, float(cs[p][0]), float(cs[p][1]), n_threads=8)
        assert b["best_angle"][p] == o
