"""GPU parity tests of the CCTA diameter search (include/mm_ccta.h) against the oracle: per-point
nearest-neighbour minima, symmetric distances, region selection and the 41-step searches are
bit-exact (exact f64 minima on the device, sums in index order on the host)."""
import math

import numpy as np
import pytest

from helpers import to_oracle_cl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def occ(oracle):
    from oracle import oracle_ccta
    oracle_ccta.lib()
    return oracle_ccta


@pytest.fixture(scope="module")
def ocl(oracle):
    from oracle import oracle_cl
    oracle_cl.lib()
    return oracle_cl


@pytest.mark.parametrize("na,nb", [(1, 1), (3, 1), (1, 700), (511, 513), (512, 1024), (513, 1025), (2500, 3333),
                                   (20000, 77), (5000, 4500), (4096, 9000)])   # last two: slab order + pruned chunks
def test_nn_min_sq_matches_oracle(engine, mm, occ, na, nb):
    rng = np.random.default_rng(na * 7919 + nb)
    a = rng.normal(0, 4, size=(na, 3)) + [10.0, -190.0, 1700.0]
    b = rng.normal(0, 4, size=(nb, 3)) + [10.0, -190.0, 1700.0]
    if na > 2 and nb > 2:
        b[1] = a[2]                               # a coincident pair: minimum exactly 0
    got = mm.ccta.nn_min_sq(a, b, engine=engine)
    assert np.array_equal(got, occ.nn_min_sq(a, b))
    assert mm.ccta.symmetric_nn_distance(a, b, engine=engine) == occ.symmetric_nn_distance(a, b)


def test_nn_batch_and_empty_sets(engine, mm, occ):
    rng = np.random.default_rng(5)
    sets = [rng.normal(0, 3, size=(n, 3)) for n in (40, 0, 700, 1300)]
    xyz = np.ascontiguousarray(np.concatenate(sets))
    off = np.zeros(len(sets) + 1, dtype=np.int64); off[1:] = np.cumsum([len(s) for s in sets])
    pairs = [(0, 2), (2, 0), (3, 2), (0, 1), (1, 0), (3, 3)]
    q = np.array([p[0] for p in pairs], dtype=np.int32); p_ = np.array([p[1] for p in pairs], dtype=np.int32)
    out_off = np.zeros(len(pairs) + 1, dtype=np.int64); out_off[1:] = np.cumsum([len(sets[a]) for a, _ in pairs])
    out = np.full(int(out_off[-1]), np.nan)
    N = mm._native
    N.check(N.lib().mm_nn_min_sq_batch(engine.handle, len(sets), N._ptr(off), N._ptr(xyz), len(pairs), N._ptr(q),
                                       N._ptr(p_), N._ptr(out_off), N._ptr(out)))
    for k, (a, b) in enumerate(pairs):
        seg = out[out_off[k]:out_off[k + 1]]
        if len(sets[b]) == 0:
            assert np.isinf(seg).all()            # fold over an empty set
        else:
            assert np.array_equal(seg, occ.nn_min_sq(sets[a], sets[b]))
    assert mm.ccta.symmetric_nn_distance(sets[0], sets[1], engine=engine) == math.inf
    assert (out[out_off[5]:out_off[6]] == 0.0).all()          # a set against itself


def test_find_region_points_matches_oracle(engine, mm, occ):
    case = mm.synth.synthetic_tube_case(n_points=3000, n_reference=900, seed=9)
    pts = case["points"].copy()
    pts[17] = pts[5]                                           # exact distance tie: the lower index goes first
    for n_sel in (0, 1, 750, 2999, 3000, 5000):
        s, r = mm.ccta.find_region_points(pts, case["reference"], n_sel, engine=engine)
        os_, or_ = occ.find_region_points(pts, case["reference"], n_sel)
        assert np.array_equal(s, os_) and np.array_equal(r, or_)


def test_aortic_scaling_matches_oracle_and_truth(engine, mm, occ, ocl):
    case = mm.synth.synthetic_tube_case(n_points=4000, n_reference=3500, true_scaling_mm=0.7, seed=4)
    best, d = mm.find_aortic_scaling(case["points"], case["reference"], case["centerline"], engine=engine,
                                     return_distances=True)
    obest, od = occ.aortic_diameter_optimization(case["points"], case["reference"], to_oracle_cl(ocl, case["centerline"]))
    assert best == obest and np.array_equal(d, od)
    assert best == pytest.approx(0.7, abs=1e-12)               # -2.0 + 27 * 0.1
    # empty inputs: f64::MAX comes back, all distances +inf (:75-76, :189-191)
    b2, d2 = mm.find_aortic_scaling(np.zeros((0, 3)), case["reference"], case["centerline"], engine=engine,
                                    return_distances=True)
    assert b2 == np.finfo(np.float64).max and np.isinf(d2).all()


def test_aortic_scaling_large_sets_pruned_path(engine, mm, occ, ocl):
    """Both clouds above the sorting threshold: slab order, device-side scaled clouds, pass A / pass B with
    skipped chunks -- all 41 distances and the winner must still be the oracle's bits.  Also with the points
    shuffled (the staging order is the engine's business, results come back in the caller's order)."""
    case = mm.synth.synthetic_tube_case(n_points=6000, n_reference=5000, true_scaling_mm=-0.4, seed=8)
    ocl_ = to_oracle_cl(ocl, case["centerline"])
    rng = np.random.default_rng(1)
    for pts in (case["points"], case["points"][rng.permutation(len(case["points"]))]):
        best, d = mm.find_aortic_scaling(pts, case["reference"], case["centerline"], engine=engine, return_distances=True)
        obest, od = occ.aortic_diameter_optimization(pts, case["reference"], ocl_)
        assert best == obest and np.array_equal(d, od)
    assert best == pytest.approx(-0.4, abs=1e-12)
    # the per-point minima of one scaled cloud, both directions, against the oracle
    moved = mm.adjust_diameter_centerline_morphing_simple(case["centerline"], case["points"], 1.3)
    assert np.array_equal(mm.ccta.nn_min_sq(moved, case["reference"], engine=engine), occ.nn_min_sq(moved, case["reference"]))
    assert np.array_equal(mm.ccta.nn_min_sq(case["reference"], moved, engine=engine), occ.nn_min_sq(case["reference"], moved))


def test_proximal_distal_scaling_matches_oracle(engine, mm, occ, ocl):
    case = mm.synth.synthetic_tube_case(n_points=6000, n_reference=10, seed=12)
    pts = case["points"]
    z = pts[:, 2]
    order = np.argsort(-z)
    prox_ref = pts[order[:800]] * 1.0
    dist_ref = pts[order[-800:]] * 1.0
    # references: the two ends of the tube pushed outwards by 0.4 / inwards by 0.3 mm
    prox_ref = mm.adjust_diameter_centerline_morphing_simple(case["centerline"], prox_ref, 0.4)
    dist_ref = mm.adjust_diameter_centerline_morphing_simple(case["centerline"], dist_ref, -0.3)
    n_sec = int(math.ceil(0.25 * len(pts)))
    got = mm.find_proximal_distal_scaling(pts, n_sec, n_sec, case["centerline"], prox_ref, dist_ref, engine=engine)
    exp = occ.diameter_optimization(pts, n_sec, n_sec, to_oracle_cl(ocl, case["centerline"]), prox_ref, dist_ref)
    assert got == exp
    assert got[0] > 0.0 > got[1]
    # the wrapper of multimodars/ccta/scaling.py:84-145 on a geometry
    g = mm.synthetic_pullback(8, 100)
    res = {"anomalous_points": g.lumen + [0.05, 0.0, 0.0]}
    cl = mm.Centerline.from_contour_points(np.stack([np.full(12, 4.5), np.full(12, 4.5), np.arange(12) * 0.5 - 1.0], axis=1))
    w = mm.find_distal_and_proximal_scaling(g, cl, res, engine=engine)
    n4 = int(math.ceil(0.25 * g.lumen.shape[0]))
    e = occ.diameter_optimization(res["anomalous_points"], n4, n4, to_oracle_cl(ocl, cl),
                                  g.lumen[:g.lumen_off[2]], g.lumen[g.lumen_off[5]:])
    assert w == e


# ---------------------------------------------------------------------------------------
# find_points_by_cl_region / clean_outlier_points (scale_coronary.rs:263-409) -- oracle parity unpinned (no
# reference test), product against the oracle bit for bit
# ---------------------------------------------------------------------------------------
def _vessel(mm, n_pts, seed, spread=0.3):
    case = mm.synth.synthetic_tube_case(n_points=n_pts, n_reference=10, seed=seed)
    rng = np.random.default_rng(seed)
    pts = case["points"] + rng.normal(0, spread, size=case["points"].shape)     # thicken the shell: neighbours within 1 mm
    return case["centerline"], pts


@pytest.mark.parametrize("n_pts,seed", [(50, 1), (700, 2), (6000, 3), (12000, 4)])     # the last two: slab order + box pruning
def test_find_points_by_cl_region_matches_oracle(engine, mm, occ, n_pts, seed):
    cl, pts = _vessel(mm, n_pts, seed)
    xyz = cl.xyz()
    k0, k1 = len(xyz) // 3, len(xyz) // 3 + 14
    cen = xyz[k0:k1:2] + 0.05                                     # frame centroids along a section of the vessel
    got = mm.find_points_by_cl_region(cl, cen, pts, engine=engine, return_labels=True)
    exp = occ.find_points_by_cl_region(to_oracle_cl(ocl_mod(), cl), cen, pts)
    assert np.array_equal(got[3], exp[3])
    for a, b in zip(got[:3], exp[:3]):
        assert np.array_equal(a, b)
    assert len(got[2]) > 0 and len(got[0]) + len(got[1]) + len(got[2]) == n_pts
    if n_pts >= 700:
        assert (got[3] >= 3).any()                                # the clean-ups moved something


def test_find_points_by_cl_region_single_frame_follows_the_reference(engine, mm, occ):
    """One frame: scale_coronary.rs:268-272 divides 0.0 by 0 -> NaN radius, no centerline point is in
    range, every point is proximal or distal relative to that frame's centroid, then the two clean-ups.  No frames at
    all: the reference panics (usize underflow) -> an error here."""
    cl, pts = _vessel(mm, 900, 5)
    xyz = cl.xyz()
    cen = xyz[len(xyz) // 2: len(xyz) // 2 + 1] + 0.05
    got = mm.find_points_by_cl_region(cl, cen, pts, engine=engine, return_labels=True)
    exp = occ.find_points_by_cl_region(to_oracle_cl(ocl_mod(), cl), cen, pts)
    assert np.array_equal(got[3], exp[3])
    for a, b in zip(got[:3], exp[:3]):
        assert np.array_equal(a, b)
    assert not (got[3] == 2).any()                               # nothing is "between" by the first pass
    assert len(got[0]) + len(got[1]) + len(got[2]) == 900
    with pytest.raises(RuntimeError):
        mm.find_points_by_cl_region(cl, np.zeros((0, 3)), pts, engine=engine)


def ocl_mod():
    from oracle import oracle_cl
    oracle_cl.lib()
    return oracle_cl


@pytest.mark.parametrize("nc,nr,radius,ratio", [(1, 1, 1.0, 0.6), (300, 500, 1.0, 0.6), (5000, 4500, 0.7, 0.5),
                                                (9000, 100, 2.0, 0.9), (0, 10, 1.0, 0.6), (10, 0, 1.0, 0.6)])
def test_clean_outlier_points_matches_oracle(engine, mm, occ, nc, nr, radius, ratio):
    rng = np.random.default_rng(nc * 31 + nr)
    c = rng.normal(0, 3, size=(nc, 3)) + [10.0, -190.0, 1700.0]
    r = rng.normal(0, 3, size=(nr, 3)) + [11.0, -190.0, 1700.0]
    if nc > 3 and nr > 3:
        c[1] = c[0]                                                # duplicates count as neighbours, the point itself does not
        r[2] = c[3] + [radius, 0.0, 0.0]                           # (nearly) on the boundary of the radius
    got = mm.clean_outlier_points(c, r, radius, ratio, engine=engine)
    exp = occ.clean_outlier_points(c, r, radius, ratio)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
