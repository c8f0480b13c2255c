"""Pin the centerline-placement oracle (oracle/mm_oracle_cl.c) against the known-answer tests the
reference holds for this path.  Each test cites the reference test whose EXPECTATION it restates
(paths relative to the reference checkout).  CPU tests: they check the checker.

Placement parity is loosely pinned: the reference's cases are identity / 90-degree / straight-line
configurations with 1e-6..1e-12 tolerances (see mm_oracle_cl.h).
"""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ocl(oracle):
    from oracle import oracle_cl
    oracle_cl.lib()
    return oracle_cl


def clp(ocl, x, y, z, tangent=(0.0, 0.0, 1.0)):
    return ocl.make_centerline([[x, y, z]], [tangent])[0]


# ---- align_algorithms.rs:573-628 FrameTransformation::apply_to_point -------------------------
def test_apply_to_point_translation_only(ocl):               # :573-600
    out = ocl.tf_apply((1.0, 2.0, 3.0), np.eye(3), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    assert out.tolist() == [2.0, 3.0, 4.0]


def test_apply_to_point_with_rotation(ocl):                  # :602-628
    r = ocl.rotation_from_axis_angle((0.0, 0.0, 1.0), math.pi / 2)
    out = ocl.tf_apply((0.0, 0.0, 0.0), r, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0))
    assert np.allclose(out, [0.0, 1.0, 0.0], atol=1e-12)


def test_align_frame_square(ocl):                            # :630-684
    sq = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float64)
    t, r, pivot = ocl.align_frame(sq, None, clp(ocl, 10.0, 10.0, 10.0))
    assert np.allclose(t, [10.0, 10.0, 10.0], atol=1e-12)
    assert np.allclose(pivot, [10.0, 10.0, 10.0], atol=1e-12)
    assert np.allclose(r, np.eye(3), atol=1e-12)             # CCW square in XY: normal already +z


def test_apply_transformation_to_contour(ocl, oracle):       # :686-733
    g = oracle.OracleGeometry.from_frames([np.array([[0.0, 0, 0], [1.0, 0, 0]])], centroids=[(0.5, 0.0, 0.0)])
    g.has_lumen_centroid = np.ones(1, dtype=np.uint8)
    g.lumen_centroids = np.array([[0.5, 0.0, 0.0]])
    # a centerline point such that align_frame yields translation (2,3,4), identity rotation
    cl = ocl.make_centerline([[2.5, 3.0, 4.0]], [[0.0, 0.0, 1.0]])
    n = ocl.apply_transformations([g], cl, (2.5, 3.0, 4.0))
    assert n == 1
    assert np.allclose(g.lumen, [[2.0, 3.0, 4.0], [3.0, 3.0, 4.0]], atol=1e-12)
    assert np.allclose(g.lumen_centroids[0], [2.5, 3.0, 4.0], atol=1e-12)
    assert np.allclose(g.centroids[0], [2.5, 3.0, 4.0], atol=1e-12)   # :532 frame.centroid = lumen.centroid


def test_calculate_normal_unit(ocl):                         # :735-775
    n = ocl.newell_normal([[0, 0, 0], [1, 0, 0], [0, 1, 0]], (0.0, 0.0, 0.0))
    assert abs(np.linalg.norm(n) - 1.0) < 1e-12
    assert np.allclose(n, [0, 0, 1])
    assert ocl.newell_normal([[0, 0, 0], [1, 0, 0]], (0.0, 0.0, 0.0)).tolist() == [0.0, 0.0, 1.0]  # :207-209


def test_rotate_contour_around_centroid(ocl):                # :777-826
    pts = np.array([[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
    out = ocl.rotate_contour_around_centroid(pts, (0.0, 0.0, 0.0), math.pi / 2)
    assert np.allclose(out[0], [0.0, 1.0, 0.0], atol=1e-6)


def test_get_transformations_one_frame(ocl, oracle):         # :828-882
    g = oracle.OracleGeometry.from_frames([np.array([[0.0, 0, 0], [1.0, 0, 0]])], centroids=[(0.5, 0.0, 0.0)])
    cl = ocl.make_centerline([[10.0, 10, 10], [11.0, 10, 10]], [[0, 0, 1.0], [0, 0, 1.0]])
    assert ocl.apply_transformations([g], cl, (10.0, 10.0, 10.0)) == 1


def test_get_transformations_skips_out_of_range(ocl, oracle):  # :112-123: frames past the centerline end stay put
    fr = [np.array([[0.0, 0, z], [1.0, 0, z], [0.0, 1, z]]) for z in (0.0, 1.0, 2.0)]
    g = oracle.OracleGeometry.from_frames(fr)
    before = g.lumen.copy()
    cl = ocl.make_centerline([[5.0, 5, 5], [5.0, 5, 4]], [[0, 0, 1.0], [0, 0, 1.0]])
    assert ocl.apply_transformations([g], cl, (5.0, 5.0, 5.0)) == 2
    assert np.array_equal(g.lumen[6:], before[6:])
    assert not np.array_equal(g.lumen[:6], before[:6])


def test_best_rotation_three_point_simple_case(ocl):         # :884-934
    ang = np.arange(8) * (math.pi / 4)
    pts = np.stack([np.cos(ang), np.sin(ang), np.zeros(8)], axis=1)
    step = math.pi / 8
    best = ocl.best_rotation_three_point(pts, (0.0, 0.0, 0.0), 0, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0),
                                         (-1.0, 0.0, 0.0), step, clp(ocl, 0.0, 0.0, 0.0))
    assert abs(best) < step + 1e-6


def test_best_rotation_three_point_recovers_known_twist(ocl):
    # same construction, targets taken from the contour twisted by 3 steps: the sweep finds it
    n = 16
    ang = np.arange(n) * (2 * math.pi / n)
    pts = np.stack([1.5 * np.cos(ang), np.sin(ang), np.zeros(n)], axis=1)
    step = 2 * math.pi / n
    tw = ocl.rotate_contour_around_centroid(pts, (0.0, 0.0, 0.0), 3 * step)
    best = ocl.best_rotation_three_point(pts, (0.0, 0.0, 0.0), 4, tw[4], tw[0], tw[n // 2], step,
                                         clp(ocl, 0.0, 0.0, 0.0))
    assert best == pytest.approx(3 * step, abs=1e-12)


# ---- centerline.rs --------------------------------------------------------------------------
def test_cl_find_ref_pt(ocl):                                # centerline.rs:988-1021
    cl = ocl.centerline_from_points([[0, 0, 0], [1, 0, 0], [2, 0, 0]])
    assert ocl.find_ref_idx(cl, (0.0, 0.0, 0.0)) == 0
    assert ocl.find_ref_idx(cl, (1.6, 0.0, 0.0)) == 2
    assert ocl.find_ref_idx(cl, (0.5, 0.0, 0.0)) == 0        # strict `<`: first of two equidistant points


def test_centerline_tangents(ocl):                           # centerline.rs:1210-1243
    cl = ocl.centerline_from_points([[0, 0, 0], [1, 0, 0], [2, 0, 0]])
    for i in range(3):
        assert (cl[i]["tx"], cl[i]["ty"], cl[i]["tz"]) == (1.0, 0.0, 0.0)


# ---- contour.rs:763-831 sort_contour_points ---------------------------------------------------
def test_sort_contour_points(ocl):
    out = ocl.sort_contour_points([[-2, 0, 0], [0, 2, 0], [2, 0, 0], [0, -2, 0]])
    assert np.allclose(out[:, :2], [[0, 2], [-2, 0], [0, -2], [2, 0]], atol=1e-6)


def test_sort_contour_points_tie_takes_last_maximum(ocl):    # Iterator::max_by keeps the last of equal maxima
    out = ocl.sort_contour_points([[-1, 1, 0], [1, 1, 0], [1, -1, 0], [-1, -1, 0]])
    # ascending atan2: (-1,-1) (1,-1) (1,1) (-1,1); maxima y=1 at positions 2,3 -> start at 3
    assert out[:, :2].tolist() == [[-1, 1], [-1, -1], [1, -1], [1, 1]]


# ---- preprocessing.rs:288-604 ---------------------------------------------------------------
def _line_cl(ocl, zs, tangent=(0.0, 0.0, 1.0)):
    return ocl.make_centerline([[0.0, 0.0, z] for z in zs], [tangent] * len(zs))


def _mesh(oracle, centroids):
    fr = [np.array([[c[0], c[1], c[2]], [c[0] + 1, c[1], c[2]], [c[0], c[1] + 1, c[2]]]) for c in centroids]
    return oracle.OracleGeometry.from_frames(fr, centroids=centroids)


def test_ensure_descending_z_and_fallback_spacing(ocl, oracle):   # :288-359, :462-530
    mesh = _mesh(oracle, [(1.0, 2.0, 3.0)])                       # one frame: no centroid spacing (:437-459)
    out, spacing = ocl.preprocess_centerline(_line_cl(ocl, [0.0, 1.0, 2.0, 3.0]), mesh)
    assert spacing == pytest.approx(1.0)                          # total_length / n_segments
    assert out["z"].tolist() == [3.0, 2.0, 1.0, 0.0]              # reversed: first z must be the largest
    out2, _ = ocl.preprocess_centerline(_line_cl(ocl, [3.0, 2.0, 1.0, 0.0]), mesh)
    assert out2["z"].tolist() == [3.0, 2.0, 1.0, 0.0]


def test_mean_spacing_drives_resampling(ocl, oracle):             # :361-435 mean of [5,5] = 5
    mesh = _mesh(oracle, [(0.0, 0.0, 0.0), (3.0, 4.0, 0.0), (6.0, 8.0, 0.0)])
    out, spacing = ocl.preprocess_centerline(_line_cl(ocl, [0.0, 10.0, 20.0]), mesh)
    assert spacing == 5.0
    assert out["z"].tolist() == [20.0, 15.0, 10.0, 5.0, 0.0]


def test_build_samples_and_interpolate(ocl, oracle):              # :532-604
    mesh = _mesh(oracle, [(0.0, 0.0, 0.0), (0.0, 0.0, 0.75)])
    out, spacing = ocl.preprocess_centerline(_line_cl(ocl, [3.0, 2.0, 1.0, 0.0]), mesh)
    assert spacing == 0.75
    assert len(out) == 5 and out["z"][0] == 3.0 and out["z"][-1] == 0.0
    assert out["z"][2] == pytest.approx(1.5, abs=1e-12)           # s = 1.5
    assert out["tz"][2] == pytest.approx(1.0, abs=1e-12)
    assert out["radius"][2] == pytest.approx(0.0, abs=1e-12)


def test_side_branches_are_stripped(ocl, oracle):                 # :23-30
    cl = ocl.make_centerline([[0, 0, 3.0], [0, 0, 2.0], [9, 9, 9.0], [0, 0, 1.0]], [[0, 0, -1.0]] * 4,
                             branch_id=[0, 0, 1, 0])
    out, spacing = ocl.preprocess_centerline(cl, _mesh(oracle, [(0.0, 0.0, 0.0), (0.0, 0.0, 1.0)]))
    assert spacing == 1.0 and out["z"].tolist() == [3.0, 2.0, 1.0]
    only_side = ocl.make_centerline([[0, 0, 1.0]], [[0, 0, 1.0]], branch_id=[2])
    with pytest.raises(RuntimeError, match="no branch-0"):
        ocl.preprocess_centerline(only_side, _mesh(oracle, [(0.0, 0.0, 0.0)]))


# ---- geometry.rs:450-503 rotate_geometry back and forth --------------------------------------
def test_rotate_geometry_back_and_forth(ocl, oracle):
    import refgeom
    g = oracle.OracleGeometry.from_frames(**refgeom.to_arrays(refgeom.dummy_frames()))
    h = g.copy()
    a = math.radians(15.0)
    ocl.rotate_geometry(h, a)
    # independent pure-Python restatement of the same reference lines (tests/refgeom.py)
    fr = refgeom.dummy_frames()
    refgeom.rotate_geometry(fr, a)
    assert np.array_equal(h.lumen, np.concatenate([np.array(f.pts) for f in fr]))
    ocl.rotate_geometry(h, -a)
    assert h.lumen[0].tolist() == g.lumen[0].tolist()             # the reference asserts exact equality (:460-463)
    assert np.allclose(h.lumen, g.lumen, atol=1e-12)
    before = h.lumen.copy()
    ocl.rotate_geometry(h, 0.0)                                   # :242-244 early return, no sort
    assert np.array_equal(h.lumen, before)
