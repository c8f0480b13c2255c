"""The reference's own unit tests of the small pieces the path is made of (rows a1-a5 of SURVEY 8 and the frame /
contour helpers the bookkeeping uses), restated by name against the ORACLE and the PRODUCT (C ABI, host only):

  src/types/native/contour_point.rs   test_rotate_point
  src/types/native/frame.rs           test_frame_rotate_with_eem_90deg, test_frame_rotate_around_point,
                                      test_frame_translate_with_eem_and_reference, test_create_catheter_points
  src/types/native/contour.rs         test_build_contour_groups_by_frame, test_build_contour_attaches_measurements_for_lumen,
                                      test_build_contour_ignores_measurements_for_non_lumen, test_downsample_geometry,
                                      test_downsample_edge_cases, test_compute_centroid, test_find_farthest_points,
                                      test_elliptic_ratio_and_area (the ratio; area() is not on the path)

Expected values are the reference's, with its tolerances (1e-6 where it uses one, == where it uses assert_eq)."""
import ctypes as C
import math

import numpy as np
import pytest

import refgeom
from helpers import to_oracle


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


def _frame(mm, lumen, extra=None, ref=None, centroid=(0.0, 0.0, 0.0)):
    g = mm.FlatGeometry.from_frames([np.asarray(lumen, dtype=np.float64)], centroids=[list(centroid)],
                                    ref_points={0: list(ref)} if ref is not None else None)
    if extra is not None:
        g.extra = np.ascontiguousarray(np.asarray(extra, dtype=np.float64))
        g.extra_off = np.array([0, len(extra)], dtype=np.int64)
    return g


def _both(mm, oracle, g, fn_product, fn_oracle):
    """Apply one transform to the product geometry (through the C ABI) and to its oracle copy; both are returned."""
    og = to_oracle(oracle, g)
    s = g.c_struct()
    fn_product(mm._native.lib(), C.byref(s))
    fn_oracle(og)
    return g, og


def test_rotate_point(built, mm, oracle):                                      # contour_point.rs: test_rotate_point
    g, og = _both(mm, oracle, _frame(mm, [[1.0, 0.0, 0.0]]),
                  lambda L, s: L.mm_frame_rotate(s, 0, math.pi / 2.0, 0.0, 0.0),
                  lambda o: oracle.frame_rotate(o, 0, math.pi / 2.0, 0.0, 0.0))
    for p in (g.lumen[0], og.lumen[0]):
        assert abs(p[0] - 0.0) < 1e-6 and abs(p[1] - 1.0) < 1e-6
    assert np.array_equal(g.lumen, og.lumen)


def test_frame_rotate_with_eem_90deg(built, mm, oracle):                       # frame.rs
    lumen = [[0, 2, 0], [2, 4, 0], [4, 2, 0], [2, 0, 0]]
    eem = [[-1, 2, 0], [2, 5, 0], [5, 2, 0], [0, -1, 0]]
    g0 = _frame(mm, lumen, extra=eem, ref=(0.0, 4.0, 0.0), centroid=(1.0, 1.0, 0.0))
    g, og = _both(mm, oracle, g0.copy(),
                  lambda L, s: L.mm_frame_rotate(s, 0, math.pi / 2.0, 1.0, 1.0),
                  lambda o: oracle.frame_rotate(o, 0, math.pi / 2.0, 1.0, 1.0))
    exp_lumen = [(0.0, 0.0), (-2.0, 2.0), (0.0, 4.0), (2.0, 2.0)]
    exp_eem = [(0.0, -1.0), (-3.0, 2.0), (0.0, 5.0), (3.0, 0.0)]
    for x in (g, og):
        assert np.allclose(x.lumen[:, :2], exp_lumen, atol=1e-6) and np.allclose(x.extra[:, :2], exp_eem, atol=1e-6)
        assert np.allclose(np.asarray(x.ref).reshape(-1, 3)[0, :2], (-2.0, 0.0), atol=1e-6)
    assert np.array_equal(g.lumen, og.lumen) and np.array_equal(g.extra, og.extra)
    # and back by -90 degrees about the (rotated) frame centroid: the originals return
    c = g.centroids[0]
    g2, og2 = _both(mm, oracle, g,
                    lambda L, s: L.mm_frame_rotate(s, 0, -math.pi / 2.0, float(c[0]), float(c[1])),
                    lambda o: oracle.frame_rotate(o, 0, -math.pi / 2.0, float(c[0]), float(c[1])))
    for x in (g2, og2):
        assert np.allclose(x.lumen, g0.lumen, atol=1e-6) and np.allclose(x.extra, g0.extra, atol=1e-6)
        assert np.allclose(np.asarray(x.ref).reshape(-1, 3)[0], g0.ref[0], atol=1e-6)


def test_frame_rotate_around_point(built, mm, oracle):                         # frame.rs (rotate_frame_around_point)
    g, og = _both(mm, oracle, _frame(mm, [[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]]),
                  lambda L, s: L.mm_frame_rotate(s, 0, math.pi, 1.0, 1.0),
                  lambda o: oracle.frame_rotate(o, 0, math.pi, 1.0, 1.0))
    exp = [(1.0, 2.0), (2.0, 1.0), (3.0, 2.0), (2.0, 3.0)]
    assert np.allclose(g.lumen[:, :2], exp, atol=1e-6) and np.allclose(og.lumen[:, :2], exp, atol=1e-6)


def test_frame_translate_with_eem_and_reference(built, mm, oracle):            # frame.rs (assert_eq: exact)
    lumen = [[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]]
    eem = [[-1, 2, 0], [2, 5, 0], [5, 2, 0], [0, -1, 0]]
    g = _frame(mm, lumen, extra=eem, ref=(0.5, -0.5, 0.0), centroid=(1.0, 1.0, 0.0))
    g.lumen_centroids, g.has_lumen_centroid = np.array([[1.0, 1.0, 0.0]]), np.ones(1, dtype=np.uint8)
    g, og = _both(mm, oracle, g, lambda L, s: L.mm_frame_translate(s, 0, 1.0, 2.0, 3.0),
                  lambda o: oracle.frame_translate(o, 0, 1.0, 2.0, 3.0))
    for x in (g, og):
        assert x.centroids[0].tolist() == [2.0, 3.0, 3.0]
        assert x.lumen.tolist() == [[1, 2, 3], [3, 2, 3], [3, 4, 3], [1, 4, 3]]
        assert x.extra.tolist() == [[0, 4, 3], [3, 7, 3], [6, 4, 3], [1, 1, 3]]
        assert np.asarray(x.ref).reshape(-1, 3)[0].tolist() == [1.5, 1.5, 3.0]
        assert x.lumen_centroids[0].tolist() == [2.0, 3.0, 3.0]                # translate recomputes Contour.centroid


def test_create_catheter_points(built, mm, tmp_path):                          # frame.rs
    pts = mm.geometry.catheter_points(5.0, (4.5, 4.5), 0.5, 20)
    assert pts.shape == (20, 3) and (pts[:, 2] == 5.0).all()
    assert np.abs(np.hypot(pts[:, 0] - 4.5, pts[:, 1] - 4.5) - 0.5).max() < 1e-6
    # and as the builder makes them: one frame with one point at z = 5 (lenient: no reference frame games here)
    d = mm.InputData(lumen=np.array([[1.0, 0.0, 0.0, 5.0]]), ref_point=np.array([1.0, 0.0, 0.0, 5.0]), diastole=True, label="c")
    g = mm.build_geometry_from_inputdata(d, n_points=20)
    c = g.frame_cath(0)
    assert c.shape == (20, 3) and (c[:, 2] == 5.0).all() and list(g.orig_frames) == [1]
    assert np.abs(np.hypot(c[:, 0] - 4.5, c[:, 1] - 4.5) - 0.5).max() < 1e-6


def test_build_contour_groups_by_frame(built, mm):                             # contour.rs
    rows = np.array([[1, 0.0, 0, 0], [1, 1.0, 0, 0], [2, 2.0, 0, 0]])
    d = mm.InputData(lumen=rows, ref_point=np.array([1.0, 0, 0, 0]), diastole=True, label="c")
    g = mm.build_geometry_from_inputdata(d, n_points=0, check_integrity=False)  # two frames of 2 and 1 points
    by_orig = {int(o): g.frame_lumen(i).shape[0] for i, o in enumerate(g.orig_frames)}
    assert g.n_frames == 2 and by_orig == {1: 2, 2: 1}
    assert sorted(int(i) for i in g.ids) == [0, 1]


@pytest.mark.parametrize("kind", ["lumen", "eem"])
def test_build_contour_measurements(built, mm, kind):                          # contour.rs: attaches / ignores measurements
    rows = np.array([[1, 0.0, 0.0, 0.0]])
    rec = [mm.Record(1, "systolic", 1.23, 4.56)]
    d = mm.InputData(lumen=rows, ref_point=np.array([1.0, 0, 0, 0]), diastole=True, label="c", record=rec,
                     eem=rows.copy() if kind == "eem" else None)
    g = mm.build_geometry_from_inputdata(d, n_points=0)
    assert g.meta["aortic_thickness"] == [1.23] and g.meta["pulmonary_thickness"] == [4.56]   # the LUMEN carries them
    if kind == "eem":
        assert int(g.meta["extra_counts"]["eem"][0]) == 1                      # the EEM contour is there, without any


def test_downsample_geometry_and_edge_cases(built, mm, oracle):                # contour.rs
    pts = np.array([[p[0], p[1], 0.0] for p in refgeom.dummy_frames()[0].pts])  # 6 points, point_index = position
    idx = lambda out: [int(np.argmin(np.abs(pts[:, :2] - q[:2]).sum(axis=1))) for q in out]
    assert idx(oracle.downsample(pts, 3))[:2] == [0, 2] and len(oracle.downsample(pts, 3)) == 3
    assert idx(oracle.downsample(pts, 6))[:2] == [0, 1] and len(oracle.downsample(pts, 6)) == 6
    assert idx(oracle.downsample(pts, 5))[-1] == 4
    two = np.array([[1.0, 2.0, 0.0], [3.0, 4.0, 0.0]])
    assert len(oracle.downsample(two, 5)) == 2 and len(oracle.downsample(two, 2)) == 2
    assert len(oracle.downsample(two, 0)) == 0 and len(oracle.downsample(np.zeros((0, 3)), 3)) == 0
    # the product's strided subset (k_build_sets on the device, mm_catheter_lumen_vec on the host) is the same rule
    g = mm.FlatGeometry.from_frames([pts], catheters=[np.zeros((0, 3))])
    for n in (3, 5, 6):
        got = mm.search_set(g, 0, n)
        assert np.array_equal(got, oracle.downsample(pts, n)[:, :2])


def test_compute_centroid(built, mm):                                          # contour.rs (assert_eq)
    sq = np.array([[0.0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]])
    c = mm._native.contour_centroids(sq, np.array([0, 4], dtype=np.int64))
    assert c.tolist() == [[1.0, 1.0, 0.0]]


def test_find_farthest_points_and_elliptic_ratio(built, mm):                   # contour.rs
    from multimoda_rs_amd import api
    sq = np.array([[0.0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]])
    (i, j), d = api._find_farthest_points(sq)
    assert abs(d - math.sqrt(8.0)) < 1e-6 and {i, j} == {0, 2}
    rhomb = np.array([[1.0, 0, 0], [0, 2, 0], [1, 4, 0], [2, 2, 0]])
    assert abs(api._elliptic_ratio(rhomb) - 2.0) < 1e-6
    # the native post-steps use the same ratio for is_anomalous_coronary (> 2.0 -> anomalous, align_within.rs:249-254):
    # a 4.2 : 2 rhombus is anomalous, the 4 : 2 one is not
    from multimoda_rs_amd import native_frames as NF
    for height, expect in ((4.0, False), (4.2, True)):
        lum = np.array([[1.0, 0, 0], [0, height / 2, 0], [1, height, 0], [2, height / 2, 0]])
        frames = [lum + [0, 0, float(k)] for k in range(3)]
        g = mm.FlatGeometry.from_frames(frames, ref_points={0: [2.0, height / 2, 0.0]})
        zero = np.zeros(3, dtype=np.int64)
        g.meta.update(extra_counts={k: zero.copy() for k in ("eem", "calcification", "sidebranch", "wall")},
                      aortic_thickness=[None] * 3, pulmonary_thickness=[None] * 3)
        _, anomalous = NF.finish_within(g, 0, False)
        assert anomalous is expect
