"""The reference's own tests for the host-side bookkeeping around the hot path, restated against
tests/mm_checkers/postproc.py on multimoda_rs_amd.frames (pure host code, no GPU): hole filling
(align_within.rs:889-941), pair post-processing (postprocessing.rs:595-978), integrity check
(io/integrity_check.rs:349-560), walls (processing/wall.rs has no tests of its own: properties), and
the frame-list <-> flat conversions."""
import math

import numpy as np
import pytest

import refgeom


@pytest.fixture(scope="module")
def PP(mm):
    from mm_checkers import postproc
    return postproc


@pytest.fixture(scope="module")
def FR(mm):
    from multimoda_rs_amd import frames
    return frames


def _from_refgeom(FR, rframes):
    out = []
    for f in rframes:
        lum = FR.Contour(f.id, f.orig, np.array(f.pts, dtype=np.float64), tuple(f.lumen_centroid), None, None, "lumen")
        out.append(FR.Frame(f.id, list(f.centroid), lum, {}, None if f.ref is None else np.array(f.ref, dtype=np.float64)))
    return out


# ---- postprocessing.rs test helpers (:478-593) ---------------------------------------------------
def t_contour(FR, cid, z, thickness, kind):
    return FR.Contour(cid, cid, np.array([[1.0, 2.0, z], [3.0, 4.0, z]]), (2.0, 3.0, z), thickness, None, kind)


def t_frame(FR, cid, z, thickness, set_ref):
    return FR.Frame(cid, [2.0, 3.0, z], t_contour(FR, cid, z, thickness, "lumen"), {"eem": t_contour(FR, cid, z, None, "eem")},
                    np.array([0.0, 0.0, z]) if set_ref else None)


def t_geometry(FR, zs, thick=()):
    fr = [t_frame(FR, i, z, thick[i] if i < len(thick) else None, i == len(zs) // 2) for i, z in enumerate(zs)]
    if fr and all(f.reference_point is None for f in fr):
        fr[0].reference_point = np.array([0.0, 0.0, fr[0].centroid[2]])
    return fr


def t_pair(FR):
    return t_geometry(FR, [0.0, 1.0, 2.0, 3.0, 4.0], [1.0] * 5), t_geometry(FR, [0.0, 2.0, 4.0, 6.0, 8.0], [2.0] * 5)


def dummy_custom(FR, z_spacing, n_frames):
    """utils/test_utils.rs:8-110 dummy_geometry_custom."""
    pts = [(1.0, 3.0), (0.0, 2.0), (0.0, 0.0), (1.0, 0.0), (2.0, 0.0), (2.0, 2.0)]
    out = []
    for i in range(n_frames):
        z = i * z_spacing
        c = FR.Contour(i, 999, np.array([[x, y, z] for x, y in pts]), (1.0, 1.0, z), None, None, "lumen")
        c.compute_centroid()
        out.append(FR.Frame(i, list(c.centroid), c, {}, np.array([3.0, 1.0, z]) if i == n_frames // 2 else None))
    return out


# ---- hole filling --------------------------------------------------------------------------------
def test_detect_holes_and_fill_one_frame(PP, FR):                 # align_within.rs:889-919
    fr = _from_refgeom(FR, refgeom.dummy_aligned_long_frames())
    fr[5].translate(0.0, 0.0, 1.0)
    hole, base = PP.detect_holes(fr)
    assert hole and base == pytest.approx(1.0, abs=1e-6)
    mid = PP.fix_one_frame_hole(fr[1], fr[2])
    assert mid.centroid[2] == pytest.approx(1.5, abs=1e-6) and np.allclose(mid.lumen.points[:, 2], 1.5, atol=1e-6)
    assert mid.reference_point is None
    new = PP.fill_holes(fr)
    assert len(new) == 7
    for i, f in enumerate(new):
        assert f.id == i and f.lumen.id == i and f.centroid[2] == float(i) and f.lumen.centroid[2] == float(i)
        assert (f.lumen.points[:, 2] == float(i)).all()


def test_detect_holes_and_fill_two_frame(PP, FR):                 # align_within.rs:921-940
    fr = _from_refgeom(FR, refgeom.dummy_aligned_long_frames())
    fr[5].translate(0.0, 0.0, 2.0)
    new = PP.fill_holes(fr)
    assert len(new) == 8
    for i, f in enumerate(new):
        assert f.id == i and f.lumen.id == i and f.centroid[2] == pytest.approx(float(i), abs=1e-12)
        assert f.lumen.centroid[2] == pytest.approx(float(i), abs=1e-12)
        assert np.allclose(f.lumen.points[:, 2], float(i), atol=1e-12)


def test_fill_large_gap_and_no_hole(PP, FR):                      # :418-445 (ratio >= 3.5) and :379-381
    fr = _from_refgeom(FR, refgeom.dummy_aligned_long_frames())
    same = PP.fill_holes([f.clone() for f in fr])
    assert len(same) == 6
    fr[5].translate(0.0, 0.0, 4.0)                                # dz = 5 baselines -> floor(5 - 1) = 4 frames
    new = PP.fill_holes(fr)
    assert len(new) == 10 and [f.id for f in new] == list(range(10))
    assert np.allclose([f.centroid[2] for f in new], np.arange(10.0), atol=1e-12)


# ---- postprocessing.rs -----------------------------------------------------------------------------
def test_check_same_sample_rate(PP, FR):                          # :595-626
    same, da, db = PP.check_same_sample_rate(t_geometry(FR, [0.0, 1.0, 2.0]), t_geometry(FR, [0.0, 1.0, 2.0]), 0.1)
    assert same and da == 1.0 and db == 1.0
    a, b = t_pair(FR)
    _, da, db = PP.check_same_sample_rate(a, b, 0.1)
    assert da > 0.0 and db > 0.0


def test_get_avg_z_diff(PP, FR):                                  # :628-635
    assert PP.get_avg_z_diff(t_geometry(FR, [0.0, 1.0, 3.0, 6.0])) == 2.0
    assert PP.get_avg_z_diff(t_geometry(FR, [7.0])) == 0.0


def test_resample_by_diff(PP, FR):                                # :637-670
    r = PP.resample_by_diff(t_geometry(FR, [0.0, 2.0, 5.0]), 1.0)
    assert [f.centroid[2] for f in r] == [0.0, 1.0, 2.0]
    assert all((f.lumen.points[:, 2] == f.centroid[2]).all() and f.extras["eem"].centroid[2] == f.centroid[2] for f in r)
    fr = [t_frame(FR, 0, 5.0, None, False), t_frame(FR, 1, 0.0, None, True), t_frame(FR, 2, 2.0, None, False)]
    r = PP.resample_by_diff(fr, 1.0)
    assert [f.centroid[2] for f in r] == [0.0, 1.0, 2.0] and r[0].reference_point is not None


def test_predict_z_positions(PP):                                 # :672-699
    assert PP.predict_z_positions(0.0, 0.0, 5.0, 1.0) == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    z = PP.predict_z_positions(5.0, 0.0, 5.0, 1.0)
    assert z and 5.0 in z
    z = PP.predict_z_positions(2.5, 0.0, 5.0, 1.0)
    assert 2.5 in z and any(v <= 1.0 for v in z) and any(v >= 4.0 for v in z) and z == sorted(z)
    assert PP.predict_z_positions(1.0, 0.0, 5.0, 0.0) == [] and PP.predict_z_positions(1.0, 0.0, 5.0, math.nan) == []


def test_new_frames_by_sample_rate(PP, FR):                       # :701-720
    new = PP.new_frames_by_sample_rate(t_geometry(FR, [0.0, 2.0, 4.0]), [0.0, 1.0, 2.0, 3.0, 4.0])
    assert len(new) == 5
    for i, f in enumerate(new):
        assert f.centroid[2] == float(i) and f.id == i and f.lumen.id == i
        assert (f.lumen.points[:, 2] == float(i)).all() and (f.extras["eem"].points[:, 2] == float(i)).all()
    assert new[1].reference_point is None and new[2].reference_point is not None


def test_blend_contour(PP, FR):                                   # :722-754
    c1 = t_contour(FR, 0, 0.0, None, "lumen")
    c2 = t_contour(FR, 1, 2.0, None, "lumen")
    c2.points[:, :2] = [[5.0, 6.0], [7.0, 8.0]]
    b = PP.blend_contour(c1, c2, 0.5)
    assert b.points[:, :2].tolist() == [[3.0, 4.0], [5.0, 6.0]] and b.centroid == (2.0, 3.0, 1.0)
    assert (b.points[:, 2] == 0.0).all() and b.id == 0


def test_trim_geom_pair(PP, FR):                                  # :756-781
    a, b = PP.trim_pair(t_geometry(FR, [0.0, 1.0, 2.0, 3.0, 4.0]), t_geometry(FR, [0.0, 1.0, 2.0]))
    assert len(a) == 3 and len(b) == 3
    assert [f.id for f in a] == [0, 1, 2] and [f.id for f in b] == [0, 1, 2]
    assert [f.centroid[2] for f in a] == [1.0, 2.0, 3.0]


def test_adjust_walls_anomalous_geom_pair(PP, FR):                # :783-827
    a, b = PP.adjust_walls_anomalous_pair(t_geometry(FR, [0.0, 1.0], [1.0, 2.0]), t_geometry(FR, [0.0, 1.0], [3.0, 4.0]))
    assert [f.lumen.aortic_thickness for f in a] == [2.0, 3.0] and [f.lumen.aortic_thickness for f in b] == [2.0, 3.0]
    assert all("wall" in f.extras for f in a + b)
    a, b = PP.adjust_walls_anomalous_pair(t_geometry(FR, [0.0, 1.0], [1.0, None]), t_geometry(FR, [0.0, 1.0], [None, 4.0]))
    assert a[0].lumen.aortic_thickness == 1.0 and b[1].lumen.aortic_thickness == 4.0


def test_postprocess_geom_pair_runs(PP, FR):                      # :829-917 (the reference accepts Ok or Err)
    for anomalous in (False, True):
        a, b = t_pair(FR)
        ra, rb = PP.postprocess_pair(a, b, 0.1, anomalous)
        assert ra and rb
    with pytest.raises(RuntimeError, match="No reference point"):
        PP.postprocess_pair([], [], 0.1, False)
    s = t_geometry(FR, [0.0], [1.0])
    ra, rb = PP.postprocess_pair(s, [f.clone() for f in s], 0.1, False)
    assert len(ra) == 1 and len(rb) == 1


def test_complex_resampling(PP, FR):                              # :919-977
    ga, gb = dummy_custom(FR, 1.0, 3), dummy_custom(FR, 0.5, 6)
    same, da, db = PP.check_same_sample_rate(ga, gb, 0.1)
    assert not same and da == 1.0 and db == 0.5
    ref_z_b = gb[PP.find_ref_frame_idx(gb)].centroid[2]
    z = PP.predict_z_positions(ref_z_b, 0.0, 2.5, 0.5)
    assert z == [i * 0.5 for i in range(6)]
    assert [f.centroid[2] for f in PP.new_frames_by_sample_rate(ga, z)] == [i * 0.5 for i in range(5)]
    assert [f.centroid[2] for f in PP.resample_by_diff(ga, 0.5)] == [0.0, 0.5, 1.0]
    pa, pb = PP.postprocess_pair(ga, gb, 0.1, True)
    assert len(pa) == len(pb) > 0
    for fa, fb in zip(pa, pb):
        assert fa.id == fb.id and fa.centroid == fb.centroid
        assert np.array_equal(fa.lumen.points, fb.lumen.points)


# ---- walls (wall.rs) -----------------------------------------------------------------------------
def _ring(FR, n, r, z=0.0, thickness=None):
    phi = math.pi / 2 + np.arange(n) * (2 * math.pi / n)         # index 0 = highest y, counter-clockwise
    c = FR.Contour(0, 0, np.stack([4.5 + r * np.cos(phi), 4.5 + r * np.sin(phi), np.full(n, z)], axis=1), None,
                   thickness, None, "lumen")
    c.compute_centroid()
    return c


def test_offset_contour(PP, FR):                                  # wall.rs:52-100
    c = _ring(FR, 64, 2.0)
    w = PP.offset_contour(c, 1.0)
    d = np.linalg.norm(w.points - np.array(c.centroid), axis=1)
    assert np.allclose(d, 3.0, atol=1e-12) and w.kind == "wall" and (w.points[:, 2] == 0.0).all()
    half = PP.offset_contour(c, 1.0, (0, 32))
    d = np.linalg.norm(half.points - np.array(c.centroid), axis=1)
    assert np.allclose(d[:33], 3.0) and np.allclose(d[33:], 2.0)
    p = FR.Contour(0, 0, np.array([[1.0, 1, 0], [1.0, 1, 0]]), None, None, None, "lumen")
    assert np.array_equal(PP.offset_contour(p, 1.0).points, p.points)        # zero-length vectors stay (:76)


def test_create_wall_frames(PP, FR):                              # wall.rs:7-47, 109-213
    lum = _ring(FR, 500, 1.5)
    eem = _ring(FR, 500, 2.5); eem.kind = "eem"
    f = FR.Frame(0, list(lum.centroid), lum, {"eem": eem}, None)
    out = PP.create_wall_frames([f], anomalous=False)[0]
    d = np.linalg.norm(out.extras["wall"].points - np.array(eem.centroid), axis=1)
    assert np.allclose(d, 3.5, atol=1e-9)                         # from the EEM when present and not anomalous
    out = PP.create_wall_frames([f], anomalous=True)[0]
    d = np.linalg.norm(out.extras["wall"].points - np.array(lum.centroid), axis=1)
    assert np.allclose(d, 2.5, atol=1e-9)                         # anomalous: from the lumen
    assert "wall" not in f.extras                                 # input untouched
    # measured aortic thickness: coronary half = lumen + 1 mm, aortic half = rectangle of that thickness
    thick = _ring(FR, 500, 1.5, z=3.0, thickness=0.8)
    PP.assign_aortic([FR.Frame(0, list(thick.centroid), thick, {}, None)])
    w = PP.create_aortic_wall(thick)
    assert len(w) == 500 and w.kind == "wall" and (w.points[250:, 2] == 3.0).all()
    d = np.linalg.norm(w.points[:250] - np.array(thick.centroid), axis=1)
    assert np.allclose(d, 2.5, atol=1e-9)
    outer_x = thick.points[375, 0] + 0.8
    assert w.points[250:, 0].max() == pytest.approx(outer_x) and w.points[250:, 1].max() == pytest.approx(thick.points[0, 1] + 1.0)
    assert w.points[250:, 1].min() == pytest.approx(thick.points[250, 1] - 1.0)
    with pytest.raises(NotImplementedError):
        PP.create_wall_frames([f], False, with_pulmonary=True)


def test_smooth_frames(PP, FR):                                   # geometry.rs:165-239
    fr = []
    for i in range(4):
        lum = _ring(FR, 32, 1.5 + 0.3 * (i % 2), z=float(i))
        eem = _ring(FR, 32, 2.5 + 0.3 * (i % 2), z=float(i)); eem.kind = "eem"
        fr.append(FR.Frame(i, list(lum.centroid), lum, {"eem": eem}, None))
    sm = PP.smooth_frames(fr)
    for i in range(4):
        p, n = fr[max(i - 1, 0)], fr[min(i + 1, 3)]
        for get in (lambda f: f.lumen, lambda f: f.extras["eem"]):
            exp = (get(p).points[:, :2] + get(fr[i]).points[:, :2] + get(n).points[:, :2]) / 3.0
            assert np.array_equal(get(sm[i]).points[:, :2], exp) and (get(sm[i]).points[:, 2] == float(i)).all()
        assert sm[i].centroid == fr[i].centroid                   # the frame centroid is not recomputed
        assert sm[i].lumen.centroid == pytest.approx(tuple(np.mean(sm[i].lumen.points, axis=0)), abs=1e-12)


# ---- integrity check -------------------------------------------------------------------------------
def _iframe(FR, fid, orig, has_ref, z, n=4):
    pts = np.array([[float(k), float(k) * 0.5, z] for k in range(n)])
    lum = FR.Contour(fid, orig, pts, None, None, None, "lumen")
    lum.compute_centroid()
    return FR.Frame(fid, list(lum.centroid), lum, {}, np.array(lum.centroid) if has_ref else None)


def test_integrity_check(PP, FR):                                 # integrity_check.rs:349-560
    ok = [_iframe(FR, 0, 12, False, 0.0), _iframe(FR, 1, 11, True, 1.0), _iframe(FR, 2, 10, False, 2.0)]
    PP.check_geometry_integrity(ok)
    with pytest.raises(RuntimeError, match="no frames"):
        PP.check_geometry_integrity([])
    with pytest.raises(RuntimeError, match="consecutive"):
        PP.check_geometry_integrity([_iframe(FR, 0, 10, False, 0.0), _iframe(FR, 2, 11, False, 1.0)])
    bad = _iframe(FR, 0, 10, True, 0.0)
    bad.lumen.points = bad.lumen.points[:0]
    bad.lumen.centroid = tuple(bad.centroid)
    with pytest.raises(RuntimeError, match="no points"):
        PP.check_geometry_integrity([bad])
    with pytest.raises(RuntimeError, match="exactly one reference point"):
        PP.check_geometry_integrity([_iframe(FR, 0, 11, True, 0.0), _iframe(FR, 1, 10, True, 1.0)])
    with pytest.raises(RuntimeError, match="Lumen point count mismatch"):
        PP.check_geometry_integrity([_iframe(FR, 0, 11, True, 0.0), _iframe(FR, 1, 10, False, 1.0, n=5)])
    ex = [_iframe(FR, 0, 11, True, 0.0), _iframe(FR, 1, 10, False, 1.0)]
    ex[0].extras["eem"] = FR.Contour(0, 11, np.zeros((3, 3)), None, None, None, "eem")
    ex[1].extras["eem"] = FR.Contour(1, 10, np.zeros((4, 3)), None, None, None, "eem")
    with pytest.raises(RuntimeError, match="Eem contour point count mismatch"):
        PP.check_geometry_integrity(ex)
    ex[1].extras["eem"] = FR.Contour(1, 99, np.zeros((3, 3)), None, None, None, "eem")
    with pytest.raises(RuntimeError, match="Original frame mismatch"):
        PP.check_geometry_integrity(ex)
    with pytest.raises(RuntimeError, match="higher z-coords|Proximal end index"):
        PP.check_geometry_integrity([_iframe(FR, 0, 11, True, 2.0), _iframe(FR, 1, 10, False, 1.0)])


# ---- conversions -----------------------------------------------------------------------------------
def test_flat_round_trip(mm, FR):
    g = mm.synthetic_pullback(5, 40, pullback_id=3)
    mm.centerline.with_lumen_centroids(g)
    g.meta["aortic_thickness"] = [None, 0.5, None, 0.7, None]
    fr = FR.to_frames(g)
    assert [f.lumen.aortic_thickness for f in fr] == [None, 0.5, None, 0.7, None] and "catheter" in fr[0].extras
    h = FR.from_frames(fr, g.label, g.meta)
    for name in ("ids", "lumen_ids", "orig_frames", "centroids", "lumen_off", "lumen", "cath_off", "cath", "has_ref", "ref",
                 "lumen_centroids"):
        assert np.array_equal(getattr(g, name), getattr(h, name)), name
    assert h.extra is None and h.meta["aortic_thickness"] == g.meta["aortic_thickness"]
    from mm_checkers import postproc as PP
    PP.assign_aortic(fr)
    w = FR.from_frames(PP.create_wall_frames(fr, True), g.label, g.meta)
    assert w.meta["extra_counts"]["wall"].tolist() == [40] * 5 and w.extra.shape == (200, 3)
    assert w.meta["lumen_aortic"].sum() == 5 * 20 and w.meta["wall_aortic"].sum() == 5 * 20
    back = FR.to_frames(w)
    assert all(np.array_equal(a.extras["wall"].points, b.extras["wall"].points) and
               np.array_equal(a.extras["wall"].aortic, b.extras["wall"].aortic)
               for a, b in zip(back, PP.create_wall_frames(fr, True)))


# ---- wall twist compensation (centerline_align/align.rs:381-595; the reference holds no test) ------
def test_align_walls_untwists(PP, FR):
    """Frames on a straight vessel whose wall contours are twisted by known angles about the lumen
    normal: align_walls rotates every wall back onto frame 0's aortic direction; lumens stay."""
    n = 80
    base = _ring(FR, n, 1.5)
    twists = [0.0, 0.35, -0.5, 1.2, 2.9]
    fr = []
    for i, a in enumerate(twists):
        lum = base.clone()
        lum.points[:, 2] = float(i)
        lum.compute_centroid()
        f = FR.Frame(i, list(lum.centroid), lum, {}, None)
        PP.assign_aortic([f])
        f = PP.create_wall_frames([f], True)[0]
        w = f.extras["wall"]
        c, s = math.cos(a), math.sin(a)
        dx, dy = w.points[:, 0] - f.centroid[0], w.points[:, 1] - f.centroid[1]
        w.points[:, 0], w.points[:, 1] = f.centroid[0] + dx * c - dy * s, f.centroid[1] + dx * s + dy * c
        fr.append(f)
    lumens = [f.lumen.points.copy() for f in fr]
    ref_wall = fr[0].extras["wall"].points.copy()
    PP.align_walls(fr, True)
    for i, f in enumerate(fr):
        assert np.array_equal(f.lumen.points, lumens[i])
        assert np.allclose(f.extras["wall"].points[:, :2], ref_wall[:, :2], atol=1e-9), i
        assert np.allclose(f.extras["wall"].points[:, 2], float(i))
    before = [f.extras["wall"].points.copy() for f in fr]
    PP.align_walls(fr, False)                                     # anomalous = false: nothing moves (:590)
    assert all(np.array_equal(f.extras["wall"].points, b) for f, b in zip(fr, before))
    # without aortic flags the major axis is used, sign-ambiguous: the smaller rotation wins (:556-565)
    el = []
    for i, a in enumerate([0.0, 0.4, math.pi - 0.3]):
        phi = np.arange(n) * (2 * math.pi / n)
        pts = np.stack([3.0 * np.cos(phi), 1.0 * np.sin(phi), np.full(n, float(i))], axis=1)
        c, s = math.cos(a), math.sin(a)
        wpts = np.stack([pts[:, 0] * c - pts[:, 1] * s, pts[:, 0] * s + pts[:, 1] * c, pts[:, 2]], axis=1)
        lum = FR.Contour(i, i, pts * [0.5, 0.5, 1.0], None, None, None, "lumen"); lum.compute_centroid()
        f = FR.Frame(i, [0.0, 0.0, float(i)], lum, {"wall": FR.Contour(i, i, wpts, None, None, None, "wall")}, None)
        el.append(f)
    PP.align_walls(el, True)
    for f in el[1:]:
        ax = f.extras["wall"].points[0, :2] - f.extras["wall"].points[n // 2, :2]
        assert abs(ax[1]) < 1e-9 and abs(abs(ax[0]) - 6.0) < 1e-9      # major axis back on x (either sign)


# ---- the batched fast path of api._finish_within equals the frame-by-frame one -------------------
@pytest.mark.parametrize("with_eem,anomalous,smooth,thick", [
    (False, False, True, ()), (True, False, True, ()), (True, True, True, ()), (False, True, False, (1, 4)),
    (True, True, True, (0, 2, 5)), (False, False, False, ()),
])
def test_finish_within_batched_equals_frame_model(mm, PP, FR, with_eem, anomalous, smooth, thick):
    from mm_checkers import api_python
    from multimoda_rs_amd import api
    from multimoda_rs_amd.io import EXTRA_KINDS
    F, m = 6, 48
    g = mm.synthetic_pullback(F, m, pullback_id=1, seed=5)
    if with_eem:
        L = g.lumen.reshape(F, m, 3)
        c = L.mean(axis=1, keepdims=True)
        g.extra = np.ascontiguousarray((c + (L - c) * [1.5, 1.5, 1.0]).reshape(-1, 3))
        g.extra_off = np.arange(F + 1, dtype=np.int64) * m
    g.meta["extra_counts"] = {k: (np.full(F, m, dtype=np.int64) if (k == "eem" and with_eem) else np.zeros(F, dtype=np.int64))
                              for k in EXTRA_KINDS}
    g.meta["aortic_thickness"] = [0.8 + 0.1 * i if i in thick else None for i in range(F)]
    g.meta["pulmonary_thickness"] = [None] * F
    a, b = g.copy(), g.copy()
    assert api_python.finish_within_batched(a, anomalous, smooth) is True
    mm.centerline.with_lumen_centroids(b)
    fr = FR.to_frames(b)
    if anomalous:
        PP.assign_aortic(fr)
    fr = PP.create_wall_frames(fr, anomalous, False)
    if smooth:
        fr = PP.smooth_frames(fr)
    ref = FR.from_frames(fr, b.label, b.meta)
    for name in ("lumen", "lumen_off", "extra", "extra_off", "cath", "centroids", "ref", "lumen_centroids"):
        assert np.array_equal(getattr(a, name), getattr(ref, name)), name
    for k in EXTRA_KINDS:
        assert np.array_equal(a.meta["extra_counts"][k], ref.meta["extra_counts"][k]), k
    for k in ("lumen_aortic", "wall_aortic"):
        assert (k in a.meta) == (k in ref.meta) and (k not in a.meta or np.array_equal(a.meta[k], ref.meta[k])), k
    # irregular geometries fall back
    c = g.copy()
    c.meta["extra_counts"]["calcification"] = np.ones(F, dtype=np.int64)
    assert api_python.finish_within_batched(c, anomalous, smooth) is False


# ---- the flat fast path of postprocess_geom_pair equals the frame-list one ---------------------------
def _regular_geom(mm, F, m, seed, z0, dz, ref_at, with_eem, with_wall, thick, roll=0, lc=True):
    from multimoda_rs_amd.io import EXTRA_KINDS
    rng = np.random.default_rng(seed)
    g = mm.synthetic_pullback(F, m, pullback_id=seed % 4, seed=seed)
    z = z0 + dz * np.arange(F)
    if roll:
        z = np.roll(z, roll)
    g.lumen[:, 2] = np.repeat(z, m); g.cath[:, 2] = np.repeat(z, 20); g.centroids[:, 2] = z
    g.has_ref[:] = 0; g.has_ref[ref_at] = 1; g.ref[:] = 0.0; g.ref[ref_at] = [7.0, 4.0, z[ref_at]]
    blobs, counts = [], {k: np.zeros(F, dtype=np.int64) for k in EXTRA_KINDS}
    L = g.lumen.reshape(F, m, 3)
    if with_eem:
        blobs.append(L * [1.3, 1.3, 1.0]); counts["eem"][:] = m
    if with_wall:
        blobs.append(L * [1.6, 1.6, 1.0]); counts["wall"][:] = m
    if blobs:
        X = np.concatenate(blobs, axis=1)
        g.extra = np.ascontiguousarray(X.reshape(-1, 3)); g.extra_off = np.arange(F + 1, dtype=np.int64) * X.shape[1]
    g.meta["extra_counts"] = counts
    g.meta["aortic_thickness"] = [0.7 + 0.05 * i if i in thick else None for i in range(F)]
    g.meta["pulmonary_thickness"] = [None] * F
    if thick:
        la = np.zeros((F, m), dtype=bool); la[:, m // 2:] = True
        g.meta["lumen_aortic"] = la.reshape(-1)
        if with_wall:
            g.meta["wall_aortic"] = la.reshape(-1).copy()
    if lc:
        mm.centerline.with_lumen_centroids(g)
    return g


@pytest.mark.parametrize("case", [
    dict(Fa=7, Fb=5, ra=3, rb=1, anomalous=False, eem=True, wall=True, thick_a=(), thick_b=()),
    dict(Fa=6, Fb=6, ra=0, rb=0, anomalous=True, eem=False, wall=True, thick_a=(0, 1, 2), thick_b=(1, 4)),
    dict(Fa=5, Fb=8, ra=2, rb=6, anomalous=True, eem=True, wall=False, thick_a=(), thick_b=(3,)),
    dict(Fa=6, Fb=6, ra=4, rb=2, anomalous=False, eem=False, wall=False, thick_a=(), thick_b=(), roll=2),
    dict(Fa=4, Fb=4, ra=1, rb=1, anomalous=True, eem=True, wall=True, thick_a=(0, 1, 2, 3), thick_b=(), lc=False),
])
def test_postprocess_pair_flat_equals_frame_model(mm, PP, FR, case):
    from mm_checkers.postproc_flat import postprocess_pair_regular
    m = 48
    a = _regular_geom(mm, case["Fa"], m, 11, 0.0, 0.5, case["ra"], case["eem"], case["wall"], case["thick_a"],
                      roll=case.get("roll", 0), lc=case.get("lc", True))
    b = _regular_geom(mm, case["Fb"], m, 12, 3.0, 0.5 + 1e-3, case["rb"], case["eem"], case["wall"], case["thick_b"],
                      lc=case.get("lc", True))
    fast = postprocess_pair_regular(a.copy(), b.copy(), 0.03, case["anomalous"])
    assert fast is not None
    fa, fb = PP.postprocess_pair(FR.to_frames(a), FR.to_frames(b), 0.03, case["anomalous"])
    for got, exp_frames, src in ((fast[0], fa, a), (fast[1], fb, b)):
        exp = FR.from_frames(exp_frames, src.label, src.meta)
        for name in ("ids", "lumen_ids", "orig_frames", "centroids", "lumen_off", "lumen", "cath_off", "cath", "extra_off",
                     "extra", "has_ref", "ref", "lumen_centroids", "has_lumen_centroid"):
            x, y = getattr(got, name), getattr(exp, name)
            assert (x is None) == (y is None), name
            assert x is None or np.array_equal(x, y), name
        for k in got.meta["extra_counts"]:
            assert np.array_equal(got.meta["extra_counts"][k], exp.meta["extra_counts"][k]), k
        assert got.meta["aortic_thickness"] == exp.meta["aortic_thickness"]
        for k in ("lumen_aortic", "wall_aortic"):
            assert (k in got.meta) == (k in exp.meta), k
            assert k not in got.meta or np.array_equal(got.meta[k], exp.meta[k]), k
    # different sampling rates take the interpolating branch: not handled here
    c = _regular_geom(mm, 6, m, 13, 0.0, 1.0, 2, False, False, ())
    assert postprocess_pair_regular(c, b, 0.03, False) is None


# ---- converters (multimodars/_converters.py) -------------------------------------------------------
def test_to_array_and_numpy_to_geometry(mm):
    g = mm.synthetic_pullback(4, 30, pullback_id=2)
    d = mm.to_array(g)
    assert set(d) == {"lumen", "eem", "calcification", "sidebranch", "catheter", "wall", "reference"}
    assert d["lumen"].shape == (120, 4) and d["catheter"].shape == (80, 4) and d["reference"].shape == (1, 4)
    assert d["lumen"][:, 0].tolist() == sorted(d["lumen"][:, 0].tolist()) and d["eem"].shape == (0, 4)
    assert np.array_equal(d["lumen"][:, 1:], g.lumen) and np.array_equal(d["reference"][0, 1:], g.ref[0])
    h = mm.numpy_to_geometry(d["lumen"], catheter_arr=d["catheter"], reference_arr=d["reference"], label="x")
    assert np.array_equal(h.lumen, g.lumen) and np.array_equal(h.cath, g.cath) and h.ids.tolist() == [0, 1, 2, 3]
    assert h.has_ref.all() and np.array_equal(h.ref, np.tile(g.ref[0], (4, 1)))        # the reference's quirk (:592)
    assert np.allclose(h.centroids, [g.frame_lumen(i).mean(axis=0) for i in range(4)])
    a, b = mm.to_array(mm.GeometryPair(g, h, "p"))
    assert np.array_equal(a["lumen"], b["lumen"])
    cl = mm.Centerline.from_contour_points([[0, 0, 0], [1, 0, 0], [2, 0, 0]])
    assert mm.to_array(cl).tolist() == [[0, 0, 0, 0], [1, 1, 0, 0], [2, 2, 0, 0]]
    with pytest.raises(ValueError, match="cannot be empty"):
        mm.numpy_to_geometry(np.zeros((0, 4)))
    with pytest.raises(TypeError):
        mm.to_array(object())
