"""Builder + entry-point tests on the data fixtures of the reference's own tests
(tests/golden/{idealized_geometry,ivus_rest,ivus_stress}) and the committed oracle vectors."""
import json
import math
import os

import numpy as np
import pytest

from helpers import to_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
VEC = json.load(open(os.path.join(GOLD, "oracle_vectors.json")))
CASES = [k for k in VEC if k != "ivus_rest_between_chain_only"]


def unhex(logs):
    return [(l[0], l[1]) + tuple(float.fromhex(v) for v in l[2:]) for l in logs]


# ---------------------------------------------------------------------------------------
# CPU: builder (io/build.rs) and the oracle against the committed vectors
# ---------------------------------------------------------------------------------------
def test_builder_idealized_geometry(mm):
    g = mm.build_geometry_from_inputdata(None, os.path.join(GOLD, "idealized_geometry"), "stress", True)
    assert g.n_frames == 11 and g.lumen.shape == (2200, 3) and g.cath.shape == (220, 3)
    assert list(g.ids) == list(range(11))
    assert list(g.orig_frames) == list(range(10, -1, -1))          # proximal end (highest original frame) first
    assert g.has_ref[0] == 1 and g.has_ref.sum() == 1              # reference point sits on original frame 10
    assert list(g.centroids[:, 2]) == sorted(g.centroids[:, 2])    # smallest z at frame 0 (geometry.rs:343-351)
    for i in range(11):
        lum = g.frame_lumen(i)
        assert lum[0, 1] == lum[:, 1].max()                        # contour starts at its highest-y point
        c = lum[:, :2].mean(axis=0)
        ang = np.unwrap(np.arctan2(lum[:, 1] - c[1], lum[:, 0] - c[0]))
        assert np.all(np.diff(ang) > 0)                            # and runs counter-clockwise
        assert np.all(lum[:, 2] == g.centroids[i, 2])
    assert g.label == "stress"


def test_builder_ivus_fixtures_and_array_path(mm):
    rest = os.path.join(GOLD, "ivus_rest")
    gd = mm.build_geometry_from_inputdata(None, rest, "rest", True)
    gs = mm.build_geometry_from_inputdata(None, rest, "rest", False)
    assert gd.n_frames == gs.n_frames == 3 and gd.lumen.shape[0] % gd.n_frames == 0
    # same data through the array entry (numpy_to_inputdata contract: (N,4) [frame, x, y, z])
    d = mm.process_directory(rest, True, "rest")
    d2 = mm.numpy_to_inputdata(d.lumen, d.ref_point, True, label="rest",
                               record=[[r.frame, r.phase, r.measurement_1, r.measurement_2] for r in d.record])
    ga = mm.build_geometry_from_inputdata(d2)
    for a, b in ((gd.lumen, ga.lumen), (gd.cath, ga.cath), (gd.centroids, ga.centroids), (gd.orig_frames, ga.orig_frames)):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError, match="lumen_arr is empty"):
        mm.numpy_to_inputdata(np.zeros((0, 4)), [0, 0, 0, 0], True)
    with pytest.raises(RuntimeError, match="required contours file missing"):
        mm.build_geometry_from_inputdata(None, os.path.join(GOLD, "nope"), "x", True)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_committed_vectors(mm, oracle, name):
    v = VEC[name]
    import refbuild                                   # oracle inputs come from the independent builder, not the product's
    og = refbuild.oracle_geometry(oracle, os.path.join(GOLD, v["folder"]), v["diastole"], v["folder"])
    logs = oracle.align_within_chain(og, v["step_deg"], v["range_deg"], v["bruteforce"], v["sample_size"], n_threads=8)
    assert logs == unhex(v["logs"])
    assert [float(np.sum(og.lumen[:, 0])).hex(), float(np.sum(og.lumen[:, 1])).hex()] == v["lumen_sum_hex"]


def test_idealized_geometry_reference_expectation_oracle(mm, oracle):
    """align_within.rs:855-887: |rot| = 15 +- 1 deg, tx = -0.01 i +- 1e-3, ty = +0.01 i +- 1e-3."""
    logs = unhex(VEC["idealized_dia_hier_0p01deg"]["logs"])
    assert len(logs) == 10
    for i, (_, _, rot, tx, ty, _, _) in enumerate(logs):
        assert abs(rot) == pytest.approx(15.0, abs=1.0)
        assert tx == pytest.approx(-0.01 * (i + 1), abs=1e-3) and ty == pytest.approx(0.01 * (i + 1), abs=1e-3)


# ---------------------------------------------------------------------------------------
# GPU: product against the vectors and the reference's expectations
# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name", CASES)
def test_product_reproduces_committed_vectors(engine, mm, name, mode):
    v = VEC[name]
    g = mm.build_geometry_from_inputdata(None, os.path.join(GOLD, v["folder"]), v["folder"], v["diastole"])
    logs, _ = mm.align_within(engine, [g], v["step_deg"], v["range_deg"], v["bruteforce"], v["sample_size"], mode=mode)
    assert logs[0] == unhex(v["logs"])
    assert [float(np.sum(g.lumen[:, 0])).hex(), float(np.sum(g.lumen[:, 1])).hex()] == v["lumen_sum_hex"]
    assert [[float(x).hex() for x in g.frame_lumen(i)[0]] for i in range(g.n_frames)] == v["first_points_hex"]


@pytest.mark.gpu
def test_between_vector(engine, mm):
    v = VEC["ivus_rest_between_chain_only"]
    ga = mm.build_geometry_from_inputdata(None, os.path.join(GOLD, "ivus_rest"), "rest", True)
    gb = mm.build_geometry_from_inputdata(None, os.path.join(GOLD, "ivus_rest"), "rest", False)
    mm.align_within(engine, [ga, gb], 1.0, 90.0, False, 500)
    best, _ = mm.align_between(engine, [(ga, gb)], 90.0, 0.5, 500)
    assert float(best[0]).hex() == v["best_rotation_hex"]
    assert [[float(x).hex() for x in gb.frame_lumen(i)[0]] for i in range(gb.n_frames)] == v["b_first_points_hex"]


@pytest.mark.gpu
def test_from_file_single_idealized_reference_expectation(engine, mm):
    """align_within.rs:855-887 through the public entry point (smooth = True, hierarchical)."""
    geom, logs = mm.from_file_single(os.path.join(GOLD, "idealized_geometry"), labels=["stress"], diastole=True,
                                     step_rotation_deg=0.01, range_rotation_deg=20.0, sample_size=200, smooth=True,
                                     engine=engine)
    assert geom.n_frames == 11 and geom.meta["anomalous"] is True
    assert logs == unhex(VEC["idealized_dia_hier_0p01deg"]["logs"])
    for i, (_, _, rot, tx, ty, _, _) in enumerate(logs):
        assert abs(rot) == pytest.approx(15.0, abs=1.0)
        assert tx == pytest.approx(-0.01 * (i + 1), abs=1e-3) and ty == pytest.approx(0.01 * (i + 1), abs=1e-3)


@pytest.mark.gpu
def test_between_optimized_geometries_reference_expectation(engine, mm):
    """align_between.rs:305-373: B = A rotated by 15 deg about its proximal-end centroid; after
    align_between(30, 0.01, 500) max point error < 0.01, mean < 0.001."""
    geom, _ = mm.from_file_single(os.path.join(GOLD, "idealized_geometry"), labels=["stress"], diastole=True,
                                  step_rotation_deg=0.01, range_rotation_deg=45.0, sample_size=200, smooth=True,
                                  engine=engine)
    a, b = geom.copy(), geom.copy()
    prox = int(b.lumen_ids[0] if b.orig_frames[0] > b.orig_frames[-1] else b.lumen_ids[-1])
    cx, cy = b.centroids[prox, 0], b.centroids[prox, 1]
    c, s = math.cos(math.radians(15.0)), math.sin(math.radians(15.0))
    for arr in (b.lumen, b.cath, b.centroids, b.ref):
        x, y = arr[:, 0] - cx, arr[:, 1] - cy
        arr[:, 0], arr[:, 1] = cx + x * c - y * s, cy + x * s + y * c
    mm.align_between(engine, [(a, b)], 30.0, 0.01, 500)
    err = np.abs(a.lumen[:, :2] - b.lumen[:, :2])
    assert err.max() < 0.01 and err.mean() < 0.001
    np.testing.assert_allclose(a.centroids[:, 2], b.centroids[:, 2], atol=1e-4)


@pytest.mark.gpu
def test_full_modes_labels_and_consistency(engine, mm):
    """binding/functions.rs:1603-1662: file and array entries give the same pairs and labels
    (step 90, range 90 like the reference's fast_full)."""
    rest, stress = os.path.join(GOLD, "ivus_rest"), os.path.join(GOLD, "ivus_stress")
    labels = ["rest_dia", "rest_sys", "stress_dia", "stress_sys"]
    kw = dict(step_rotation_deg=90.0, range_rotation_deg=90.0, smooth=False, engine=engine)
    ab, cd, ac, bd, logs = mm.from_file_full(rest, stress, labels=labels, **kw)
    assert (ab.label, cd.label, ac.label, bd.label) == ("rest_dia - rest_sys", "stress_dia - stress_sys",
                                                        "rest_dia - stress_dia", "rest_sys - stress_sys")
    assert len(logs) == 4 and all(len(l) == 2 for l in logs)
    ins = [mm.process_directory(p, d, l) for (p, d), l in zip([(rest, True), (rest, False), (stress, True), (stress, False)], labels)]
    ab2, cd2, ac2, bd2, logs2 = mm.from_array_full(*ins, **kw)
    assert (ab2.label, cd2.label, ac2.label, bd2.label) == (ab.label, cd.label, ac.label, bd.label)
    assert logs2 == logs
    for p, q in ((ab, ab2), (cd, cd2), (ac, ac2), (bd, bd2)):
        assert np.array_equal(p.geom_a.lumen, q.geom_a.lumen) and np.array_equal(p.geom_b.lumen, q.geom_b.lumen)
    # the CD pair keeps its batch-1 state; AC/BD hold C and D moved again (entry.rs:206-277)
    assert not np.array_equal(cd.geom_a.lumen, ac.geom_b.lumen)
    # double-pair and single-pair modes
    dab, dcd, dlogs = mm.from_file_doublepair(rest, stress, labels=labels, **kw)
    assert np.array_equal(dab.geom_b.lumen, ab.geom_b.lumen) and dlogs == logs
    pair, (la, lb) = mm.from_file_singlepair(rest, labels=labels[:2], **kw)
    assert pair.label == "rest_dia - rest_sys" and (la, lb) == (logs[0], logs[1])
    assert np.array_equal(pair.geom_b.lumen, ab.geom_b.lumen)
    g1, l1 = mm.from_array_single(ins[2], step_rotation_deg=90.0, range_rotation_deg=90.0, smooth=False, engine=engine)
    assert l1 == logs[2] and g1.label == "stress_dia"


@pytest.mark.gpu
def test_entry_point_errors(engine, mm):
    rest = os.path.join(GOLD, "ivus_rest")
    with pytest.raises(RuntimeError, match="sample_size must be > 0"):
        mm.from_file_single(rest, sample_size=0, engine=engine)
    with pytest.raises(RuntimeError, match="required contours file missing"):
        mm.from_file_singlepair(os.path.join(GOLD, "nope"), engine=engine)


def _array_input(mm, n_frames=12, n_points=120, drop=(), with_eem=False, thickness=None, seed=1):
    """(N,4) [frame, x, y, z] arrays of a synthetic pullback, optionally with missing frames, an EEM
    contour (the lumen scaled about its centre) and per-frame wall thickness records."""
    g = mm.synthetic_pullback(n_frames, n_points, pullback_id=0, seed=seed, torsion_sigma_deg=1.0)
    keep = [i for i in range(n_frames) if i not in drop]
    rows, eem = [], []
    for i in keep:
        L = g.frame_lumen(i)
        rows.append(np.column_stack([np.full(n_points, i + 1), L]))
        if with_eem:
            c = L.mean(axis=0)
            eem.append(np.column_stack([np.full(n_points, i + 1), c + (L - c) * [1.4, 1.4, 1.0]]))
    rec = None
    if thickness is not None:
        rec = [[i + 1, "D", thickness, None] for i in keep]
    # reference point on the LAST input frame: the builder puts the proximal end (highest frame number) at
    # index 0, so the reference frame index stays valid when fill_holes inserts frames behind it (the
    # reference computes ref_idx before the chain and does not update it, align_within.rs:42-44, 139)
    last = g.frame_lumen(n_frames - 1)
    ref = [n_frames] + list(last[np.argmax(last[:, 0])])
    return mm.numpy_to_inputdata(np.concatenate(rows), ref, True, record=rec,
                                 eem_arr=np.concatenate(eem) if with_eem else None, label="synthetic")


@pytest.mark.gpu
def test_hole_filling_eem_and_walls_through_the_entry_point(engine, mm):
    """align_within.rs:136-160 through from_array_single: a pullback with a missing frame gets the averaged
    frame back (fill_holes), EEM and wall contours are smoothed along with the lumen, and the wall comes
    from the EEM (non-anomalous) offset by 1 mm (wall.rs:7-47)."""
    full, logs_full = mm.from_array_single(_array_input(mm, with_eem=True), step_rotation_deg=1.0,
                                           range_rotation_deg=20.0, engine=engine)
    holed, logs = mm.from_array_single(_array_input(mm, drop=(5,), with_eem=True), step_rotation_deg=1.0,
                                       range_rotation_deg=20.0, engine=engine)
    assert full.n_frames == 12 and holed.n_frames == 12 and len(logs) == 10          # one chain step fewer, one frame back
    assert holed.ids.tolist() == list(range(12)) and holed.lumen_ids.tolist() == list(range(12))
    dz = np.diff(holed.centroids[:, 2])
    assert np.allclose(dz, dz[0], atol=1e-9)                                          # the z gap is closed
    assert holed.meta["anomalous"] is False
    cnt = holed.meta["extra_counts"]
    assert cnt["eem"].tolist() == [120] * 12 and cnt["wall"].tolist() == [120] * 12
    from multimoda_rs_amd import frames as FR
    fr = FR.to_frames(holed)
    for f in fr:
        d_wall = np.linalg.norm(f.extras["wall"].points[:, :2] - f.extras["wall"].points[:, :2].mean(axis=0), axis=1)
        d_eem = np.linalg.norm(f.extras["eem"].points[:, :2] - f.extras["eem"].points[:, :2].mean(axis=0), axis=1)
        assert 0.5 < (d_wall - d_eem).mean() < 1.1                                    # EEM + ~1 mm (smoothed)
        assert (f.extras["wall"].points[:, 2] == f.centroid[2]).all()
    # an anomalous (thickness-bearing) pullback builds the aortic wall from the lumen and flags the aortic half
    an, _ = mm.from_array_single(_array_input(mm, thickness=0.9), step_rotation_deg=1.0, range_rotation_deg=20.0,
                                 engine=engine)
    assert an.meta["anomalous"] is True and an.meta["lumen_aortic"].sum() == 12 * 60
    assert an.meta["extra_counts"]["wall"].tolist() == [120] * 12 and an.meta["aortic_thickness"] == [0.9] * 12


@pytest.mark.gpu
@pytest.mark.parametrize("bruteforce", [False, True])
def test_config1_example_data_singlepair_matches_oracle(engine, mm, oracle, bruteforce):
    """BASELINE config 1: single-pair mode on the reference's examples/data/ivus_rest (20 + 17 frames x 501
    points, committed as tests/golden/examples_ivus_rest): chain logs of both pullbacks and the between
    rotation are identical to the oracle's restatement of align_frames_in_geometry / align_between."""
    from helpers import to_oracle
    path = os.path.join(GOLD, "examples_ivus_rest")
    pair, (logs_d, logs_s) = mm.from_file_singlepair(path, step_rotation_deg=0.5, range_rotation_deg=90.0,
                                                     bruteforce=bruteforce, smooth=False, postprocessing=False,
                                                     engine=engine)
    assert pair.geom_a.n_frames == 20 and pair.geom_b.n_frames == 17 and pair.label == "examples_ivus_rest - examples_ivus_rest"
    for dia, logs in ((True, logs_d), (False, logs_s)):
        import refbuild                               # oracle inputs from the independent builder (tests/refbuild.py)
        og = refbuild.oracle_geometry(oracle, path, dia, "x")
        assert list(logs) == oracle.align_within_chain(og, 0.5, 90.0, bruteforce, 500, n_threads=8)
    assert len(logs_d) == 19 and len(logs_s) == 16


# ---------------------------------------------------------------------------------------
# Frame.lumen.centroid: recomputed by Frame::translate, left alone by Frame::rotate (frame.rs:17-63)
# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("folder", ["idealized_geometry", "examples_ivus_rest"])
def test_lumen_contour_centroid_is_carried_like_the_reference(engine, mm, oracle, folder, mode):
    """After the chain a frame's lumen.centroid is the mean of its lumen BEFORE the step's rotation (the last
    Frame::translate computed it); align_between ends with a translation, so the moved geometry carries fresh
    means.  Product (mm_geometry.lumen_centroid through mm_align_within / mm_align_between) against the oracle's
    restatement fed by the independent builder, bit for bit."""
    import refbuild
    path = os.path.join(GOLD, folder)
    ga = mm.build_geometry_from_inputdata(None, path, "x", True)
    gb = mm.build_geometry_from_inputdata(None, path, "x", False)
    oa = refbuild.oracle_geometry(oracle, path, True, "x")
    ob = refbuild.oracle_geometry(oracle, path, False, "x")
    assert np.array_equal(ga.lumen_centroids, oa.lumen_centroids) and np.array_equal(ga.lumen_centroids, ga.centroids)
    mm.align_within(engine, [ga, gb], 0.5, 90.0, False, 500, mode=mode)
    for o in (oa, ob):
        oracle.align_within_chain(o, 0.5, 90.0, False, 500, n_threads=8)
    for g, o in ((ga, oa), (gb, ob)):
        assert np.array_equal(g.lumen_centroids, o.lumen_centroids)
        assert np.array_equal(g.lumen_centroids[0], g.centroids[0])          # frame 0 is never translated
        fresh = np.array([mm.contour_centroid(g.frame_lumen(i)) for i in range(g.n_frames)])
        assert not np.array_equal(g.lumen_centroids[1:, :2], fresh[1:, :2])   # stale in x, y ...
        assert np.allclose(g.lumen_centroids[:, 2], fresh[:, 2], atol=1e-9)   # ... z agrees up to the mean's rounding
    mm.align_between(engine, [(ga, gb)], 90.0, 0.5, 500)
    oracle.align_between(oa, ob, 90.0, 0.5, 500, n_threads=8)
    assert np.array_equal(gb.lumen_centroids, ob.lumen_centroids)
    assert np.array_equal(gb.lumen_centroids, np.array([mm.contour_centroid(gb.frame_lumen(i)) for i in range(gb.n_frames)]))
    assert np.array_equal(ga.lumen_centroids, oa.lumen_centroids)             # the reference side is not touched


@pytest.mark.gpu
def test_from_file_single_without_smoothing_returns_the_tracked_centroid(engine, mm, oracle):
    """smooth = False: the returned Frame.lumen.centroid is what the chain left, untouched by the post-step rotation
    (geometry.rs:241-250 -> Frame::rotate); smooth = True: the mean of the smoothed points (geometry.rs:204)."""
    import refbuild
    path = os.path.join(GOLD, "idealized_geometry")
    g, _logs = mm.from_file_single(path, diastole=True, step_rotation_deg=0.5, range_rotation_deg=90.0, smooth=False,
                                   write_obj=False, engine=engine)
    o = refbuild.oracle_geometry(oracle, path, True, "x")
    oracle.align_within_chain(o, 0.5, 90.0, False, 500, n_threads=8)
    assert g.lumen_centroids is not None and np.array_equal(g.lumen_centroids, o.lumen_centroids)
    s, _ = mm.from_file_single(path, diastole=True, step_rotation_deg=0.5, range_rotation_deg=90.0, smooth=True,
                               write_obj=False, engine=engine)
    assert np.array_equal(s.lumen_centroids, np.array([mm.contour_centroid(s.frame_lumen(i)) for i in range(s.n_frames)]))
