"""The product's reader + geometry builder (multimoda_rs_amd.io) against the independent pure-Python
restatement of the reference's builder in tests/refbuild.py, on every fixture directory, bit for bit.
The golden vectors and the config-1 parity tests take their ORACLE inputs from refbuild, so a bug in
the product builder shows up here and there instead of cancelling out."""
import os

import numpy as np
import pytest

import refbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FOLDERS = ["ivus_rest", "ivus_stress", "idealized_geometry", "examples_ivus_rest", "examples_ivus_stress"]


@pytest.mark.parametrize("builder", ["native", "python"])
@pytest.mark.parametrize("diastole", [True, False])
@pytest.mark.parametrize("folder", FOLDERS)
def test_product_builder_equals_independent_builder(mm, folder, diastole, builder, monkeypatch):
    """native = mm_build_geometry behind the C ABI (the product path); python = its in-package checker."""
    import __graft_entry__ as ge
    ge.build()
    if builder == "python":
        monkeypatch.setenv("MM_PY_BUILDER", "1")
    path = os.path.join(GOLD, folder)
    g = mm.build_geometry_from_inputdata(None, path, folder, diastole)
    b = refbuild.build_geometry(path, diastole)
    F = len(b["ids"])
    assert g.n_frames == F
    assert list(g.ids) == b["ids"] and list(g.lumen_ids) == b["ids"] and list(g.orig_frames) == b["orig_frames"]
    assert np.array_equal(g.centroids, np.array(b["centroids"]))
    for i in range(F):
        assert np.array_equal(g.frame_lumen(i), b["lumens"][i]), (folder, diastole, i)
        assert np.array_equal(g.frame_cath(i), b["catheters"][i]), (folder, diastole, i)
    has = {i for i in range(F) if g.has_ref[i]}
    assert has == set(b["ref_points"])
    for i, p in b["ref_points"].items():
        assert list(g.ref[i]) == p


@pytest.mark.parametrize("builder", ["native", "python"])
def test_fixture_with_eem_calcium_and_branch_files(mm, builder, monkeypatch):
    """data/fixtures/ivus_full of the reference (build.rs: test_full_directory_all_frames_explicit_types): EEM contours
    on the lumen frames, calcium and side-branch contours on frames that have NO lumen (they take part in the id
    mapping and are dropped).  Every frame carries lumen, EEM and catheter with one id and one original frame; the
    systolic phase of this fixture has its reference point on a frame without contours -- the builder's integrity
    check refuses it (the reference's test only builds the diastolic phase)."""
    import __graft_entry__ as ge
    ge.build()
    if builder == "python":
        monkeypatch.setenv("MM_PY_BUILDER", "1")
    path = os.path.join(GOLD, "ivus_full")
    g = mm.build_geometry_from_inputdata(None, path, "full", True)
    b = refbuild.build_geometry(path, True)
    F = g.n_frames
    assert F == len(b["ids"]) == 3 and list(g.ids) == b["ids"] == [0, 1, 2] and list(g.orig_frames) == b["orig_frames"]
    cnt = g.meta["extra_counts"]
    assert cnt["eem"].tolist() == [501] * 3 and not cnt["calcification"].any() and not cnt["sidebranch"].any()
    assert np.array_equal(g.centroids, np.array(b["centroids"]))
    for i in range(F):
        assert set(b["extras"][i]) == {"eem"}
        assert np.array_equal(g.frame_lumen(i), b["lumens"][i]) and np.array_equal(g.frame_cath(i), b["catheters"][i])
        assert np.array_equal(g.extra[g.extra_off[i]:g.extra_off[i + 1]], b["extras"][i]["eem"])
        assert g.frame_cath(i).shape[0] == 20 and (g.frame_cath(i)[:, 2] == g.centroids[i, 2]).all()
    with pytest.raises(RuntimeError, match="Expected exactly one reference point, found 0"):
        mm.build_geometry_from_inputdata(None, path, "full", False)


def test_rest_directory_area_elliptic(mm):
    """build.rs: test_rest_directory_area_elliptic on data/fixtures/ivus_rest -- the reference's expected values."""
    import __graft_entry__ as ge
    from multimoda_rs_amd import api
    ge.build()
    g = mm.build_geometry_from_inputdata(None, os.path.join(GOLD, "ivus_rest"), "full", True)
    lum = g.frame_lumen(0)
    assert int(g.orig_frames[0]) == 385
    x, y = lum[:, 0], lum[:, 1]
    area = 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))     # Contour::area is the shoelace formula
    assert area == pytest.approx(5.42, abs=0.1)
    assert api._find_farthest_points(lum)[1] == pytest.approx(5.2, abs=0.1)
    assert api._elliptic_ratio(lum) == pytest.approx(4.52, abs=0.1)
    assert g.meta["aortic_thickness"][0] == 0.96 and g.meta["pulmonary_thickness"][0] == 1.68
    assert g.has_ref[0] == 1                                    # the reference point sits on frame 0 (= original frame 385)
    assert (g.frame_cath(0)[:, 2] == g.centroids[0, 2]).all()   # test_catheter_contour_properties


def test_label_argument_names_the_geometry(mm):
    """build.rs:181 / test_build_geometry_with_input_data: Geometry.label is the `label` argument, not InputData.label."""
    import __graft_entry__ as ge
    ge.build()
    d = mm.InputData(lumen=np.array([[0, 1.0, 2.0, 3.0]]), eem=np.array([[0, 1.0, 2.0, 3.0]]),
                     ref_point=np.array([0, 1.0, 2.0, 3.0]), diastole=True, label="test")
    g = mm.build_geometry_from_inputdata(d, None, "test_label", True, (0.0, 0.0), 1.0, 10)
    assert g.n_frames == 1 and g.label == "test_label"
    assert mm.build_geometry_from_inputdata(None, os.path.join(GOLD, "ivus_rest"), "path_test", True).label == "path_test"
    with pytest.raises(RuntimeError, match="Either input_data or path must be provided"):
        mm.build_geometry_from_inputdata(None, None, "x", True)                        # test_error_on_no_input


def test_rows_the_reference_reader_skips(tmp_path):
    """read_contour_data (input.rs:172-194): rows that do not deserialize into ContourPoint are skipped."""
    p = tmp_path / "c.csv"
    p.write_text("1,1.0,2.0,3.0\n1.0,1.0,2.0,3.0\n2,1e0,2.5,3\n 3,1,2,3\n4,1_0,2,3\n5,1,2,3,true\n6,1,2,3,maybe\n7,1,2\n8,1,2,3abc\n")
    rows = refbuild.read_contour_data(str(p))
    # (row 5 would deserialize, but has another field count than the first record: csv::ReaderBuilder is not flexible)
    assert [(r["frame"], r["x"], r["aortic"]) for r in rows] == [(1, 1.0, False), (2, 1.0, False)]
    p.write_text("5,1,2,3,true\n6,1,2,3,maybe\n7,1.5,2,3,false\n8,1,2,3\n")
    rows = refbuild.read_contour_data(str(p))
    assert [(r["frame"], r["x"], r["aortic"]) for r in rows] == [(5, 1.0, True), (7, 1.5, False)]


def test_native_builder_equals_python_builder_with_extras_and_ragged_input(mm, monkeypatch):
    """EEM / calcification / sidebranch contours (also for frames without a lumen and a lumen-less reference frame),
    ragged contours, records that reorder, duplicate and skip frames, per-point aortic flags, no catheter."""
    import math
    import __graft_entry__ as ge
    ge.build()
    rng = np.random.default_rng(3)

    def ring(frame, n, r, z, jitter=0.02):
        t = np.sort(rng.uniform(0, 2 * math.pi, n))
        return np.stack([np.full(n, float(frame)), 4.5 + r * np.cos(t) + rng.normal(0, jitter, n),
                         4.4 + 0.8 * r * np.sin(t) + rng.normal(0, jitter, n), np.full(n, z)], 1)
    frames = [12, 15, 19, 23, 31, 40]
    lum = np.concatenate([ring(f, 40 + 3 * k, 2.0, 0.4 * f) for k, f in enumerate(frames)])
    rng.shuffle(lum)                                                     # rows of a frame are not contiguous in the file
    eem = np.concatenate([ring(f, 30, 2.6, 0.4 * f) for f in (12, 19, 23, 77)])    # 77: no lumen -> dropped, but mapped
    calc = np.concatenate([ring(f, 7, 1.0, 0.4 * f) for f in (15, 40)])
    side = ring(31, 9, 0.7, 0.4 * 31)
    flags = lum[:, 1] > 4.5
    recs = [mm.Record(23, "D", 1.1, None), mm.Record(12, "S", None, 2.0), mm.Record(40, "D", None, None),
            mm.Record(23, "D", 0.9, 1.9), mm.Record(99, "D", 1.0, 1.0), mm.Record(15, "X", 3.0, 3.0)]
    for n_points in (20, 0):
        for dia in (True, False):
            for ref_frame in (19, 14):                                    # 14: a frame index no contour has
                d = mm.InputData(lumen=lum, ref_point=np.array([ref_frame, 6.5, 4.4, 7.6]), diastole=dia, label="x",
                                 eem=eem, calcification=calc, sidebranch=side, record=recs, lumen_aortic=flags)
                monkeypatch.delenv("MM_PY_BUILDER", raising=False)
                # ragged frames / a lumen-less reference frame are what check_geometry_integrity rejects
                # (test_builder_ends_with_the_integrity_check below): the builder's mechanics are compared without it
                a = mm.build_geometry_from_inputdata(d, n_points=n_points, check_integrity=False)
                monkeypatch.setenv("MM_PY_BUILDER", "1")
                b = mm.build_geometry_from_inputdata(d, n_points=n_points, check_integrity=False)
                for name in ("ids", "lumen_ids", "orig_frames", "centroids", "lumen_off", "lumen", "cath_off", "cath",
                             "extra_off", "extra", "has_ref", "ref"):
                    x, y = getattr(a, name), getattr(b, name)
                    assert (x is None) == (y is None), name
                    if x is not None:
                        assert np.array_equal(x, y), (name, n_points, dia, ref_frame)
                assert a.label == b.label
                for k in ("eem", "calcification", "sidebranch", "wall"):
                    assert np.array_equal(a.meta["extra_counts"][k], b.meta["extra_counts"][k])
                assert a.meta["aortic_thickness"] == b.meta["aortic_thickness"]
                assert a.meta["pulmonary_thickness"] == b.meta["pulmonary_thickness"]
                assert np.array_equal(a.meta["lumen_aortic"], b.meta["lumen_aortic"])
                assert np.array_equal(a.meta["lumen_aortic"], a.lumen[:, 0] > 4.5)


def test_builder_ends_with_the_integrity_check(mm, monkeypatch):
    """build.rs:199 -> integrity_check.rs:8-33: what a built geometry can fail -- no frame carries the reference point
    (:107-118), frames differ in lumen / extras point count (:121-166) -- is an error with the reference's message,
    from the native builder and from the Python checker alike; a consistent input passes both."""
    import math
    import __graft_entry__ as ge
    ge.build()

    def ring(frame, n, r, z):
        t = np.linspace(0, 2 * math.pi, n, endpoint=False)
        return np.stack([np.full(n, float(frame)), 4.5 + r * np.cos(t), 4.4 + 0.8 * r * np.sin(t), np.full(n, z)], 1)

    def lumen(counts):
        return np.concatenate([ring(f, n, 2.0, 0.5 * f) for f, n in counts])
    ok_lum = lumen([(3, 40), (5, 40), (9, 40)])
    ref = np.array([5.0, 6.5, 4.4, 2.5])
    cases = [
        (dict(lumen=ok_lum, ref_point=ref), None),
        (dict(lumen=ok_lum, ref_point=np.array([4.0, 6.5, 4.4, 2.0])), "Expected exactly one reference point, found 0"),
        (dict(lumen=lumen([(3, 40), (5, 40), (9, 37)]), ref_point=ref),
         "Lumen point count mismatch in frame 1 (ID 1). Expected 37, found 40"),         # frame 0 = the highest original frame
        (dict(lumen=ok_lum, ref_point=ref, eem=np.concatenate([ring(3, 30, 2.6, 1.5), ring(9, 31, 2.6, 4.5)])),
         "Eem contour point count mismatch in frame 2 (ID 2). Expected 31, found 30"),
    ]
    for kw, msg in cases:
        got = []
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("MM_PY_BUILDER", env)
            else:
                monkeypatch.delenv("MM_PY_BUILDER", raising=False)
            d = mm.InputData(diastole=True, label="x", **kw)
            if msg is None:
                assert mm.build_geometry_from_inputdata(d).n_frames == 3
                continue
            with pytest.raises(RuntimeError) as e:
                mm.build_geometry_from_inputdata(d)
            assert msg in str(e.value)
            got.append(str(e.value).split(" (mm_status")[0])
            assert mm.build_geometry_from_inputdata(d, check_integrity=False).n_frames == 3
        assert len(set(got)) <= 1, got
