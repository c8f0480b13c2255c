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


@pytest.mark.parametrize("diastole", [True, False])
@pytest.mark.parametrize("folder", FOLDERS)
def test_product_builder_equals_independent_builder(mm, folder, diastole):
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


def test_rows_the_reference_reader_skips(tmp_path):
    """read_contour_data (input.rs:172-194): rows that do not deserialize into ContourPoint are skipped."""
    p = tmp_path / "c.csv"
    p.write_text("1,1.0,2.0,3.0\n1.0,1.0,2.0,3.0\n2,1e0,2.5,3\n 3,1,2,3\n4,1_0,2,3\n5,1,2,3,true\n6,1,2,3,maybe\n7,1,2\n8,1,2,3abc\n")
    rows = refbuild.read_contour_data(str(p))
    assert [(r["frame"], r["x"], r["aortic"]) for r in rows] == [(1, 1.0, False), (2, 1.0, False), (5, 1.0, True)]
