"""CPU-side tests of the centerline placement path: every host f64 stage of
csrc/mm_centerline.cpp (through the C ABI, include/mm_centerline.h) is bit-identical to the oracle
(oracle/mm_oracle_cl.c).  None of these entry points touches the GPU; the refinement grid, which
does, is tested in test_gpu_centerline.py."""
import math

import numpy as np
import pytest

from helpers import geoms_equal, to_oracle, to_oracle_cl


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="module")
def ocl(oracle):
    from oracle import oracle_cl
    oracle_cl.lib()
    return oracle_cl


@pytest.fixture(scope="module")
def case(built, mm):
    return mm.synth.synthetic_centerline_case(n_frames=14, n_points=96, n_ccta=1500, seed=11,
                                              true_rotation_deg=-52.0, true_index=9)


def _cl_equal(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ("x", "y", "z", "tx", "ty", "tz", "radius", "branch_id"))


def test_centerline_from_points_and_ref_idx(built, mm, ocl):
    rng = np.random.default_rng(0)
    pts = np.cumsum(rng.normal(0, 1, size=(40, 3)), axis=0)
    cl = mm.Centerline.from_contour_points(pts)
    assert _cl_equal(cl.points, ocl.centerline_from_points(pts))
    for q in rng.normal(0, 5, size=(20, 3)):
        assert cl.find_reference_cl_point_idx(q) == ocl.find_ref_idx(to_oracle_cl(ocl, cl), q)
    with pytest.raises(RuntimeError, match="at least two points"):
        mm.Centerline.from_contour_points([[0.0, 0.0, 0.0]])


def test_numpy_to_centerline_nan_handling(built, mm):          # multimodars/_converters.py:605-686
    a = np.array([[0.0, 0, 0], [np.nan, 0, 1], [2.0, 0, 2]])
    cl = mm.numpy_to_centerline(a)
    assert cl.points["x"].tolist() == [0.0, 1.0, 2.0]
    with pytest.raises(ValueError, match="All values are NaN"):
        mm.numpy_to_centerline(np.array([[np.nan, 0, 0], [np.nan, 0, 1]]))
    with pytest.raises(ValueError, match=r"\(N,3\)"):
        mm.numpy_to_centerline(np.zeros((3, 2)))


def test_preprocess_centerline_matches_oracle(built, mm, ocl, oracle, case):
    g = case["geometry"]
    cl = case["centerline"]
    rcl, sp = mm.preprocess_centerline(cl, g)
    ocl_cl, osp = ocl.preprocess_centerline(to_oracle_cl(ocl, cl), to_oracle(oracle, g))
    assert sp == osp and len(rcl) == len(ocl_cl) and _cl_equal(rcl.points, ocl_cl)
    # ascending z input is reversed, side branches are dropped, a one-frame mesh falls back to the mean segment
    rev = mm.Centerline(cl.points[::-1].copy())
    rev.points["branch_id"][::7] = 3
    g1 = mm.FlatGeometry.from_frames([g.frame_lumen(0)])
    r2, s2 = mm.preprocess_centerline(rev, g1)
    o2, os2 = ocl.preprocess_centerline(to_oracle_cl(ocl, rev), to_oracle(oracle, g1))
    assert s2 == os2 and _cl_equal(r2.points, o2)
    only_side = mm.Centerline(cl.points.copy())
    only_side.points["branch_id"] = 1
    with pytest.raises(RuntimeError, match="no branch-0 points"):
        mm.preprocess_centerline(only_side, g)


def test_sort_contour_points_matches_oracle(built, mm, ocl):
    rng = np.random.default_rng(3)
    L = mm._native.lib()
    for n in (1, 2, 3, 7, 64, 501):
        p = rng.normal(0, 2, size=(n, 3))
        if n >= 7:
            p[3] = p[5]                       # duplicate point: equal keys keep their order (stable sort)
            p[1, 1] = p[:, 1].max(); p[6, 1] = p[1, 1]   # two maxima of y: the later one starts the contour
        q = np.ascontiguousarray(p.copy())
        assert L.mm_sort_contour_points(mm._native._ptr(q), n) == 0
        assert np.array_equal(q, ocl.sort_contour_points(p))


def test_rotate_geometry_matches_oracle(built, mm, ocl, oracle):
    g = mm.synthetic_pullback(6, 120, pullback_id=2)
    mm.centerline.with_lumen_centroids(g)
    og = to_oracle(oracle, g)
    for ang in (0.3, -1.9, 0.0, math.pi):
        mm.centerline.rotate_geometry(g, ang)
        ocl.rotate_geometry(og, ang)
        assert geoms_equal(g, og)


def test_apply_transformations_matches_oracle(built, mm, ocl, oracle, case):
    g = case["geometry"].copy()
    h = mm.synthetic_pullback(g.n_frames, 80, pullback_id=1, seed=5)      # second geometry of a pair
    mm.centerline.with_lumen_centroids(h)
    rcl, _ = mm.preprocess_centerline(case["centerline"], g)
    og, oh, orcl = to_oracle(oracle, g), to_oracle(oracle, h), to_oracle_cl(ocl, rcl)
    ref = rcl.xyz()[5] + 0.01
    n = mm.centerline.apply_transformations([g, h], rcl, ref)
    assert n == ocl.apply_transformations([og, oh], orcl, ref) == g.n_frames
    assert geoms_equal(g, og) and geoms_equal(h, oh)
    # frames that run past the end of the centerline keep their place
    g2 = case["geometry"].copy()
    og2 = to_oracle(oracle, g2)
    tail = rcl.xyz()[len(rcl) - 4]
    n2 = mm.centerline.apply_transformations([g2], rcl, tail)
    assert n2 == ocl.apply_transformations([og2], orcl, tail) == 4
    assert geoms_equal(g2, og2)
    assert np.array_equal(g2.lumen[g2.lumen_off[4]:], case["geometry"].lumen[g2.lumen_off[4]:])
    # without contour centroids align_frame falls back to the mean and the frame centroid becomes 0 (:532)
    g3 = case["geometry"].copy(); g3.lumen_centroids = None; g3.has_lumen_centroid = None
    og3 = to_oracle(oracle, g3)
    mm.centerline.apply_transformations([g3], rcl, ref)
    ocl.apply_transformations([og3], orcl, ref)
    assert geoms_equal(g3, og3) and not g3.centroids.any()


def test_best_rotation_three_point_matches_oracle(built, mm, ocl, case):
    g = case["geometry"]
    rcl, _ = mm.preprocess_centerline(case["centerline"], g)
    orcl = to_oracle_cl(ocl, rcl)
    k = rcl.find_reference_cl_point_idx(case["main_ref_pt"])
    for step_deg, cen in ((1.0, g.lumen_centroids[0]), (0.37, None), (22.5, g.lumen_centroids[0])):
        a = mm.centerline.best_rotation_three_point(g.frame_lumen(0), cen, g.meta["ref_point_index"],
                                                    case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"],
                                                    math.radians(step_deg), rcl.points[k])
        b = ocl.best_rotation_three_point(g.frame_lumen(0), cen, g.meta["ref_point_index"], case["main_ref_pt"],
                                          case["ccw_ref_pt"], case["cw_ref_pt"], math.radians(step_deg), orcl[k])
        assert a == b
    assert abs(math.degrees(a) - (360.0 - 52.0)) <= 22.5 + 1e-9


def test_align_three_point_and_manual_match_oracle(built, mm, ocl, oracle, case):
    g = case["geometry"]
    ocl_cl = to_oracle_cl(ocl, case["centerline"])
    og = to_oracle(oracle, g)
    out, sp, rot_deg = mm.align_three_point(case["centerline"], g, case["main_ref_pt"], case["ccw_ref_pt"],
                                            case["cw_ref_pt"], angle_step_deg=1.0)
    osp, orot = ocl.align_three_point(ocl_cl, [og], g.meta["ref_point_index"], case["main_ref_pt"],
                                      case["ccw_ref_pt"], case["cw_ref_pt"], math.radians(1.0))
    assert sp == osp and rot_deg == orot * (180.0 / math.pi)
    assert geoms_equal(out, og)
    assert np.array_equal(g.lumen, case["geometry"].lumen)              # the input is not modified
    # the sweep lands within a step of the constructed twist and the placement reproduces the truth
    assert abs(rot_deg - (360.0 - 52.0)) <= 1.0 + 1e-9
    assert np.abs(out.lumen - case["truth"]["placed"].lumen).max() < 0.12

    pair = mm.GeometryPair(g, mm.synthetic_pullback(g.n_frames, 96, pullback_id=1, seed=11), "a - b")
    mm.centerline.with_lumen_centroids(pair.geom_b)
    oa, ob = to_oracle(oracle, pair.geom_a), to_oracle(oracle, pair.geom_b)
    ref = case["truth"]["placed"].centroids[0]
    outp, sp2, rot2 = mm.align_manual(case["centerline"], pair, -52.0, ref)
    osp2, orot2 = ocl.align_manual(ocl_cl, [oa, ob], -52.0, ref)
    assert sp2 == osp2 and rot2 == orot2 * (180.0 / math.pi) == pytest.approx(-52.0, abs=1e-12)
    assert geoms_equal(outp.geom_a, oa) and geoms_equal(outp.geom_b, ob)
    assert np.abs(outp.geom_a.lumen - case["truth"]["placed"].lumen).max() < 1e-9


def test_align_errors(built, mm, case):
    g = case["geometry"].copy()
    g.has_ref[:] = 0
    with pytest.raises(RuntimeError, match="No reference point found"):
        mm.align_three_point(case["centerline"], g, case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"])
    with pytest.raises(TypeError):
        mm.align_manual(case["centerline"], object(), 10.0, (0, 0, 0))


def test_centerline_resample_and_branches(built, mm):              # centerline.rs:1378-1403, py_centerline.rs:211-232
    cl = mm.Centerline.from_contour_points([[0.0, 0, 0], [10.0, 0, 0]]).resample(2.5)
    assert len(cl) == 5 and np.allclose(cl.points["x"], np.arange(5) * 2.5, atol=1e-9) and cl.points["x"][-1] == 10.0
    assert np.allclose(cl.points["tx"], 1.0) and (cl.points["ty"] == 0.0).all()
    two = mm.Centerline.from_arrays([[0, 0, 0], [10, 0, 0], [10, 0, 0], [10, 5, 0]], np.zeros((4, 3)), branch_id=[0, 0, 1, 1])
    r = two.resample(2.0)
    b = r.points["branch_id"]
    assert sorted(set(b.tolist())) == [0, 1] and (np.diff(b) >= 0).all()
    side = r.points[b == 1]
    assert abs(side["y"][0]) < 1e-9 and side["x"].tolist() == [10.0] * len(side) and side["y"][-1] == 5.0
    assert (r.points[b == 0]["x"] <= 10.0).all() and r.points[b == 0]["y"].max() == 0.0
    main = r.get_branch(1)
    assert (main.points["branch_id"] == 0).all() and len(main) == len(side)
    with pytest.raises(ValueError, match="branch_id 7 not found"):
        r.get_branch(7)
    assert two.mean_spacing() == 10.0 and mm.Centerline.from_contour_points([[0.0, 0, 0], [3.0, 4.0, 0]]).mean_spacing() == 5.0
    assert len(cl.resample(0.0)) == 5


def _walled(mm, case, thickness):
    """The case's pullback after the post-steps of align_frames_in_geometry (host only): with an aortic thickness
    it is an anomalous coronary -- lumen and wall points carry ContourPoint.aortic -- otherwise plain offset walls."""
    from multimoda_rs_amd import native_frames as NF
    g = case["geometry"].copy()
    F = g.n_frames
    g.meta["aortic_thickness"] = [thickness] * F
    g.meta["pulmonary_thickness"] = [None] * F
    w, anomalous = NF.finish_within(g, int(np.nonzero(g.has_ref)[0][0]), True)
    assert anomalous == (thickness is not None) and int(w.meta["extra_counts"]["wall"].sum()) > 0
    w.meta["ref_point_index"] = g.meta["ref_point_index"]
    return w


@pytest.mark.parametrize("thickness", [0.9, None])
def test_align_walls_native_equals_oracle_and_python(built, mm, ocl, oracle, case, thickness, monkeypatch):
    """align_walls (align.rs:381-595; the reference holds no test of it -> parity is against the restatements only):
    mm_align_walls inside mm_align_manual == oracle == the Python checker, bit for bit, for walls with an aortic side
    (unambiguous direction) and without (major axis, smaller rotation); the per-point aortic flags follow their
    points through rotate_geometry's sort (contour.rs:385-390 moves whole ContourPoints)."""
    g = _walled(mm, case, thickness)
    ocl_cl = to_oracle_cl(ocl, case["centerline"])
    ref = case["truth"]["placed"].centroids[0]
    out, sp, rot = mm.align_manual(case["centerline"], g, -52.0, ref, align_wall_anomalous=True)
    og = to_oracle(oracle, g)
    assert og.wall_kind1 > 0 and (og.wall_aortic is not None) == (thickness is not None)
    osp, orot = ocl.align_manual(ocl_cl, [og], -52.0, ref, align_wall_anomalous=True)
    assert sp == osp and geoms_equal(out, og)
    plain, _, _ = mm.align_manual(case["centerline"], g, -52.0, ref)
    assert np.array_equal(plain.lumen, out.lumen) and not np.array_equal(plain.extra, out.extra)   # walls moved, only walls
    if thickness is not None:
        assert np.array_equal(out.meta["wall_aortic"].astype(np.uint8), og.wall_aortic)
        assert np.array_equal(out.meta["lumen_aortic"].astype(np.uint8), og.lumen_aortic)
        assert not np.array_equal(out.meta["wall_aortic"], g.meta["wall_aortic"])                   # the sort moved them
        # the aortic side of every wall points along one parallel-transported direction: consecutive frames agree
        F = out.n_frames
        per = out.meta["extra_counts"]["wall"]
        off = np.concatenate([[0], np.cumsum(per)])
        walls = out.extra.reshape(-1, 3)[-int(per.sum()):] if out.meta["extra_counts"]["eem"].sum() == 0 else None
        if walls is not None:
            dirs = []
            for i in range(F):
                w, fl = walls[off[i]:off[i + 1]], out.meta["wall_aortic"][off[i]:off[i + 1]]
                d = w[fl].mean(axis=0) - out.centroids[i]
                dirs.append(d / np.linalg.norm(d))
            assert min(float(np.dot(dirs[i], dirs[i + 1])) for i in range(F - 1)) > 0.99
    monkeypatch.setenv("MM_PY_POSTPROC", "1")
    py, _, _ = mm.align_manual(case["centerline"], g, -52.0, ref, align_wall_anomalous=True)
    assert np.array_equal(py.extra, out.extra) and np.array_equal(py.lumen, out.lumen)
    monkeypatch.delenv("MM_PY_POSTPROC")
    # the standalone entry point, and anomalous = false / a single frame: nothing moves (:589-592)
    again = mm.centerline.align_walls(plain, True)
    assert np.array_equal(again.extra, out.extra)
    assert np.array_equal(mm.centerline.align_walls(plain, False).extra, plain.extra)


def test_read_centerline_vtp_rca(built, mm):                       # io/input.rs: test_read_centerline_vtp_rca
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "examples_centerlines", "rca_cl.vtp")
    cl = mm.read_centerline_vtp(path)
    starts = cl.branch_start_indices
    assert len(starts) == 4 and len(cl) == 2652
    assert starts[1] - starts[0] == 763                            # branch 0 = the longest VTK line by arc length
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(cl)
        assert (cl.points["branch_id"][s:e] == i).all()
    assert (cl.points["radius"] > 0.0).any()
    b0 = cl.points[starts[0]:starts[1]]
    assert (np.sqrt(b0["tx"] ** 2 + b0["ty"] ** 2 + b0["tz"] ** 2) > 0.5).any()
    # it feeds the placement like any other centerline: branch 0, descending z, resampled
    g = mm.synthetic_pullback(12, 64, pullback_id=0)
    mm.centerline.with_lumen_centroids(g)
    rcl, spacing = mm.preprocess_centerline(cl, g)
    assert len(rcl) > 20 and spacing > 0 and (rcl.points["branch_id"] == 0).all()
    assert rcl.points["z"][0] >= rcl.points["z"][-1]


def test_read_centerline_vtp_picks_longest_by_arc_length_not_point_count(built, mm, tmp_path):     # io/input.rs
    a = [(i * 10.0, 0.0, 0.0) for i in range(5)]                   # 40 mm, 5 points
    b = [(0.0, i * 0.1, 0.0) for i in range(20)]                   # 1.9 mm, 20 points
    pts = a + b
    n = len(pts)
    xml = f"""<?xml version="1.0"?>
<VTKFile type="PolyData" version="0.1" byte_order="LittleEndian" header_type="UInt32">
  <PolyData>
    <Piece NumberOfPoints="{n}" NumberOfVerts="0" NumberOfLines="2" NumberOfStrips="0" NumberOfPolys="0">
      <PointData>
        <DataArray type="Float64" Name="MaximumInscribedSphereRadius" format="ascii">
          {" ".join(["1.0"] * n)}
        </DataArray>
      </PointData>
      <Points>
        <DataArray type="Float64" Name="Points" NumberOfComponents="3" format="ascii">
          {" ".join(f"{x} {y} {z}" for x, y, z in pts)}
        </DataArray>
      </Points>
      <Lines>
        <DataArray type="Int64" Name="connectivity" format="ascii">
          {" ".join(str(i) for i in range(n))}
        </DataArray>
        <DataArray type="Int64" Name="offsets" format="ascii">
          {len(a)} {n}
        </DataArray>
      </Lines>
    </Piece>
  </PolyData>
</VTKFile>
"""
    p = tmp_path / "arc.vtp"
    p.write_text(xml)
    cl = mm.read_centerline_vtp(str(p))
    starts = cl.branch_start_indices
    assert len(starts) == 2 and starts[1] - starts[0] == len(a)    # the longer (but sparser) line A is branch 0
    assert cl.points["tx"][0] == 1.0 and cl.points["tx"][len(a) - 1] == 1.0        # last point repeats its predecessor's
    # refusals (input.rs:269-294)
    (tmp_path / "bin.vtp").write_bytes(b"<VTKFile>\x00\x01\x02")
    with pytest.raises(RuntimeError, match="appears to be a binary VTP file"):
        mm.read_centerline_vtp(str(tmp_path / "bin.vtp"))
    (tmp_path / "app.vtp").write_text(xml.replace('Name="Points" NumberOfComponents="3" format="ascii"',
                                                  'Name="Points" NumberOfComponents="3" format="appended"'))
    with pytest.raises(RuntimeError, match="binary-encoded DataArrays detected"):
        mm.read_centerline_vtp(str(tmp_path / "app.vtp"))
    (tmp_path / "odd.vtp").write_text(xml.replace(f"{len(a)} {n}\n", f"{len(a)} {n - 1}\n"))
    with pytest.raises(RuntimeError, match="last offset"):
        mm.read_centerline_vtp(str(tmp_path / "odd.vtp"))
