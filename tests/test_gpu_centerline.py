"""GPU parity tests of the Hausdorff refinement grid (third call site of the search,
align_algorithms.rs:339-451) and of align_combined (align.rs:169-285), through the C ABI
(include/mm_centerline.h) against the oracle: bit-exact costs, winner and transformed geometry."""
import math

import numpy as np
import pytest

from helpers import geoms_equal, to_oracle, to_oracle_cl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ocl(oracle):
    from oracle import oracle_cl
    oracle_cl.lib()
    return oracle_cl


def _aligned(mm, case):
    """`aligned` of align_combined_rs (:219-223): rotated by the three-point result and placed."""
    g = case["geometry"]
    out, _, rot_deg = mm.align_three_point(case["centerline"], g, case["main_ref_pt"], case["ccw_ref_pt"],
                                           case["cw_ref_pt"], angle_step_deg=1.0)
    rcl, _ = mm.preprocess_centerline(case["centerline"], g)
    return out, rcl, rcl.find_reference_cl_point_idx(case["main_ref_pt"])


@pytest.mark.parametrize("n_frames,n_points,n_ccta,idx_range", [
    (10, 64, 600, 2),        # small sets: search kernel with a one-angle table
    (16, 200, 3000, 1),      # downsampled frames (n_down < points per frame)
    (24, 300, 9000, 2),      # both sets above the LDS budget: streaming kernel
    (12, 80, 500, 0),        # index_search_range == 0: initial index only (:363-367)
])
def test_refine_grid_matches_oracle(engine, oracle, ocl, mm, n_frames, n_points, n_ccta, idx_range):
    case = mm.synth.synthetic_centerline_case(n_frames=n_frames, n_points=n_points, n_ccta=n_ccta, seed=n_frames,
                                              true_rotation_deg=23.0, true_index=8, clutter_frac=0.05)
    aligned, rcl, idx0 = _aligned(mm, case)
    og, orcl = to_oracle(oracle, aligned), to_oracle_cl(ocl, rcl)
    rng_, step = math.radians(6.0), math.radians(1.0)
    ba, bi, mh, costs = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.0, case["points"],
                                                                 rng_, step, idx_range)
    oba, obi, omh, ocosts = ocl.refine_alignment_hausdorff([og], orcl, idx0, 0.0, case["points"], rng_, step,
                                                           idx_range)
    assert len(costs) == len(ocosts) > 0
    assert np.array_equal(costs, ocosts)                      # bit-exact f64 Hausdorff of every candidate
    assert (ba, bi, mh) == (oba, obi, omh)
    assert np.array_equal(aligned.lumen, og.lumen)            # the target is not modified by the search
    # winner only (what align_combined asks for): lower bounds rule candidates out on the streaming
    # kernel's sets, everything is evaluated on the others -- the same first minimum either way
    wa, wi, wh, none = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.0, case["points"],
                                                                rng_, step, idx_range, return_costs=False)
    assert (wa, wi, wh) == (oba, obi, omh) and len(none) == 0


def test_refine_winner_only_prunes_and_agrees_on_large_sets(engine, mm):
    """Both sets far above the LDS budget (the bench's shape, scaled down): the winner-only path must
    return exactly what the evaluate-everything path returns (that one is checked against the oracle above
    and, at full size, by tools/bench_refine.py against the CPU restatement)."""
    case = mm.synth.synthetic_centerline_case(n_frames=40, n_points=300, n_ccta=16000, seed=5,
                                              true_rotation_deg=-31.0, true_index=10, clutter_frac=0.05)
    aligned, rcl, idx0 = _aligned(mm, case)
    for rng_deg, step_deg, idx_range in ((12.0, 1.0, 2), (4.0, 0.5, 3)):
        full = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.0, case["points"],
                                                        math.radians(rng_deg), math.radians(step_deg), idx_range)
        win = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.0, case["points"],
                                                       math.radians(rng_deg), math.radians(step_deg), idx_range,
                                                       return_costs=False)
        assert len(full[3]) >= 8 and win[:3] == full[:3]
        k = int(np.argmin(full[3]))
        assert full[2] == full[3][k]


def test_refine_grid_edge_cases(engine, oracle, ocl, mm):
    case = mm.synth.synthetic_centerline_case(n_frames=10, n_points=64, n_ccta=400, seed=3, true_index=2)
    aligned, rcl, idx0 = _aligned(mm, case)
    og, orcl = to_oracle(oracle, aligned), to_oracle_cl(ocl, rcl)
    step = math.radians(2.0)
    # negative indices and segments that overrun the centerline are skipped (:371-378)
    tail = len(rcl) - aligned.n_frames - 2
    shifted = case["points"] + (rcl.xyz()[tail] - rcl.xyz()[2])      # the cloud moved along to the tail
    for start, rng_idx, pts in ((1, 3, case["points"]), (tail, 3, shifted)):
        r = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, start, 0.1, pts,
                                                     math.radians(4.0), step, rng_idx)
        o = ocl.refine_alignment_hausdorff([og], orcl, start, 0.1, pts, math.radians(4.0), step, rng_idx)
        assert np.array_equal(r[3], o[3]) and r[:3] == o[:3] and 0 < len(r[3]) < (2 * rng_idx + 1) * 5
    # no CCTA point inside any bounding box: nothing is evaluated, the initial values come back (:406-409)
    far = case["points"] + 1.0e4
    r = mm.centerline.refine_alignment_hausdorff(engine, [aligned], rcl, idx0, 0.25, far, math.radians(4.0), step, 1)
    o = ocl.refine_alignment_hausdorff([og], orcl, idx0, 0.25, far, math.radians(4.0), step, 1)
    assert r[0] == o[0] == 0.25 and r[1] == o[1] == idx0 and len(r[3]) == len(o[3]) == 0
    assert r[2] == o[2] == np.finfo(np.float64).max


@pytest.mark.parametrize("pair", [False, True])
def test_align_combined_matches_oracle_and_truth(engine, oracle, ocl, mm, pair):
    case = mm.synth.synthetic_centerline_case(n_frames=20, n_points=160, n_ccta=3000, seed=21,
                                              true_rotation_deg=37.0, true_index=12)
    g = case["geometry"]
    geo = g
    if pair:
        b = mm.synthetic_pullback(g.n_frames, 128, pullback_id=1, seed=21)
        mm.centerline.with_lumen_centroids(b)
        geo = mm.GeometryPair(g, b, "dia - sys")
    ogs = [to_oracle(oracle, x) for x in ([geo.geom_a, geo.geom_b] if pair else [g])]
    out, sp, rot_deg = mm.align_combined(case["centerline"], geo, case["main_ref_pt"], case["ccw_ref_pt"],
                                         case["cw_ref_pt"], case["points"], angle_step_deg=1.0, angle_range_deg=5.0,
                                         index_range=2, engine=engine)
    osp, orot, oidx = ocl.align_combined(to_oracle_cl(ocl, case["centerline"]), ogs, g.meta["ref_point_index"],
                                         case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"], case["points"],
                                         math.radians(1.0), math.radians(5.0), 2)
    first = out.geom_a if pair else out
    assert sp == osp and rot_deg == orot * (180.0 / math.pi) and first.meta["refined_cl_ref_idx"] == oidx
    assert geoms_equal(first, ogs[0])
    if pair:
        assert geoms_equal(out.geom_b, ogs[1])
    # and it recovers the constructed pose (no clutter in this cloud): the three-point sweep is off by a
    # step because of the landmark noise, the refinement corrects it
    assert rot_deg == pytest.approx(37.0, abs=1e-9) and oidx == 12
    assert np.abs(first.lumen - case["truth"]["placed"].lumen).max() < 1e-9
    assert first.meta["refine_evals"] == 5 * 11


def test_from_file_pair_then_align_manual(engine, oracle, ocl, mm):
    """The composition the reference's examples/fullworkflow.py runs: from_file_singlepair -> align_* on a
    centerline.  The returned geometries carry Frame.lumen.centroid as the reference leaves it (fresh mean
    after smoothing / after align_between's final translation), and the placement of the pair equals the
    oracle's placement of the same pair."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ivus_rest")
    raw, _ = mm.from_file_singlepair(gold, step_rotation_deg=1.0, range_rotation_deg=30.0, postprocessing=False,
                                     engine=engine)
    for g in (raw.geom_a, raw.geom_b):
        assert g.lumen_centroids is not None and g.has_lumen_centroid.all()
        for i in range(g.n_frames):
            assert tuple(g.lumen_centroids[i]) == mm.contour_centroid(g.frame_lumen(i))
    # the reference's defaults (postprocessing=True): both geometries trimmed to the common frame range
    # around the reference frame and resampled to one z spacing; walls present
    pair, _ = mm.from_file_singlepair(gold, step_rotation_deg=1.0, range_rotation_deg=30.0, engine=engine)
    # write_obj=True is the default, as in the reference: start + end geometry per contour type (entry.rs:655-670)
    written = sorted(os.listdir(os.path.join("output", "singlepair")))
    assert written == sorted(f"{k}_{i:03d}_{pair.label}.{e}" for k in ("lumen", "catheter", "wall") for i in (0, 1)
                             for e in ("obj", "mtl", "png"))
    assert pair.geom_a.n_frames == pair.geom_b.n_frames <= min(raw.geom_a.n_frames, raw.geom_b.n_frames)
    assert np.allclose(pair.geom_a.centroids[:, 2], pair.geom_b.centroids[:, 2], atol=1e-9)
    assert np.allclose(np.diff(pair.geom_a.centroids[:, 2]), np.diff(pair.geom_a.centroids[:, 2])[0], atol=1e-9)
    for g in (pair.geom_a, pair.geom_b):
        assert g.ids.tolist() == list(range(g.n_frames)) and g.has_ref.sum() == 1
        assert g.meta["extra_counts"]["wall"].tolist() == [int(g.lumen_off[1])] * g.n_frames
        assert np.allclose(g.lumen_centroids, [mm.contour_centroid(g.frame_lumen(i)) for i in range(g.n_frames)], atol=1e-9)
    stale, _ = mm.from_file_single(gold, smooth=False, step_rotation_deg=1.0, range_rotation_deg=30.0, engine=engine)
    # smooth = False: Frame.lumen.centroid is what the chain's last Frame::translate left (the reference never recomputes it
    # after that); carried exactly -- see test_golden_and_api.py::test_lumen_contour_centroid_is_carried_like_the_reference
    assert stale.lumen_centroids is not None and np.array_equal(stale.lumen_centroids[0], stale.centroids[0])
    a = pair.geom_a
    s = np.arange(0.0, 40.0, 0.25)
    cl = mm.Centerline.from_contour_points(np.stack([20.0 + 4.0 * np.sin(s / 11.0), -150.0 + 0.2 * s, 900.0 - 0.9 * s], axis=1))
    ref = cl.xyz()[10]
    out, sp, rot = mm.align_manual(cl, pair, 25.0, ref, write=True, output_dir="aligned", case_name="case7")
    assert "lumen_001_case7.obj" in os.listdir("aligned") and "wall_000_case7.png" in os.listdir("aligned")
    oa, ob = to_oracle(oracle, pair.geom_a), to_oracle(oracle, pair.geom_b)
    osp, orot = ocl.align_manual(to_oracle_cl(ocl, cl), [oa, ob], 25.0, ref)
    assert sp == osp and rot == orot * (180.0 / math.pi)
    assert geoms_equal(out.geom_a, oa) and geoms_equal(out.geom_b, ob)
    assert sp == pytest.approx(np.linalg.norm(np.diff(a.centroids, axis=0), axis=1).mean(), rel=1e-12)
