"""BASELINE config 5 in one piece, on synthetic data (the reference's examples/fullworkflow.py needs a CCTA
mesh that is not in the checkout and trimesh, which is not installed): register a diastolic / systolic
pullback pair (from_array_singlepair, reference defaults incl. post-processing and walls), place it on a
vessel centerline with align_combined (three-point start, GPU-scored Hausdorff refinement, wall twist
compensation), then run the three diameter searches on clouds derived from the placed geometry.  Every GPU
stage is checked against the oracle on the same inputs."""
import math

import numpy as np
import pytest

from helpers import geoms_equal, to_oracle, to_oracle_cl
from test_golden_and_api import _array_input

pytestmark = pytest.mark.gpu


def test_full_workflow_synthetic(engine, mm, oracle):
    from oracle import oracle_ccta as occ, oracle_cl as ocl
    occ.lib(); ocl.lib()
    dia = _array_input(mm, n_frames=16, n_points=120, thickness=0.9, seed=3)
    sys_ = _array_input(mm, n_frames=14, n_points=120, thickness=1.1, seed=4)
    sys_.diastole = False
    pair, (logs_d, logs_s) = mm.from_array_singlepair(dia, sys_, step_rotation_deg=1.0, range_rotation_deg=20.0,
                                                      engine=engine)
    a, b = pair.geom_a, pair.geom_b
    assert a.n_frames == b.n_frames and a.meta["anomalous"] and a.meta["extra_counts"]["wall"].sum() == a.lumen.shape[0]
    assert a.has_ref[0] == 1 and len(logs_d) == 15 and len(logs_s) == 13

    case = mm.synth.synthetic_centerline_case(geometry=a, n_ccta=2500, seed=9, true_rotation_deg=21.0, true_index=7)
    geo = mm.GeometryPair(case["geometry"], b, pair.label)
    aligned, spacing_mm, rot_deg = mm.align_combined(case["centerline"], geo, case["main_ref_pt"], case["ccw_ref_pt"],
                                                     case["cw_ref_pt"], case["points"], angle_range_deg=6.0,
                                                     align_wall_anomalous=True, engine=engine)
    # placement parity, without and with the wall step, and recovery of the pose
    plain, sp2, rot2 = mm.align_combined(case["centerline"], geo, case["main_ref_pt"], case["ccw_ref_pt"],
                                         case["cw_ref_pt"], case["points"], angle_range_deg=6.0, engine=engine)
    oa, ob = to_oracle(oracle, geo.geom_a), to_oracle(oracle, geo.geom_b)
    osp, orot, oidx = ocl.align_combined(to_oracle_cl(ocl, case["centerline"]), [oa, ob], geo.geom_a.meta["ref_point_index"],
                                         case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"], case["points"],
                                         math.radians(1.0), math.radians(6.0), 2)
    assert (sp2, rot2) == (osp, orot * (180.0 / math.pi)) == (spacing_mm, rot_deg)
    assert geoms_equal(plain.geom_a, oa) and geoms_equal(plain.geom_b, ob)
    wa, wb = to_oracle(oracle, geo.geom_a), to_oracle(oracle, geo.geom_b)
    assert wa.wall_kind1 > 0 and wa.wall_aortic is not None and wa.wall_aortic.any()
    ocl.align_combined(to_oracle_cl(ocl, case["centerline"]), [wa, wb], geo.geom_a.meta["ref_point_index"],
                       case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"], case["points"],
                       math.radians(1.0), math.radians(6.0), 2, align_wall_anomalous=True)
    assert geoms_equal(aligned.geom_a, wa) and geoms_equal(aligned.geom_b, wb)
    assert np.array_equal(aligned.geom_a.meta["wall_aortic"].astype(np.uint8), wa.wall_aortic)
    assert np.array_equal(aligned.geom_a.lumen, plain.geom_a.lumen)                  # align_walls moves walls only
    assert not np.array_equal(aligned.geom_a.extra, plain.geom_a.extra)
    assert rot_deg == pytest.approx(21.0, abs=1.0 + 1e-9) and abs(oidx - 7) <= 2
    assert spacing_mm == pytest.approx(np.linalg.norm(np.diff(a.centroids, axis=0), axis=1).mean(), rel=1e-12)

    # diameter searches on the placed geometry: CCTA-like clouds = the placed lumens pushed out / in
    g = aligned.geom_a
    rcl, _ = mm.preprocess_centerline(case["centerline"], a)
    cloud = mm.adjust_diameter_centerline_morphing_simple(rcl, g.lumen, 0.4)
    res = {"anomalous_points": cloud, "rca_removed_points": cloud[::3], "aorta_points": cloud[::5] + [6.0, 0.0, 0.0]}
    prox, dist = mm.find_distal_and_proximal_scaling(g, rcl, res, engine=engine)
    n4 = int(math.ceil(0.25 * len(cloud)))
    F = g.n_frames
    assert (prox, dist) == occ.diameter_optimization(cloud, n4, n4, to_oracle_cl(ocl, rcl), g.lumen[:g.lumen_off[2]],
                                                     g.lumen[g.lumen_off[F - 3]:])
    assert prox == pytest.approx(-0.4, abs=0.1 + 1e-9) and dist == pytest.approx(-0.4, abs=0.1 + 1e-9)
    ao = mm.find_aorta_scaling(g, rcl, res, engine=engine)
    wall_ref = mm.ccta._extract_wall_from_frames(g)
    assert wall_ref.shape == (60, 3)
    assert ao == occ.aortic_diameter_optimization(res["rca_removed_points"], wall_ref, to_oracle_cl(ocl, rcl))[0]
    # round lumens (elliptic ratio < 1.3) give the coronary reference point of the wall search
    try:
        w = mm.find_aortic_wall_scaling(g, rcl, res)
        assert w >= 0.0
    except ValueError as e:
        assert "No coronary reference point" in str(e)
