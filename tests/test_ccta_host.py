"""CPU-side tests of the host-only entry points of include/mm_ccta.h against the oracle (bit-exact)."""
import numpy as np
import pytest

from helpers import to_oracle_cl


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="module")
def occ(oracle):
    from oracle import oracle_ccta, oracle_cl
    oracle_ccta.lib()
    return oracle_ccta, oracle_cl


def test_diameter_morphing_matches_oracle(built, mm, occ):
    oc, ocl = occ
    case = mm.synth.synthetic_tube_case(n_points=2500, n_reference=10, seed=1)
    pts = case["points"].copy()
    pts[3] = case["centerline"].xyz()[7]                     # a point ON the centerline: it must not move
    ocl_cl = to_oracle_cl(ocl, case["centerline"])
    for adj in (-2.0, -0.3, 0.0, 0.1, 1.7000000000000002):
        got = mm.adjust_diameter_centerline_morphing_simple(case["centerline"], pts, adj)
        assert np.array_equal(got, oc.diameter_morphing(ocl_cl, pts, adj))
        assert np.array_equal(got[3], pts[3])
    # radial displacement by the requested amount
    moved = mm.adjust_diameter_centerline_morphing_simple(case["centerline"], pts, 0.5)
    assert np.allclose(np.linalg.norm(moved - pts, axis=1)[np.arange(len(pts)) != 3], 0.5, atol=1e-12)
    with pytest.raises(RuntimeError, match="empty centerline"):
        mm.adjust_diameter_centerline_morphing_simple(mm.Centerline(case["centerline"].points[:0].copy()), pts, 0.5)


def test_wall_scaling_matches_oracle(built, mm, occ):
    oc, ocl = occ
    case = mm.synth.synthetic_tube_case(n_points=50, n_reference=4000, seed=2)
    ocl_cl = to_oracle_cl(ocl, case["centerline"])
    rng = np.random.default_rng(0)
    for _ in range(25):
        ref = case["points"][rng.integers(0, 50)] + rng.normal(0, 0.5, 3)
        assert mm.find_aortic_wall_scaling(case["centerline"], ref, case["reference"]) == \
            oc.wall_diameter_optimization(ocl_cl, ref, case["reference"])
    assert mm.find_aortic_wall_scaling(case["centerline"], (0, 0, 0), np.zeros((0, 3))) == 0.0
