"""The reference's own Python-level alignment tests, restated by name against this package
(/root/reference/tests/test_intravascular.py:95-236: TestAlignThreePoint x 5, TestAlignManual x 4) -- the only tests
the reference holds that drive row f1 (centerline placement + three-point / manual rotation) end to end on real data:
the idealized-geometry fixture (data/fixtures/idealized_geometry -> tests/golden/idealized_geometry) placed on
examples/data/centerline_raw.csv (-> tests/golden/examples_centerlines/centerline_raw.csv), with the reference points
of the align_three_point docstring (test_intravascular.py:18-20).

Same constructors (numpy_to_geometry / numpy_to_centerline), same calls, same assertions and tolerances; the calls go
through the C ABI (mm_align_three_point / mm_align_manual, include/mm_centerline.h).  Host-only code: runs in the CPU
suite.  What these tests pin is consistency (pair vs single, spacing, frame counts) -- the reference asserts no absolute
coordinates here, so the placement itself stays "loosely pinned" (DESIGN 2)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# test_intravascular.py:18-20
AORTIC_REF_PT = (12.2605, -201.3643, 1751.0554)
UPPER_REF_PT = (11.7567, -202.1920, 1754.7975)
LOWER_REF_PT = (15.6605, -202.1920, 1749.9655)


def _geom_to_points_array(geom) -> np.ndarray:
    """test_intravascular.py:23-29: all lumen points of all frames, frame by frame, as (N, 3)."""
    return np.concatenate([geom.frame_lumen(i) for i in range(geom.n_frames)], axis=0)


def _fixture_geometry(mm, phase):
    raw = np.genfromtxt(os.path.join(GOLD, "idealized_geometry", f"{phase}_contours.csv"), delimiter=",")
    ref = np.genfromtxt(os.path.join(GOLD, "idealized_geometry", f"{phase}_reference_points.csv"), delimiter=",")
    return mm.numpy_to_geometry(lumen_arr=raw, catheter_arr=np.zeros((0, 4)), wall_arr=np.zeros((0, 4)), reference_arr=ref)


@pytest.fixture(scope="module")
def geom_a(mm):
    return _fixture_geometry(mm, "diastolic")


@pytest.fixture(scope="module")
def geom_b(mm):
    return _fixture_geometry(mm, "systolic")


@pytest.fixture(scope="module")
def geometry_pair(mm, geom_a, geom_b):
    return mm.GeometryPair(geom_a=geom_a, geom_b=geom_b, label="test")


@pytest.fixture(scope="module")
def centerline(mm):
    arr = np.genfromtxt(os.path.join(GOLD, "examples_centerlines", "centerline_raw.csv"), delimiter=",")
    return mm.numpy_to_centerline(arr)


class TestAlignThreePoint:
    def _call(self, mm, centerline, g):
        return mm.align_three_point(centerline, g, AORTIC_REF_PT, UPPER_REF_PT, LOWER_REF_PT, write=False)

    def test_pair_returns_geometry_pair(self, mm, centerline, geometry_pair):
        result, _, _ = self._call(mm, centerline, geometry_pair)
        assert isinstance(result, mm.GeometryPair)

    def test_single_returns_geometry(self, mm, centerline, geom_a):
        result, _, _ = self._call(mm, centerline, geom_a)
        assert isinstance(result, mm.FlatGeometry)

    def test_pair_geom_a_matches_single(self, mm, centerline, geom_a, geometry_pair):
        """Aligning geom_a alone must produce the same points as geom_a inside a pair."""
        result_pair, _, _ = self._call(mm, centerline, geometry_pair)
        result_geom, _, _ = self._call(mm, centerline, geom_a)
        np.testing.assert_allclose(_geom_to_points_array(result_pair.geom_a), _geom_to_points_array(result_geom), atol=1e-10)

    def test_resampled_spacing_matches(self, mm, centerline, geom_a, geometry_pair):
        """Aligning geom_a alone must derive the same resample spacing as inside a pair."""
        _, spacing_pair, _ = self._call(mm, centerline, geometry_pair)
        _, spacing_geom, _ = self._call(mm, centerline, geom_a)
        assert spacing_pair == pytest.approx(spacing_geom, abs=1e-10)

    def test_frame_count_preserved(self, mm, centerline, geom_a, geometry_pair):
        n_frames = geom_a.n_frames
        result_pair, _, _ = self._call(mm, centerline, geometry_pair)
        result_geom, _, _ = self._call(mm, centerline, geom_a)
        assert result_pair.geom_a.n_frames == n_frames
        assert result_geom.n_frames == n_frames


class TestAlignManual:
    def _call(self, mm, centerline, g):
        return mm.align_manual(centerline, g, rotation_angle_deg=286.0, ref_point=AORTIC_REF_PT, write=False)

    def test_pair_returns_geometry_pair(self, mm, centerline, geometry_pair):
        result, _, _ = self._call(mm, centerline, geometry_pair)
        assert isinstance(result, mm.GeometryPair)

    def test_single_returns_geometry(self, mm, centerline, geom_a):
        result, _, _ = self._call(mm, centerline, geom_a)
        assert isinstance(result, mm.FlatGeometry)

    def test_pair_geom_a_matches_single(self, mm, centerline, geom_a, geometry_pair):
        """Aligning geom_a alone must produce the same points as geom_a inside a pair."""
        result_pair, _, _ = self._call(mm, centerline, geometry_pair)
        result_geom, _, _ = self._call(mm, centerline, geom_a)
        np.testing.assert_allclose(_geom_to_points_array(result_pair.geom_a), _geom_to_points_array(result_geom), atol=1e-10)

    def test_frame_count_preserved(self, mm, centerline, geom_a, geometry_pair):
        n_frames = geom_a.n_frames
        result_pair, _, _ = self._call(mm, centerline, geometry_pair)
        result_geom, _, _ = self._call(mm, centerline, geom_a)
        assert result_pair.geom_a.n_frames == n_frames
        assert result_geom.n_frames == n_frames


def test_against_the_oracle_on_the_same_data(mm, geom_a, geom_b, centerline):
    """Beyond the reference's assertions: the placement of the fixture on this centerline, bit for bit against
    oracle/mm_oracle_cl.c (both calls, single and pair), and the inputs untouched (by-value semantics, binding/align.rs)."""
    import math
    from helpers import geoms_equal, to_oracle, to_oracle_cl
    from oracle import oracle as orc, oracle_cl as ocl
    ocl.lib()
    before = _geom_to_points_array(geom_a).copy()
    ocl_cl = to_oracle_cl(ocl, centerline)
    for g_in in (geom_a, mm.GeometryPair(geom_a=geom_a, geom_b=geom_b, label="t")):
        srcs = [g_in] if isinstance(g_in, mm.FlatGeometry) else [g_in.geom_a, g_in.geom_b]
        for which in ("three_point", "manual"):
            ogs = [to_oracle(orc, s.copy()) for s in srcs]
            idx = int(srcs[0].meta.get("ref_point_index", 0))
            if which == "three_point":
                out, sp, rot = mm.align_three_point(centerline, g_in, AORTIC_REF_PT, UPPER_REF_PT, LOWER_REF_PT)
                osp, orot = ocl.align_three_point(ocl_cl, ogs, idx, AORTIC_REF_PT, UPPER_REF_PT, LOWER_REF_PT, math.radians(1.0))
            else:
                out, sp, rot = mm.align_manual(centerline, g_in, 286.0, AORTIC_REF_PT)
                osp, orot = ocl.align_manual(ocl_cl, ogs, 286.0, AORTIC_REF_PT)
            outs = [out] if isinstance(out, mm.FlatGeometry) else [out.geom_a, out.geom_b]
            assert sp == osp and rot == orot * (180.0 / math.pi), which
            for o, og in zip(outs, ogs):
                assert geoms_equal(o, og), which
    assert np.array_equal(_geom_to_points_array(geom_a), before)
