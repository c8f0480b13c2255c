"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares;
host logic (candidate enumeration, set construction, frame transforms) is bit-identical to
the oracle.  No compute entry point is called (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import refgeom
from helpers import to_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", hdr))


def test_header_symbols_exported(built, mm):
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["mm_build.h", "mm_ccta.h", "mm_centerline.h", "mm_hausdorff.h"]
    L = mm._native.lib()
    for header, exports in (("mm_hausdorff.h", mm._native.EXPORTS), ("mm_centerline.h", mm._native.EXPORTS_CENTERLINE),
                            ("mm_ccta.h", mm._native.EXPORTS_CCTA), ("mm_build.h", mm._native.EXPORTS_BUILD)):
        names = _declared(header)
        assert len(names) >= 4
        for n in sorted(names):
            assert hasattr(L, n), f"{n} declared in include/{header} but not exported"
        assert names == set(exports)
    assert b"gfx950" in L.mm_version()


def test_no_cpu_fallback(built, mm):
    if mm.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mm.Engine()


@pytest.mark.parametrize("step,rng,center,limes", [
    (1.0, 180.0, None, 180.0), (0.5, 180.0, None, 180.0), (0.5, 90.0, None, 90.0), (0.05, 90.0, None, 90.0),
    (0.01, 6.0, None, 6.0), (0.5, 5.0, 0.3, 90.0), (0.1, 5.0, -1.55, 90.0), (0.01, 0.1, 3.1, 180.0),
    (1.0, 180.0, None, 90.0), (0.0, 90.0, 1.0, 180.0), (-1.0, 90.0, 0.5, 180.0), (1.0, 0.0, None, 0.0),
    (7.0, 45.0, 0.2, 30.0), (0.3, 20.0, -0.1, 20.0),
])
def test_search_angles_match_oracle(built, mm, oracle, step, rng, center, limes):
    a, d, e = mm.search_angles(step, rng, center, limes)
    oa, od, oe = oracle.search_angles(step, rng, center, limes)
    assert d == od and e == oe
    assert np.array_equal(a, oa)


def test_search_set_and_between_points_match_oracle(built, mm, oracle):
    g = mm.synthetic_pullback(9, 501, pullback_id=1)
    og = to_oracle(oracle, g)
    for ss in (6, 200, 500, 501, 600):
        for i in (0, 3, 8):
            assert np.array_equal(mm.search_set(g, i, ss), oracle.catheter_lumen_vec(og, i, ss)[:, :2])
    for ss in (6, 500, 800):
        assert np.array_equal(mm.between_points(g, ss), oracle.extract_between_points(og, ss)[:, :2])
    assert mm.search_set(g, 0, 500).shape == (520, 2) and mm.search_set(g, 0, 501).shape == (521, 2)


def test_frame_transforms_match_oracle(built, mm, oracle):
    g = mm.synthetic_pullback(4, 64, pullback_id=2)
    og = to_oracle(oracle, g)
    s = g.c_struct()
    L = mm._native.lib()
    L.mm_frame_translate(C.byref(s), 1, 0.3, -0.2, 1.5)
    oracle.frame_translate(og, 1, 0.3, -0.2, 1.5)
    L.mm_frame_rotate(C.byref(s), 1, 0.37, g.centroids[1, 0], g.centroids[1, 1])
    oracle.frame_rotate(og, 1, 0.37, og.centroids[1, 0], og.centroids[1, 1])
    L.mm_frame_rotate(C.byref(s), 2, 0.0, 1.0, 1.0)  # angle == 0.0 -> untouched (contour_point.rs:39-41)
    oracle.frame_rotate(og, 2, 0.0, 1.0, 1.0)
    L.mm_frame_rotate(C.byref(s), 0, -2.1, 4.4, 4.6)  # frame 0 carries the reference point
    oracle.frame_rotate(og, 0, -2.1, 4.4, 4.6)
    for a, b in ((g.lumen, og.lumen), (g.cath, og.cath), (g.centroids, og.centroids), (g.ref, og.ref)):
        assert np.array_equal(a, b)


def test_dummy_geometry_builders():
    f = refgeom.dummy_frames()
    assert [fr.id for fr in f] == [0, 1, 2] and f[0].ref == [3.0, 1.0, 0.0]
    assert f[1].centroid[0] == pytest.approx(2.0) and f[2].centroid[1] == pytest.approx(19.0 / 6.0)
    g = refgeom.dummy_aligned_long_frames()
    assert len(g) == 6 and g[3].ref is None and g[0].ref is not None
    assert g[1].pts[0][0] == pytest.approx(1.0, abs=1e-6) and g[1].pts[0][1] == pytest.approx(3.0, abs=1e-6)
    assert [fr.centroid[2] for fr in g] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]


def test_refine_helpers_match_oracle(built, mm, oracle):
    """Host pieces of refine_alignment_hausdorff (align_algorithms.rs:339-451)."""
    for init, rng, step in [(0.0, 0.5235987755982988, 0.017453292519943295), (0.3, 0.1, 0.03), (-1.0, 0.0, 0.1),
                            (0.0, 1.0471975511965976, 0.017453292519943295)]:
        a, oa = mm.refine_angles(init, rng, step), oracle.refine_angles(init, rng, step)
        assert np.array_equal(a, oa) and len(a) >= 1
    assert len(mm.refine_angles(0.0, 0.2617993877991494, 0.017453292519943295)) in (30, 31)   # +-15 deg at 1 deg
    rng = np.random.default_rng(3)
    pts = rng.uniform(-20, 20, size=(5000, 3))
    s, e = np.array([1.0, -2.0, 3.0]), np.array([4.0, 6.0, -1.0])
    idx = mm.filter_points_in_region(pts, s, e)
    assert np.array_equal(idx, oracle.filter_points_in_region(pts, s, e))
    assert 0 < len(idx) < 5000
    lo, hi = np.minimum(s, e) - 5.0, np.maximum(s, e) + 5.0
    assert np.array_equal(idx, np.nonzero(np.all((pts >= lo) & (pts <= hi), axis=1))[0])
    for nf, m, f in [(0, 501, 20), (1, 501, 20), (3000, 501, 20), (10020, 501, 20), (50000, 501, 20), (777, 200, 11)]:
        assert mm.refine_downsample_count(nf, m, f) == oracle.refine_downsample_count(nf, m, f)
    assert mm.refine_downsample_count(50000, 501, 20) == 501 and mm.refine_downsample_count(0, 501, 20) == 1


def test_rotation_uses_one_sincos_call(built, mm, oracle):
    """The reference's `angle.cos()` / `angle.sin()` pair compiles to ONE glibc `sincos` call on
    x86_64-linux-gnu, and sincos differs from separate sin()/cos() by 1 ulp for some arguments
    (found at config3 scale: cumulative chain rotations of -23.17, -32.35, -67.66, -85.02 rad).
    Host C++ and oracle must both follow sincos, at exactly those arguments."""
    import math
    from multimoda_rs_amd._libm import sincos
    args = [-23.169245820224717, -32.349677685714894, -67.65768845356021, -85.02371451090376, -142.50613342533705]
    assert any((math.sin(a), math.cos(a)) != sincos(a) for a in args)      # the trap is real on this libm
    g = mm.synthetic_pullback(len(args) + 1, 64, pullback_id=3)
    og = to_oracle(oracle, g)
    s = g.c_struct()
    L = mm._native.lib()
    for i, a in enumerate(args):
        cx, cy = float(g.centroids[i + 1, 0]) + 0.3, float(g.centroids[i + 1, 1]) - 0.2
        L.mm_frame_rotate(C.byref(s), i + 1, a, cx, cy)
        oracle.frame_rotate(og, i + 1, a, cx, cy)
        si, co = sincos(a)
        x, y = g.frame_lumen(0)[0, 0], g.frame_lumen(0)[0, 1]
    for arr, oarr in ((g.lumen, og.lumen), (g.cath, og.cath), (g.centroids, og.centroids)):
        assert np.array_equal(arr, oarr)
    # and the result is the sincos one, not the sin()/cos() one
    base = mm.synthetic_pullback(len(args) + 1, 64, pullback_id=3)
    for i, a in enumerate(args):
        cx, cy = float(base.centroids[i + 1, 0]) + 0.3, float(base.centroids[i + 1, 1]) - 0.2
        si, co = sincos(a)
        p = base.frame_lumen(i + 1)
        ex = (p[:, 0] - cx) * co - (p[:, 1] - cy) * si + cx
        assert np.array_equal(g.frame_lumen(i + 1)[:, 0], ex)


def test_enumerations_terminate_on_steps_that_never_advance(built, mm):
    """A step below the spacing of the doubles it is added to never advances; the reference's loops would not
    terminate (`while angle <= ..{ angle += step }`, align_algorithms.rs:386-439; search_range's take_while). Here the
    refine enumeration fails after 2^22 angles and a candidate list of more than 2^24 entries is refused before it
    is allocated."""
    with pytest.raises(RuntimeError, match="more than 2\\^22 angles"):
        mm.refine_angles(0.0, 1.0, 1e-300)
    L = mm._native.lib()
    deg, early = C.c_int(0), C.c_double(0.0)
    n = L.mm_search_angles(1e-300, 1.0, 1, 1.0, 180.0, None, 0, C.byref(deg), C.byref(early))   # centre 1: start + i*step == start
    assert n == -3 and b"2^24 candidates" in L.mm_last_error()               # MM_ERR_TOO_LARGE
    with pytest.raises(RuntimeError, match="2\\^24 candidates"):
        mm.search_angles(1e-300, 1.0, center=1.0, limes_deg=180.0)
    assert len(mm.search_angles(0.0001, 180.0)[0]) == 3600001                # a fine but finite grid still enumerates
    assert mm.refine_downsample_count(10, 0, 0) == 0 and mm.refine_downsample_count(0, 5, 0) == 1   # NaN / inf saturate like `as usize`


def test_generated_asm_of_the_matrix_screen_is_current(tmp_path):
    """csrc/mm_screen_mx_asm.inc is generated by tools/gen_screen_mx.py and committed: the committed text must be what the
    generator writes today (an edit to one without the other would ship a kernel nobody can regenerate)."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "multimoda-rs_amd", "csrc", "mm_screen_mx_asm.inc")
    committed = open(inc).read()
    work = tmp_path / "repo"
    (work / "tools").mkdir(parents=True)
    (work / "multimoda-rs_amd" / "csrc").mkdir(parents=True)
    shutil.copy(os.path.join(root, "tools", "gen_screen_mx.py"), work / "tools" / "gen_screen_mx.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MX_DBG")}
    subprocess.check_call([sys.executable, str(work / "tools" / "gen_screen_mx.py")], env=env, stdout=subprocess.DEVNULL)
    assert open(work / "multimoda-rs_amd" / "csrc" / "mm_screen_mx_asm.inc").read() == committed
