"""The bookkeeping around the searches behind the C ABI (include/mm_build.h: mm_frames_finish_within,
mm_frames_postprocess_pair; csrc/mm_frames.cpp) against the Python implementation it replaces (api._finish_within,
tests/mm_checkers: postproc.postprocess_pair / postproc_flat.postprocess_pair_regular), which restates the reference's own tests in
tests/test_postproc.py.  Host only; bit for bit."""
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FIELDS = ("ids", "lumen_ids", "orig_frames", "centroids", "lumen_off", "lumen", "cath_off", "cath", "extra_off", "extra",
          "has_ref", "ref", "has_lumen_centroid", "lumen_centroids")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


def assert_same(a, b, what=""):
    for n in FIELDS:
        x, y = getattr(a, n), getattr(b, n)
        assert (x is None) == (y is None), (what, n)
        if x is not None:
            x, y = np.asarray(x), np.asarray(y)
            # a degenerate contour may give NaN coordinates on both sides (0/0 in the reference's arithmetic too):
            # the same positions must be NaN, everything else identical
            assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), (what, n)
    for k in ("eem", "calcification", "sidebranch", "wall"):
        assert np.array_equal(a.meta["extra_counts"][k], b.meta["extra_counts"][k]), (what, k)
    for k in ("aortic_thickness", "pulmonary_thickness"):
        assert list(a.meta[k]) == list(b.meta[k]), (what, k)
    for k in ("lumen_aortic", "wall_aortic"):
        assert (k in a.meta) == (k in b.meta), (what, k)
        if k in a.meta:
            assert np.array_equal(np.asarray(a.meta[k], dtype=bool), np.asarray(b.meta[k], dtype=bool)), (what, k)


def python_finish(mm, g, ref_idx, smooth, monkeypatch):
    from multimoda_rs_amd import api
    monkeypatch.setenv("MM_PY_POSTPROC", "1")
    h = g.copy()
    an = api._finish_within(h, ref_idx, smooth)
    monkeypatch.delenv("MM_PY_POSTPROC")
    return h, an


def native_finish(mm, g, ref_idx, smooth):
    from multimoda_rs_amd import api
    h = g.copy()
    an = api._finish_within(h, ref_idx, smooth)          # the product path: mm_frames_finish_within
    return h, an


def variants(mm):
    """(name, geometry, reference frame index)"""
    out = []
    for folder in ("ivus_rest", "idealized_geometry", "examples_ivus_rest"):
        for dia in (True, False):
            g = mm.build_geometry_from_inputdata(None, os.path.join(GOLD, folder), folder, dia)
            out.append((f"{folder}-{dia}", g, int(np.nonzero(g.has_ref)[0][0])))
    s = mm.synthetic_pullback(24, 120, pullback_id=1)
    s.meta.setdefault("extra_counts", {k: np.zeros(24, dtype=np.int64) for k in ("eem", "calcification", "sidebranch", "wall")})
    s.meta["aortic_thickness"] = [None] * 24
    s.meta["pulmonary_thickness"] = [None] * 24
    out.append(("synthetic", s, 0))
    t = s.copy()                                     # measured thicknesses on some frames -> anomalous, aortic walls
    t.meta["aortic_thickness"] = [0.9 if i % 3 == 0 else None for i in range(24)]
    t.meta["pulmonary_thickness"] = [1.4 if i % 4 == 0 else None for i in range(24)]
    out.append(("thickness", t, 0))
    e = s.copy()                                     # an EEM contour per frame: walls from the EEM (non-anomalous)
    F, m = e.n_frames, 120
    L = e.lumen.reshape(F, m, 3)
    c = e.centroids[:, None, :]
    eem = c + (L - c) * 1.3
    eem[..., 2] = L[..., 2]
    e.extra = np.ascontiguousarray(eem.reshape(-1, 3))
    e.extra_off = np.arange(F + 1, dtype=np.int64) * m
    e.meta["extra_counts"] = {"eem": np.full(F, m, dtype=np.int64), "calcification": np.zeros(F, dtype=np.int64),
                              "sidebranch": np.zeros(F, dtype=np.int64), "wall": np.zeros(F, dtype=np.int64)}
    out.append(("eem", e, 0))
    # holes: drop frames so that gaps of 2, 3 and 5 spacings appear (one averaged frame, two and four interpolated ones)
    keep = [i for i in range(24) if i not in (4, 9, 10, 15, 16, 17, 18)]
    h = mm.FlatGeometry.from_frames([s.frame_lumen(i) for i in keep], catheters=[s.frame_cath(i) for i in keep],
                                    centroids=[s.centroids[i] for i in keep], ids=list(range(len(keep))),
                                    orig_frames=[int(s.orig_frames[i]) for i in keep], ref_points={0: s.ref[0]})
    h.meta = {"extra_counts": {k: np.zeros(len(keep), dtype=np.int64) for k in ("eem", "calcification", "sidebranch", "wall")},
              "aortic_thickness": [0.8 if i in (3, 8) else None for i in range(len(keep))],
              "pulmonary_thickness": [None] * len(keep)}
    out.append(("holes", h, 0))
    return out


@pytest.mark.parametrize("smooth", [False, True])
def test_finish_within_native_equals_python(built, mm, smooth, monkeypatch):
    for name, g, ref_idx in variants(mm):
        hp, ap = python_finish(mm, g, ref_idx, smooth, monkeypatch)
        hn, an = native_finish(mm, g, ref_idx, smooth)
        assert ap == an, name
        assert_same(hn, hp, name)


def _pair(mm, monkeypatch, fa=30, fb=26, dz_b=0.5, thick=False, shift_ref=0):
    """Two finished geometries of one pullback pair."""
    from multimoda_rs_amd import api
    a = mm.synthetic_pullback(fa, 96, pullback_id=0)
    b = mm.synthetic_pullback(fb, 96, pullback_id=1)
    if dz_b != 0.5:                                  # a different sampling rate along z
        for arr in (b.lumen, b.cath, b.centroids, b.ref):
            arr[:, 2] = arr[:, 2] / 0.5 * dz_b
    if shift_ref:                                    # reference point on another frame
        for g in (a, b):
            g.has_ref[:] = 0
            g.has_ref[shift_ref] = 1
            g.ref[shift_ref] = g.frame_lumen(shift_ref)[5]
    for g, F in ((a, fa), (b, fb)):
        g.meta["extra_counts"] = {k: np.zeros(F, dtype=np.int64) for k in ("eem", "calcification", "sidebranch", "wall")}
        g.meta["aortic_thickness"] = [(0.7 + 0.01 * i) if thick and i % 2 == 0 else None for i in range(F)]
        g.meta["pulmonary_thickness"] = [None] * F
    monkeypatch.setenv("MM_PY_POSTPROC", "1")
    fl = [api._finish_within(g, int(np.nonzero(g.has_ref)[0][0]), True) for g in (a, b)]
    monkeypatch.delenv("MM_PY_POSTPROC")
    return a, b, any(fl)


@pytest.mark.parametrize("case", [dict(), dict(thick=True), dict(dz_b=0.8), dict(dz_b=0.3), dict(dz_b=0.3, thick=True),
                                  dict(shift_ref=7), dict(fa=12, fb=31, shift_ref=3, thick=True)])
def test_postprocess_pair_native_equals_python(built, mm, case, monkeypatch):
    from multimoda_rs_amd import frames as FR, native_frames as NF
    from mm_checkers import postproc as PP
    a, b, anomalous = _pair(mm, monkeypatch, **case)
    for an in (anomalous, not anomalous):
        fa, fb = PP.postprocess_pair(FR.to_frames(a), FR.to_frames(b), 0.03, an)
        pa, pb = FR.from_frames(fa, a.label, a.meta), FR.from_frames(fb, b.label, b.meta)
        na, nb = NF.postprocess_pair(a, b, 0.03, an)
        assert_same(na, pa, f"a {case} {an}")
        assert_same(nb, pb, f"b {case} {an}")


def test_native_errors_carry_the_reference_messages(built, mm):
    from multimoda_rs_amd import native_frames as NF
    from multimoda_rs_amd.centerline import with_lumen_centroids
    g = mm.synthetic_pullback(6, 64)
    g.meta["extra_counts"] = {k: np.zeros(6, dtype=np.int64) for k in ("eem", "calcification", "sidebranch", "wall")}
    g.meta["aortic_thickness"] = [None] * 6
    g.meta["pulmonary_thickness"] = [None] * 6
    with_lumen_centroids(g)
    with pytest.raises(RuntimeError, match="reference frame index out of range"):
        NF.finish_within(g, 9, True)
    g.has_ref[:] = 0
    with pytest.raises(RuntimeError, match="No reference point found in frame"):
        NF.finish_within(g, 0, True)
    with pytest.raises(RuntimeError, match="No reference point found in any frame"):
        NF.postprocess_pair(g, g, 0.03, False)


def test_failed_postprocess_pair_leaves_both_handles_as_they_were(built, mm):
    """every fallible step of mm_frames_postprocess_pair comes before the first move.  A is four times sparser
    than B and carries its reference point on the last frame: re-sampled to B's spacing that frame's index lies beyond A's
    original length -> the reference's `index out of bounds` (postprocessing.rs:70-76), which used to arrive AFTER B had
    been moved out of its handle.  A C / Rust host that inspects or retries must still find 8 and 40 frames."""
    from multimoda_rs_amd import native_frames as NF
    from multimoda_rs_amd.centerline import with_lumen_centroids

    def geom(n, dz, ref_at):
        g = mm.synthetic_pullback(n, 64)
        g.meta["extra_counts"] = {k: np.zeros(n, dtype=np.int64) for k in ("eem", "calcification", "sidebranch", "wall")}
        g.meta["aortic_thickness"] = [None] * n
        g.meta["pulmonary_thickness"] = [None] * n
        z = np.arange(n) * dz
        g.centroids[:, 2] = z
        for i in range(n):
            g.lumen[g.lumen_off[i]:g.lumen_off[i + 1], 2] = z[i]
            if g.cath_off is not None:
                g.cath[g.cath_off[i]:g.cath_off[i + 1], 2] = z[i]
        g.has_ref[:] = 0
        g.has_ref[ref_at] = 1
        g.ref[ref_at] = (g.centroids[ref_at, 0] + 1.0, g.centroids[ref_at, 1], z[ref_at])
        with_lumen_centroids(g)
        return g

    a, b = geom(8, 1.0, 7), geom(40, 0.25, 3)
    fa, fb = NF.stage_pair(a, b)
    try:
        before = (fa.to_flat(), fb.to_flat())
        with pytest.raises(RuntimeError, match="index out of bounds"):
            fa.postprocess_pair(fb, 0.03, False)
        after = (fa.to_flat(), fb.to_flat())
        for x, y in zip(before, after):
            assert x.n_frames == y.n_frames and x.n_frames in (40, 8)
            assert np.array_equal(x.lumen, y.lumen) and np.array_equal(x.centroids, y.centroids) and np.array_equal(x.ids, y.ids)
    finally:
        fa.close(); fb.close()
