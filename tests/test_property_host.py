"""Randomised parity of the host code behind the C ABI (hypothesis): the reader + geometry builder against the
independent restatement in tests/refbuild.py on generated CSV directories, and the frame bookkeeping
(mm_frames_finish_within / mm_frames_postprocess_pair) against its Python checker on generated pullbacks with
holes, measured wall thicknesses and EEM contours.  Host only; every comparison is bit for bit."""
import math
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import refbuild
from test_native_frames import assert_same

# derandomised and without an example database: the suite runs the same examples every time (MM_HYP_SCALE widens it)
SET = dict(deadline=None, derandomize=True, database=None,
           suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
SCALE = int(os.environ.get("MM_HYP_SCALE", "1"))      # MM_HYP_SCALE=20: a longer soak of the same properties


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


def _fmt(rng, v):
    """One of the float spellings Rust's str::parse::<f64> and the product's reader must agree on."""
    k = int(rng.integers(0, 5))
    if k == 0:
        return repr(float(v))
    if k == 1:
        return f"{v:.6f}"
    if k == 2:
        return f"{v:.9e}"
    if k == 3:
        return f"{v:.3f}".rstrip("0").rstrip(".") or "0"          # integers without a dot
    return ("+" if v >= 0 else "") + f"{v:.12g}"


@settings(max_examples=40 * SCALE, **SET)
@given(seed=st.integers(0, 2**31 - 1), n_frames=st.integers(1, 9), m=st.integers(3, 24), tab=st.booleans(),
       with_records=st.booleans(), diastole=st.booleans())
def test_reader_and_builder_equal_the_independent_builder(built, mm, tmp_path_factory, seed, n_frames, m, tab, with_records,
                                                          diastole):
    rng = np.random.default_rng(seed)
    d = tmp_path_factory.mktemp("csv")
    sep = "\t" if tab else ","
    frames = sorted(int(f) for f in rng.choice(np.arange(1, 60), size=n_frames, replace=False))
    phase = "diastolic" if diastole else "systolic"
    rows = []
    for f in frames:
        t = np.sort(rng.uniform(0, 2 * math.pi, m))
        r = 2.0 + 0.4 * rng.standard_normal()
        z = round(float(0.37 * f + rng.uniform(0, 0.01)), 4)
        zs = _fmt(rng, z)
        for a in t:
            x, y = 4.5 + r * math.cos(a) + 0.05 * rng.standard_normal(), 4.4 + 0.8 * r * math.sin(a) + 0.05 * rng.standard_normal()
            rows.append(sep.join([str(f), _fmt(rng, x), _fmt(rng, y), zs]))
    order = rng.permutation(len(rows)) if rng.integers(0, 2) else np.arange(len(rows))    # rows of a frame need not be adjacent
    junk = ["", sep.join(["x", "1", "2", "3"]), sep.join(["7.0", "1", "2", "3"]), sep.join(["3", "1", "2"]),
            sep.join(["3", "1", "2", "3", "true"])]     # a valid 5-field record in a 4-field file: another width than the first row's
    lines = [rows[i] for i in order]
    for j in junk:                                                 # rows the reader must skip, anywhere but first
        lines.insert(int(rng.integers(1, len(lines) + 1)), j)
    (d / f"{phase}_contours.csv").write_text("\n".join(lines) + "\n")
    rf = frames[int(rng.integers(0, len(frames)))]
    (d / f"{phase}_reference_points.csv").write_text(sep.join([str(rf), _fmt(rng, 6.1), _fmt(rng, 4.2), _fmt(rng, 0.37 * rf)]) + "\n")
    if with_records:
        rec = ["frame" + sep + "phase" + sep + "measurement_1" + sep + "measurement_2"]
        pool = frames + [int(f) for f in rng.integers(60, 70, size=2)]
        for f in rng.choice(pool, size=int(rng.integers(1, len(pool) + 2)), replace=True):
            rec.append(sep.join([str(int(f)), str(rng.choice(["D", "S", "X"])),
                                 "" if rng.integers(0, 3) == 0 else _fmt(rng, rng.uniform(0.5, 3)),
                                 "" if rng.integers(0, 3) == 0 else _fmt(rng, rng.uniform(0.5, 3))]))
        (d / "combined_sorted_manual.csv").write_text("\n".join(rec) + "\n")
    n_points = int(rng.choice([0, 5, 20]))
    g = mm.build_geometry_from_inputdata(None, str(d), "x", diastole, n_points=n_points)
    b = refbuild.build_geometry(str(d), diastole, n_points=n_points)
    assert list(g.ids) == b["ids"] and list(g.orig_frames) == b["orig_frames"]
    assert np.array_equal(g.centroids, np.array(b["centroids"]))
    for i in range(len(b["ids"])):
        assert np.array_equal(g.frame_lumen(i), b["lumens"][i]), i
        if n_points:
            assert np.array_equal(g.frame_cath(i), b["catheters"][i]), i
    assert {i for i in range(g.n_frames) if g.has_ref[i]} == set(b["ref_points"])
    for i, p in b["ref_points"].items():
        assert list(g.ref[i]) == p


def _pullback(mm, rng, F, m, holes, thick, eem):
    """A pullback with `holes` missing original frame indices in the middle, optional measured thicknesses / EEM."""
    s = mm.synthetic_pullback(F, m, pullback_id=int(rng.integers(0, 4)), seed=int(rng.integers(0, 1000)))
    orig = np.arange(F, dtype=np.int64) * 1
    for h in sorted(holes, reverse=True):
        orig[orig > h] += 1                        # a gap in the original frame numbering: fill_holes inserts a frame
    # the builder leaves frames proximal-first = descending original index
    s.orig_frames[:] = orig[::-1].astype(np.uint32)
    zero = lambda: np.zeros(F, dtype=np.int64)
    s.meta["extra_counts"] = {"eem": zero(), "calcification": zero(), "sidebranch": zero(), "wall": zero()}
    s.meta["aortic_thickness"] = [float(rng.uniform(0.5, 1.2)) if thick and rng.integers(0, 2) else None for _ in range(F)]
    s.meta["pulmonary_thickness"] = [float(rng.uniform(0.8, 1.6)) if thick and rng.integers(0, 3) == 0 else None for _ in range(F)]
    if eem:
        L = s.lumen.reshape(F, m, 3)
        c = s.centroids[:, None, :]
        e = c + (L - c) * float(rng.uniform(1.15, 1.5))
        e[..., 2] = L[..., 2]
        s.extra = np.ascontiguousarray(e.reshape(-1, 3))
        s.extra_off = np.arange(F + 1, dtype=np.int64) * m
        s.meta["extra_counts"]["eem"] = np.full(F, m, dtype=np.int64)
    return s


@settings(max_examples=25 * SCALE, **SET)
@given(seed=st.integers(0, 2**31 - 1), F=st.integers(3, 10), m=st.sampled_from([8, 12, 30, 41]), n_holes=st.integers(0, 2),
       thick=st.booleans(), eem=st.booleans(), smooth=st.booleans())
def test_finish_within_and_postprocess_pair_equal_the_python_checker(built, mm, seed, F, m, n_holes, thick, eem, smooth):
    from multimoda_rs_amd import api
    rng = np.random.default_rng(seed)
    mp = pytest.MonkeyPatch()
    try:
        pair = []
        for _ in range(2):
            holes = sorted(int(h) for h in rng.choice(np.arange(1, max(F - 1, 2)), size=min(n_holes, max(F - 2, 0)), replace=False))
            g = _pullback(mm, rng, F, m, holes, thick, eem)
            ref_idx = int(np.nonzero(g.has_ref)[0][0]) if g.has_ref.any() else 0
            nat, py = g.copy(), g.copy()
            a_n = api._finish_within(nat, ref_idx, smooth)
            mp.setenv("MM_PY_POSTPROC", "1")
            a_p = api._finish_within(py, ref_idx, smooth)
            mp.delenv("MM_PY_POSTPROC")
            assert a_n == a_p
            assert_same(nat, py, "finish_within")
            pair.append((nat, a_n))
        (ga, an_a), (gb, an_b) = pair
        gb = gb.copy()
        gb.lumen[:, 2] += float(rng.uniform(-0.4, 0.4)); gb.centroids[:, 2] = gb.lumen.reshape(gb.n_frames, -1, 3)[:, 0, 2]
        anomalous = bool(an_a or an_b)
        pr = api.GeometryPair(ga, gb, "p")
        out_n = api._maybe_postprocess(pr, anomalous, True)
        mp.setenv("MM_PY_POSTPROC", "1")
        out_p = api._maybe_postprocess(api.GeometryPair(ga, gb, "p"), anomalous, True)
        mp.delenv("MM_PY_POSTPROC")
        assert_same(out_n.geom_a, out_p.geom_a, "postprocess a")
        assert_same(out_n.geom_b, out_p.geom_b, "postprocess b")
    finally:
        mp.undo()


@settings(max_examples=25 * SCALE, **SET)
@given(seed=st.integers(0, 2**31 - 1), F=st.integers(2, 9), m=st.sampled_from([6, 12, 33, 64]), thick=st.booleans(),
       eem=st.booleans(), angle=st.sampled_from([0.0, 17.0, -52.0, 180.0, 271.5]), walls=st.booleans(),
       pair=st.booleans(), short_cl=st.booleans())
def test_placement_with_walls_equals_the_oracle(built, mm, oracle, seed, F, m, thick, eem, angle, walls, pair, short_cl):
    """align_manual (align.rs:126-166) = preprocess_centerline + rotate_geometry + apply_transformations
    [+ align_walls] on generated pullbacks (measured thicknesses -> aortic flags on lumen and wall, EEM -> offset
    walls), single geometries and pairs, centerlines shorter than the pullback (trailing frames keep their place):
    coordinates, frame / contour centroids and the per-point flags equal oracle/mm_oracle_cl.c bit for bit."""
    from oracle import oracle_cl as ocl
    from helpers import geoms_equal, to_oracle, to_oracle_cl
    from multimoda_rs_amd import native_frames as NF
    ocl.lib()
    rng = np.random.default_rng(seed)

    def make():
        g = _pullback(mm, rng, F, m, [], thick, eem)
        g, _an = NF.finish_within(g, int(np.nonzero(g.has_ref)[0][0]) if g.has_ref.any() else 0, bool(rng.integers(0, 2)))
        if not walls:                                         # drop the Wall contours again: align_walls has nothing to do
            cnt = g.meta["extra_counts"]
            keep = np.concatenate([np.r_[np.ones(int(e), bool), np.zeros(int(w), bool)]
                                   for e, w in zip(cnt["eem"], cnt["wall"])]) if g.extra is not None else None
            if keep is not None and keep.any():
                g.extra = np.ascontiguousarray(g.extra[keep])
                g.extra_off = np.concatenate([[0], np.cumsum(cnt["eem"])]).astype(np.int64)
            else:
                g.extra, g.extra_off = None, None
            cnt["wall"] = np.zeros(g.n_frames, dtype=np.int64)
            g.meta.pop("wall_aortic", None)
        return g
    ga = make()
    target = mm.GeometryPair(ga, make(), "p") if pair else ga
    n_cl = (ga.n_frames + 6) if not short_cl else max(ga.n_frames - 2, 3)
    s = np.arange(0.0, 0.5 * n_cl, 0.25)
    path = np.stack([12.0 + 6.0 * np.sin(s / 9.0), -200.0 + 5.0 * np.cos(s / 11.0), 1750.0 - 0.9 * s], axis=1)
    cl = mm.Centerline.from_contour_points(path)
    ref = path[int(rng.integers(0, 3))] + rng.normal(0, 0.05, 3)
    out, sp, rot = mm.align_manual(cl, target, angle, ref, align_wall_anomalous=True)
    gs = [target.geom_a, target.geom_b] if pair else [target]
    ogs = [to_oracle(oracle, g) for g in gs]
    osp, orot = ocl.align_manual(to_oracle_cl(ocl, cl), ogs, angle, ref, align_wall_anomalous=True)
    assert sp == osp and rot == orot * (180.0 / math.pi)
    outs = [out.geom_a, out.geom_b] if pair else [out]
    for o, og in zip(outs, ogs):
        assert geoms_equal(o, og)
        for key in ("lumen_aortic", "wall_aortic"):
            have = getattr(og, key)
            if have is not None:
                assert np.array_equal(np.asarray(o.meta[key], dtype=np.uint8), have), key


@settings(max_examples=150 * SCALE, **SET)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 70), kind=st.sampled_from(["sorted", "rotated", "random", "two_wraps"]),
       dup_at_wrap=st.booleans(), dups=st.booleans())
def test_sort_contour_points_fast_paths_equal_the_oracle(built, mm, oracle, seed, n, kind, dup_at_wrap, dups):
    """Contour::sort_contour_points (contour.rs:368-405): contours that arrive in angular order -- ascending keys,
    or ascending with one wrap -- skip the sort (csrc/mm_sort.h); ties that straddle the wrap, duplicate points and
    sequences with two descents must still come out in the stable order of the oracle's full sort."""
    from oracle import oracle_cl as ocl
    ocl.lib()
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-math.pi, math.pi, n))
    if dups and n >= 4:
        t[1] = t[2]                                   # equal keys inside a run
    r = rng.uniform(0.5, 2.0, n) if rng.integers(0, 2) else np.full(n, 1.5)
    p = np.stack([r * np.cos(t), r * np.sin(t), rng.normal(0, 1, n)], axis=1)
    # shift so that the xy mean (the sort centre) is the origin the angles were drawn about -- approximately; the
    # parity below does not depend on it, only how often the fast paths are taken does
    if kind in ("rotated", "two_wraps") and n >= 2:
        k = int(rng.integers(1, n))
        p = np.concatenate([p[k:], p[:k]])
        if dup_at_wrap:
            p[-1, :2] = p[0, :2]                      # last and first point coincide: a tie across the wrap
    if kind == "two_wraps" and n >= 6:
        i, j = sorted(int(v) for v in rng.choice(n, 2, replace=False))
        p[[i, j]] = p[[j, i]]
    if kind == "random":
        p = p[rng.permutation(n)]
    p = np.ascontiguousarray(p)
    q = p.copy()
    assert mm._native.lib().mm_sort_contour_points(mm._native._ptr(q), n) == 0
    assert np.array_equal(q, ocl.sort_contour_points(p))
