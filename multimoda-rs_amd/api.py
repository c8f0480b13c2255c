"""The reference's Python entry points for this path, mirrored: ``from_file_*`` /
``from_array_*`` in the full / double-pair / single-pair / single modes
(multimodars/_processing.py:42-1007 -> binding/functions.rs:143-1423 -> binding/entry.rs).

Orchestration follows entry.rs: build geometries (io.py), align frames within every pullback
(device search, decoupled mode), the post-steps of ``align_frames_in_geometry``
(align_within.rs:136-170: hole filling, reference point to the right, aortic flags, wall contours,
smoothing -- postproc.py), the between-pullback alignments in the reference's order (AB | CD, then
AC | BD), then ``postprocess_geom_pair`` per pair and, with ``write_obj`` (the default, as in the
reference), the OBJ / MTL / texture files (export.py).  Same argument names, meaning and defaults as the
reference.

Return values: ``FlatGeometry`` / ``GeometryPair`` (numpy containers) instead of the PyO3 value
classes; logs are the reference's 7-tuples ``(id, matched_to, rot_deg, tx, ty, cx, cy)``.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from . import geometry as G
from ._libm import sincos
from .io import InputData, build_geometry_from_inputdata

AlignLog = List[Tuple[int, int, float, float, float, float, float]]
TOLERANCE = 0.03  # entry.rs:21 (used by postprocessing only)

_engine: Optional[N.Engine] = None


def default_engine() -> N.Engine:
    """One lazily created engine on the current device (fails loudly without a GPU)."""
    global _engine
    if _engine is None or not _engine.handle.value:
        _engine = N.Engine()
    return _engine


def _checker(name: str):
    """A module of tests/mm_checkers (the Python implementations the native host code replaced; test infrastructure).  The
    MM_PY_POSTPROC / MM_PY_BUILDER switches of the test suite route through it; the product itself never does."""
    import importlib
    try:
        return importlib.import_module("mm_checkers." + name)
    except ImportError as e:
        raise RuntimeError("MM_PY_POSTPROC / MM_PY_BUILDER select the Python checkers of the test suite (tests/mm_checkers), "
                           "which are not part of the product: put tests/ on sys.path") from e


@dataclass
class GeometryPair:
    """types/native/geometry_pair.rs: geom_a (e.g. diastole / rest) and geom_b aligned onto it."""
    geom_a: G.FlatGeometry
    geom_b: G.FlatGeometry
    label: str = ""


def _with_contour_centroids(g: G.FlatGeometry) -> G.FlatGeometry:
    """Frame.lumen.centroid of a returned geometry (read by the centerline placement, centerline.py).
    The reference recomputes it as the mean of the points in smooth_frames (geometry.rs:204) and in
    every Frame::translate (frame.rs:20), and leaves it stale in Frame::rotate: it is the fresh mean
    after smoothing and for every geometry moved by align_between (which ends with a translation,
    align_between.rs:68).  Otherwise (smooth=False, reference side of a pair) it is the value the chain's
    last translation of the frame left -- the mean of the lumen BEFORE the step's rotation and before the
    post-step rotation; geometries built by ``build_geometry_from_inputdata`` carry it through the chain
    (``mm_geometry.lumen_centroid``) and the post-steps and return exactly that.  A geometry that entered
    without contour centroids (hand-made FlatGeometry) keeps None."""
    if g.meta.get("lumen_centroid_fresh"):
        from .centerline import with_lumen_centroids
        with_lumen_centroids(g)
    elif g.lumen_centroids is None or not g.meta.get("lumen_centroid_tracked"):
        g.has_lumen_centroid, g.lumen_centroids = None, None
    return g


def _make_pair(a: G.FlatGeometry, b: G.FlatGeometry) -> GeometryPair:
    """GeometryPair::new: label = "<a> - <b>" (the labels the reference's plumbing tests check,
    binding/functions.rs:1607-1616)."""
    return GeometryPair(_with_contour_centroids(a), _with_contour_centroids(b), f"{a.label} - {b.label}")


# ---------------------------------------------------------------------------------------
# post-steps of align_frames_in_geometry that move lumen coordinates (align_within.rs:136-170)
# ---------------------------------------------------------------------------------------
def _median(values):
    v = sorted(values)
    n = len(v)
    if n == 0:
        return 0.0
    return v[n // 2] if n % 2 == 1 else (v[n // 2 - 1] + v[n // 2]) / 2.0


def _detect_holes(g: G.FlatGeometry):
    """align_within.rs:352-376"""
    z = g.centroids[:, 2]
    diffs = [abs(float(z[i]) - float(z[i - 1])) for i in range(1, g.n_frames)]
    if not diffs:
        return False, 0.0
    baseline = _median(diffs)
    if baseline <= 2.220446049250313e-16:
        return False, baseline
    return any(d >= 1.5 * baseline for d in diffs), baseline


def _dist3(p, q):
    dx, dy, dz = float(p[0]) - float(q[0]), float(p[1]) - float(q[1]), float(p[2]) - float(q[2])
    return math.sqrt(dx * dx + dy * dy + dz * dz)


def _find_farthest_points(pts: np.ndarray):
    """contour.rs:227-242 (first pair with the strictly largest 3-D distance)."""
    d = pts[:, None, :] - pts[None, :, :]
    d2 = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    n = pts.shape[0]
    iu = np.triu_indices(n, k=1)
    vals = d2[iu]
    if vals.size == 0 or vals.max() <= 0.0:
        return (0, 0), 0.0
    k = int(np.argmax(vals))            # first maximum in (i asc, j asc) order == the reference's strict `>`
    return (int(iu[0][k]), int(iu[1][k])), float(vals[k])


def _find_closest_opposite_3d(pts: np.ndarray):
    """contour.rs:313-333"""
    n = pts.shape[0]
    half = n // 2
    best, best_pair = float("inf"), (0, half)
    for i in range(n):
        j = (i + half) % n
        dd = _dist3(pts[i], pts[j])
        if dd < best:
            best, best_pair = dd, (i, j)
    return best_pair, best


def _elliptic_ratio(pts: np.ndarray) -> float:
    """contour.rs:335-343"""
    major = _find_farthest_points(pts)[1]
    minor = _find_closest_opposite_3d(pts)[1]
    return minor / major if major < minor else major / minor


def _angle_ref_point_to_right(g: G.FlatGeometry, ref_idx: int, anomalous: bool) -> float:
    """align_within.rs:256-318"""
    if not g.has_ref[ref_idx]:
        raise RuntimeError("No reference point found in frame")
    rp = g.ref[ref_idx]
    lum = g.frame_lumen(ref_idx)
    if anomalous:
        (i, j), _ = _find_farthest_points(lum)
        p1, p2 = tuple(float(v) for v in lum[i]), tuple(float(v) for v in lum[j])
    else:
        p1 = tuple(float(v) for v in g.centroids[ref_idx])
        p2 = (float(rp[0]), float(rp[1]), float(rp[2]))
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    line_angle = math.atan2(dy, dx)
    desired = math.pi / 2.0 if anomalous else 0.0
    two_pi = 2.0 * math.pi
    rotation = math.fmod(desired - line_angle, two_pi)
    if rotation < 0.0:
        rotation += two_pi

    def rotate2(pt, center, angle):
        ddx, ddy = pt[0] - center[0], pt[1] - center[1]
        s, c = sincos(angle)
        return (ddx * c - ddy * s + center[0], ddx * s + ddy * c + center[1])

    center = (p1[0], p1[1])
    ref2 = (float(rp[0]), float(rp[1]))
    rotated_ref = rotate2(ref2, center, rotation)
    all_good = True
    for op in ((p1[0], p1[1]), (p2[0], p2[1])):
        # approx::abs_diff_eq! default epsilon = f64::EPSILON
        if abs(op[0] - ref2[0]) <= 2.220446049250313e-16 and abs(op[1] - ref2[1]) <= 2.220446049250313e-16:
            continue
        if rotated_ref[0] <= rotate2(op, center, rotation)[0]:
            all_good = False
            break
    if not all_good:
        rotation = math.fmod(rotation + math.pi, two_pi)
        if rotation < 0.0:
            rotation += two_pi
    return rotation


def _rotate_geometry(g: G.FlatGeometry, angle: float):
    """Geometry::rotate_geometry (geometry.rs:241-250): every frame about its own centroid, then its
    contours re-sorted -- one call into the C ABI (mm_rotate_geometry, csrc/mm_centerline.cpp)."""
    if angle == 0.0:
        return
    from .centerline import rotate_geometry
    rotate_geometry(g, angle)


def _ref_or_proximal(g: G.FlatGeometry) -> int:
    """find_ref_frame_idx().unwrap_or(find_proximal_end_idx()) (geometry.rs:42-69)"""
    for i in range(g.n_frames):
        if g.has_ref is not None and g.has_ref[i]:
            return int(g.ids[i])
    n = g.n_frames
    if n == 0:
        return 0
    if n == 1:
        return int(g.lumen_ids[0])
    return int(g.lumen_ids[0] if g.orig_frames[0] > g.orig_frames[-1] else g.lumen_ids[-1])


def _replace(g: G.FlatGeometry, h: G.FlatGeometry) -> None:
    """g <- h in place (callers hold references to g)."""
    for name in ("ids", "lumen_ids", "orig_frames", "centroids", "lumen_off", "lumen", "cath_off", "cath",
                 "extra_off", "extra", "has_ref", "ref", "label", "meta", "has_lumen_centroid", "lumen_centroids"):
        setattr(g, name, getattr(h, name))


def _finish_within(g: G.FlatGeometry, ref_idx: int, smooth: bool) -> bool:
    """align_within.rs:136-160 after the chain: hole filling, reference point to the right, aortic flags,
    wall contours, smoothing; returns the anomalous flag.  The lumen contour centroid the reference
    carries at this point is the mean before the chain's last rotation (frame.rs:20); here it is the
    mean of the current points (x, y may differ by that rotation; z is exact)."""
    from .centerline import with_lumen_centroids
    if not os.environ.get("MM_PY_POSTPROC"):
        # the product path: behind the C ABI (mm_frames_finish_within, csrc/mm_frames.cpp); the Python below is the
        # same logic and stays as its checker (tests/test_native_frames.py)
        from . import native_frames as NF
        tracked = g.lumen_centroids is not None and (g.has_lumen_centroid is None or bool(np.all(g.has_lumen_centroid)))
        if not tracked:
            with_lumen_centroids(g)                  # no contour centroids came in: the mean of the points stands in
        h, anomalous = NF.finish_within(g, ref_idx, smooth)
        _replace(g, h)
        g.meta["anomalous"] = bool(anomalous)
        g.meta["lumen_centroid_fresh"] = bool(smooth)
        g.meta["lumen_centroid_tracked"] = bool(tracked)
        return bool(anomalous)
    return _checker("api_python").finish_within_python(g, ref_idx, smooth)     # MM_PY_POSTPROC=1 (tests only)


def align_frames_in_geometries(geoms: Sequence[G.FlatGeometry], step_deg: float, range_deg: float, smooth: bool,
                               bruteforce: bool, sample_size: int, engine: Optional[N.Engine] = None,
                               precision: int = N.MM_PRECISION_F32_BOUNDED, mode: int = 1):
    """``align_frames_in_geometry`` (align_within.rs:24-171) for several pullbacks at once (the
    reference's crossbeam scope, entry.rs:140-203).  In place; returns (logs, anomalous flags).
    The default precision resolves large candidate grids by bounds (small batches are
    screened outright) -- same winners, logs and coordinates as any other precision."""
    eng = engine or default_engine()
    for g in geoms:                                      # align_within.rs:32-40
        if g.n_frames == 0:
            raise RuntimeError("Geometry contains no frames")
        if int(g.lumen_off[1] - g.lumen_off[0]) == 0:
            raise RuntimeError("Lumen contours have no points")
    if sample_size == 0:
        raise RuntimeError("sample_size must be > 0")
    ref_idx = [_ref_or_proximal(g) for g in geoms]       # :42-44, before the chain
    logs, _ = G.align_within(eng, geoms, step_deg, range_deg, bruteforce, sample_size, precision=precision, mode=mode)
    _mark("  within: mm_align_within (sets, search, chain walk)")
    if len(geoms) > 1:
        # the reference finishes the pullbacks in parallel threads (entry.rs:140-203); the bulk numpy and
        # C-ABI calls of the post-steps release the GIL
        flags = list(_pool().map(lambda gr: _finish_within(gr[0], gr[1], smooth), zip(geoms, ref_idx)))
    else:
        flags = [_finish_within(g, r, smooth) for g, r in zip(geoms, ref_idx)]
    return logs, flags


# ---------------------------------------------------------------------------------------
# prepare_n_geometries (preprocessing.rs:27-201)
# ---------------------------------------------------------------------------------------
def _basename(path: str) -> str:
    b = os.path.basename(os.path.normpath(path))
    return b if b else "unknown"


def _prepare_from_paths(paths: Sequence[str], labels, n_expected, image_center, radius, n_points, single_diastole=None):
    use_labels = labels is not None and len(labels) == n_expected
    geoms, idx = [], 0
    for p in paths:
        phases = [single_diastole] if single_diastole is not None else [True, False]
        for dia in phases:
            label = labels[idx] if use_labels else _basename(p)
            geoms.append(build_geometry_from_inputdata(None, p, label, dia, image_center, radius, n_points))
            idx += 1
    return geoms


def _prepare_from_inputs(inputs: Sequence[InputData], image_center, radius, n_points):
    build = lambda d: build_geometry_from_inputdata(d, None, d.label, d.diastole, image_center, radius, n_points)
    if len(inputs) > 1:              # the native builder releases the interpreter lock: the pullbacks build in parallel
        return list(_pool().map(build, inputs))
    return [build(d) for d in inputs]


def _write_pairs(write_obj: bool, pairs, paths, interpolation_steps: int, watertight: bool, contour_types):
    """to_object::process_case per pair (entry.rs:291-349): OBJ + MTL + textures into the pair's directory."""
    if not write_obj:
        return
    from . import export as EX
    from . import frames as FR
    kinds = EX.DEFAULT_CONTOUR_TYPES if contour_types is None else contour_types
    for pr, path in zip(pairs, paths):
        try:
            EX.process_case(pr.label, FR.to_frames(pr.geom_a), FR.to_frames(pr.geom_b), path, interpolation_steps,
                            watertight, kinds)
        except RuntimeError as e:
            raise RuntimeError(f"process case failed for {pr.label}: {e}") from e


def _write_single(write_obj: bool, g: G.FlatGeometry, path: str, watertight: bool, contour_types):
    """entry.rs:740-776."""
    if write_obj:
        from . import export as EX
        EX.write_single_mode(g, path, watertight, EX.DEFAULT_CONTOUR_TYPES if contour_types is None else contour_types)


def _maybe_postprocess(pair: GeometryPair, anomalous: bool, postprocessing: bool) -> GeometryPair:
    """maybe_postprocess (entry.rs:57-69): postprocess_geom_pair(pair, TOLERANCE, anomalous)."""
    if not postprocessing:
        return pair
    if not os.environ.get("MM_PY_POSTPROC"):
        from . import native_frames as NF             # mm_frames_postprocess_pair (csrc/mm_frames.cpp)
        try:
            a, b = NF.postprocess_pair(pair.geom_a, pair.geom_b, TOLERANCE, anomalous)
        except RuntimeError as e:
            raise RuntimeError(f"Failed postprocessing of {pair.label}: {e}") from e
        return GeometryPair(a, b, pair.label)
    return _checker("api_python").maybe_postprocess_python(pair, anomalous)     # MM_PY_POSTPROC=1 (tests only)


# ---------------------------------------------------------------------------------------
# the four modes (entry.rs:71, 363, 572, 691)
# ---------------------------------------------------------------------------------------
_TRACE = [] if os.environ.get("MM_API_TRACE") else None      # [(label, perf_counter)]: tools/bench_api.py --stages


def _mark(label):
    if _TRACE is not None:
        import time
        _TRACE.append((label, time.perf_counter()))


def _full(geoms, step, rng, smooth, bruteforce, sample_size, engine, both_batches=True, postprocessing=False):
    eng = engine or default_engine()
    _mark("geometries built")
    logs, flags = align_frames_in_geometries(geoms, step, rng, smooth, bruteforce, sample_size, eng)
    _mark("within: search + chain + post-steps")
    anomalous = any(flags)                                                         # entry.rs:279-280
    post = lambda pr: _maybe_postprocess(pr, anomalous, postprocessing)
    a, b, c, d = geoms
    G.align_between(eng, [(a, b), (c, d)], rng, step, sample_size)                 # entry.rs:206-240
    _mark("between AB | CD")
    b.meta["lumen_centroid_fresh"] = d.meta["lumen_centroid_fresh"] = True         # moved last by a translation
    # The reference clones the geometries into every pair.  With the native post-processing each pair's geometries are
    # rebuilt from (and never written through) the ones handed in, so a, b -- final after the first batch -- and the
    # final c, d can be handed in as they are; only the state of c and d BEFORE the second batch needs a snapshot.
    share = postprocessing and not os.environ.get("MM_PY_POSTPROC")
    keep = (lambda g: g) if share else (lambda g: g.copy())
    pool = _pool()
    if not both_batches:
        pair_ab, pair_cd = _make_pair(keep(a), keep(b)), _make_pair(keep(c), keep(d))
        out = [post(pair_ab), post(pair_cd)] if not postprocessing else list(pool.map(post, [pair_ab, pair_cd]))
        return (*out, tuple(logs))
    # The pairs are independent (maybe_postprocess x4, entry.rs:282-290) and the native post-processing releases the
    # interpreter lock: AB and CD are post-processed on pool threads WHILE the second batch of between alignments runs
    # (a and b are final after the first batch; c and d move again, so pair CD works on snapshots taken -- in parallel
    # -- before the second batch starts).
    pair_ab = _make_pair(keep(a), keep(b))
    if share:
        # the native post-processing starts by copying the pair into the library's frame lists: done HERE for C and D, that
        # copy is the snapshot of their state before the second batch moves them again
        from . import native_frames as NF

        def post_staged(staged, label):
            try:
                return GeometryPair(*NF.postprocess_staged(*staged, TOLERANCE, anomalous), label)
            except RuntimeError as e:
                raise RuntimeError(f"Failed postprocessing of {label}: {e}") from e

        import copy
        views = [copy.copy(g) for g in (c, d)]          # shallow: the arrays are shared, attribute updates are not
        for v, g in zip(views, (c, d)):
            v.meta = dict(g.meta)
        view_cd = _make_pair(*views)
        staged_cd = NF.stage_pair(view_cd.geom_a, view_cd.geom_b)
        early = [pool.submit(post, pair_ab), pool.submit(post_staged, staged_cd, view_cd.label)]
    else:
        snap_c, snap_d = pool.map(lambda g: g.copy(), (c, d))
        pair_cd = _make_pair(snap_c, snap_d)
        early = [pool.submit(post, pr) for pr in (pair_ab, pair_cd)] if postprocessing else None
    _mark("snapshot of C, D; post-processing of AB, CD started")
    G.align_between(eng, [(a, c), (b, d)], rng, step, sample_size)                 # entry.rs:243-277
    _mark("between AC | BD")
    c.meta["lumen_centroid_fresh"] = d.meta["lumen_centroid_fresh"] = True
    pair_ac = _make_pair(keep(a), keep(c))
    pair_bd = _make_pair(keep(b), keep(d))
    if postprocessing:
        late = [pool.submit(post, pr) for pr in (pair_ac, pair_bd)]
        out = [f.result() for f in early + late]
    else:
        out = [pair_ab, pair_cd, pair_ac, pair_bd]
    _mark("postprocess x 4")
    return (*out, tuple(logs))


_POOL = None


def _pool():
    """One small thread pool for the per-pullback / per-pair host work of the entry points (the reference uses crossbeam
    scopes of 4 and 2 threads, entry.rs:140-290); created once -- starting and joining one per call costs a millisecond."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="mm-api")
    return _POOL


def from_array_full(input_data_a: InputData, input_data_b: InputData, input_data_c: InputData, input_data_d: InputData,
                    step_rotation_deg: float = 0.5, range_rotation_deg: float = 90.0, sample_size: int = 500,
                    image_center=(4.5, 4.5), radius: float = 0.5, n_points: int = 20, write_obj: bool = True,
                    watertight: bool = True, contour_types=None, output_path_ab: str = "output/rest",
                    output_path_cd: str = "output/stress", output_path_ac: str = "output/diastole",
                    output_path_bd: str = "output/systole", interpolation_steps: int = 0, bruteforce: bool = False,
                    smooth: bool = True, postprocessing: bool = True, engine: Optional[N.Engine] = None):
    """_processing.py:553 / functions.rs:827 / entry.rs:71 -> (pair_ab, pair_cd, pair_ac, pair_bd, (logs x4))."""
    geoms = _prepare_from_inputs([input_data_a, input_data_b, input_data_c, input_data_d], image_center, radius, n_points)
    out = _full(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size, engine,
                postprocessing=postprocessing)
    _write_pairs(write_obj, out[:4], (output_path_ab, output_path_cd, output_path_ac, output_path_bd), interpolation_steps,
                 watertight, contour_types)
    return out


def from_file_full(input_path_ab: str, input_path_cd: str, labels=None, step_rotation_deg: float = 0.5,
                   range_rotation_deg: float = 90.0, sample_size: int = 500, image_center=(4.5, 4.5),
                   radius: float = 0.5, n_points: int = 20, write_obj: bool = True, watertight: bool = True,
                   contour_types=None, output_path_ab: str = "output/rest", output_path_cd: str = "output/stress",
                   output_path_ac: str = "output/diastole", output_path_bd: str = "output/systole",
                   interpolation_steps: int = 0, bruteforce: bool = False, smooth: bool = True,
                   postprocessing: bool = True, engine: Optional[N.Engine] = None):
    """_processing.py:42 / functions.rs:168 / entry.rs:71."""
    geoms = _prepare_from_paths([input_path_ab, input_path_cd], labels, 4, image_center, radius, n_points)
    out = _full(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size, engine,
                postprocessing=postprocessing)
    _write_pairs(write_obj, out[:4], (output_path_ab, output_path_cd, output_path_ac, output_path_bd), interpolation_steps,
                 watertight, contour_types)
    return out


def from_array_doublepair(input_data_a: InputData, input_data_b: InputData, input_data_c: InputData,
                          input_data_d: InputData, step_rotation_deg: float = 0.5, range_rotation_deg: float = 90.0,
                          sample_size: int = 500, image_center=(4.5, 4.5), radius: float = 0.5, n_points: int = 20,
                          write_obj: bool = True, watertight: bool = True, contour_types=None,
                          output_path_ab: str = "output/rest", output_path_cd: str = "output/stress",
                          interpolation_steps: int = 0, bruteforce: bool = False, smooth: bool = True,
                          postprocessing: bool = True, engine: Optional[N.Engine] = None):
    """_processing.py:698 / entry.rs:363 -> (pair_ab, pair_cd, (logs x4))."""
    geoms = _prepare_from_inputs([input_data_a, input_data_b, input_data_c, input_data_d], image_center, radius, n_points)
    out = _full(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size, engine, both_batches=False,
                postprocessing=postprocessing)
    _write_pairs(write_obj, out[:2], (output_path_ab, output_path_cd), interpolation_steps, watertight, contour_types)
    return out


def from_file_doublepair(input_path_ab: str, input_path_cd: str, labels=None, step_rotation_deg: float = 0.5,
                         range_rotation_deg: float = 90.0, sample_size: int = 500, image_center=(4.5, 4.5),
                         radius: float = 0.5, n_points: int = 20, write_obj: bool = True, watertight: bool = True,
                         contour_types=None, output_path_ab: str = "output/rest", output_path_cd: str = "output/stress",
                         interpolation_steps: int = 0, bruteforce: bool = False, smooth: bool = True,
                         postprocessing: bool = True, engine: Optional[N.Engine] = None):
    """_processing.py:201 / entry.rs:363."""
    geoms = _prepare_from_paths([input_path_ab, input_path_cd], labels, 4, image_center, radius, n_points)
    out = _full(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size, engine, both_batches=False,
                postprocessing=postprocessing)
    _write_pairs(write_obj, out[:2], (output_path_ab, output_path_cd), interpolation_steps, watertight, contour_types)
    return out


def _pair(geoms, step, rng, smooth, bruteforce, sample_size, engine, postprocessing=False):
    eng = engine or default_engine()
    logs, flags = align_frames_in_geometries(geoms, step, rng, smooth, bruteforce, sample_size, eng)
    a, b = geoms
    G.align_between(eng, [(a, b)], rng, step, sample_size)                          # entry.rs:655
    b.meta["lumen_centroid_fresh"] = True
    return _maybe_postprocess(_make_pair(a, b), any(flags), postprocessing), (logs[0], logs[1])


def from_array_singlepair(input_data_a: InputData, input_data_b: InputData, step_rotation_deg: float = 0.5,
                          range_rotation_deg: float = 90.0, sample_size: int = 500, image_center=(4.5, 4.5),
                          radius: float = 0.5, n_points: int = 20, write_obj: bool = True, watertight: bool = True,
                          contour_types=None, output_path: str = "output/singlepair", interpolation_steps: int = 0,
                          bruteforce: bool = False, smooth: bool = True, postprocessing: bool = True,
                          engine: Optional[N.Engine] = None):
    """_processing.py:822 / entry.rs:572 -> (pair, (logs_a, logs_b))."""
    geoms = _prepare_from_inputs([input_data_a, input_data_b], image_center, radius, n_points)
    out = _pair(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size, engine, postprocessing)
    _write_pairs(write_obj, out[:1], (output_path,), interpolation_steps, watertight, contour_types)
    return out


def from_file_singlepair(input_path: str, labels=None, step_rotation_deg: float = 0.5, range_rotation_deg: float = 90.0,
                         sample_size: int = 500, image_center=(4.5, 4.5), radius: float = 0.5, n_points: int = 20,
                         write_obj: bool = True, watertight: bool = True, contour_types=None,
                         output_path: str = "output/singlepair", interpolation_steps: int = 0, bruteforce: bool = False,
                         smooth: bool = True, postprocessing: bool = True, engine: Optional[N.Engine] = None):
    """_processing.py:333 / functions.rs:517 / entry.rs:572: one folder read twice (diastole, systole)."""
    geoms = _prepare_from_paths([input_path], labels, 2, image_center, radius, n_points)
    out = _pair(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size, engine, postprocessing)
    _write_pairs(write_obj, out[:1], (output_path,), interpolation_steps, watertight, contour_types)
    return out


def from_array_single(input_data: InputData, step_rotation_deg: float = 0.5, range_rotation_deg: float = 90.0,
                      sample_size: int = 500, image_center=(4.5, 4.5), radius: float = 0.5, n_points: int = 20,
                      write_obj: bool = True, watertight: bool = True, contour_types=None,
                      output_path: str = "output/single", bruteforce: bool = False, smooth: bool = True,
                      engine: Optional[N.Engine] = None):
    """_processing.py:922 / functions.rs:1350 / entry.rs:691 -> (geometry, logs)."""
    geoms = _prepare_from_inputs([input_data], image_center, radius, n_points)
    logs, _ = align_frames_in_geometries(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size,
                                         engine)
    g = _with_contour_centroids(geoms[0])
    _write_single(write_obj, g, output_path, watertight, contour_types)
    return g, logs[0]


def from_file_single(input_path: str, labels=None, diastole: bool = True, step_rotation_deg: float = 0.5,
                     range_rotation_deg: float = 90.0, sample_size: int = 500, image_center=(4.5, 4.5),
                     radius: float = 0.5, n_points: int = 20, write_obj: bool = True, watertight: bool = True,
                     contour_types=None, output_path: str = "output/single", bruteforce: bool = False,
                     smooth: bool = True, engine: Optional[N.Engine] = None):
    """_processing.py:449 / functions.rs:656 / entry.rs:691."""
    geoms = _prepare_from_paths([input_path], labels, 1, image_center, radius, n_points, single_diastole=diastole)
    logs, _ = align_frames_in_geometries(geoms, step_rotation_deg, range_rotation_deg, smooth, bruteforce, sample_size,
                                         engine)
    g = _with_contour_centroids(geoms[0])
    _write_single(write_obj, g, output_path, watertight, contour_types)
    return g, logs[0]
