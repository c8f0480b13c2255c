"""CCTA diameter search, mirroring the reference's entry points
``adjust_diameter_centerline_morphing_simple`` / ``find_proximal_distal_scaling`` /
``find_aortic_scaling`` / ``find_aortic_wall_scaling`` (src/ccta/binding/ccta_py.rs:263-481;
implementation src/ccta/adjust_mesh/scale_coronary.rs:8-261) and the wrapper
``find_distal_and_proximal_scaling`` of multimodars/ccta/scaling.py:84-145.

Same argument names, order and meaning; points are ``(N, 3)`` arrays (or lists of tuples).  Each
search scores its 41 scalings in one GPU batch (exact f64 nearest-neighbour minima,
csrc/mm_nn_kernels.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from . import geometry as G
from .centerline import Centerline

SCALING_STEPS = 41


def _p3(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 3))


def _engine(engine: Optional[N.Engine]) -> N.Engine:
    if engine is not None:
        return engine
    from .api import default_engine
    return default_engine()


def nn_min_sq(a, b, engine: Optional[N.Engine] = None) -> np.ndarray:
    """min_b |a_i - b|^2 for every a_i (the inner fold of symmetric_nn_distance / find_region_points)."""
    a, b = _p3(a), _p3(b)
    xyz = np.ascontiguousarray(np.concatenate([a, b], axis=0))
    set_off = np.array([0, a.shape[0], a.shape[0] + b.shape[0]], dtype=np.int64)
    q, p = np.array([0], dtype=np.int32), np.array([1], dtype=np.int32)
    out_off = np.array([0, a.shape[0]], dtype=np.int64)
    out = np.zeros(a.shape[0], dtype=np.float64)
    N.check(N.lib().mm_nn_min_sq_batch(_engine(engine).handle, 2, N._ptr(set_off), N._ptr(xyz), 1, N._ptr(q),
                                       N._ptr(p), N._ptr(out_off), N._ptr(out)), "nn_min_sq")
    return out


def symmetric_nn_distance(a, b, engine: Optional[N.Engine] = None) -> float:
    """scale_coronary.rs:188-216 (RMS of the two mean squared nearest-neighbour distances)."""
    a, b = _p3(a), _p3(b)
    out = C.c_double(0.0)
    N.check(N.lib().mm_symmetric_nn_distance(_engine(engine).handle, N._ptr(a), a.shape[0], N._ptr(b), b.shape[0],
                                             C.byref(out)), "symmetric_nn_distance")
    return out.value


def adjust_diameter_centerline_morphing_simple(centerline: Centerline, points, diameter_adjustment_mm: float) -> np.ndarray:
    """ccta_py.rs:263-278: every point moves by ``diameter_adjustment_mm`` along the direction from
    its closest centerline point."""
    p = _p3(points)
    out = np.zeros_like(p)
    N.check(N.lib().mm_diameter_morphing(N._ptr(centerline.points), len(centerline), N._ptr(p), p.shape[0],
                                         float(diameter_adjustment_mm), N._ptr(out)), "diameter_morphing")
    return out


def find_region_points(anomalous_points, reference_points, n_points: int, engine: Optional[N.Engine] = None):
    """scale_coronary.rs:133-183 -> (selected, remaining)."""
    a, r = _p3(anomalous_points), _p3(reference_points)
    sel, rem = np.zeros_like(a), np.zeros_like(a)
    k = N.lib().mm_find_region_points(_engine(engine).handle, N._ptr(a), a.shape[0], N._ptr(r), r.shape[0],
                                      int(n_points), N._ptr(sel), N._ptr(rem))
    if k < 0:
        N.check(int(k), "find_region_points")
    return sel[:k].copy(), rem[: a.shape[0] - k].copy()


def find_proximal_distal_scaling(anomalous_points, n_proximal: int, n_distal: int, centerline: Centerline,
                                 proximal_reference, distal_reference,
                                 engine: Optional[N.Engine] = None) -> Tuple[float, float]:
    """ccta_py.rs:389-407 / multimodars/_processing.py:1431-1473."""
    a, pr, dr = _p3(anomalous_points), _p3(proximal_reference), _p3(distal_reference)
    pb, db = C.c_double(0.0), C.c_double(0.0)
    N.check(N.lib().mm_diameter_optimization(_engine(engine).handle, N._ptr(a), a.shape[0], int(n_proximal),
                                             int(n_distal), N._ptr(centerline.points), len(centerline), N._ptr(pr),
                                             pr.shape[0], N._ptr(dr), dr.shape[0], C.byref(pb), C.byref(db)),
            "find_proximal_distal_scaling")
    return pb.value, db.value


def find_aortic_scaling(intramural_points, reference_points, centerline: Centerline,
                        engine: Optional[N.Engine] = None, return_distances: bool = False):
    """ccta_py.rs:428-442."""
    i, r = _p3(intramural_points), _p3(reference_points)
    best = C.c_double(0.0)
    d = np.zeros(SCALING_STEPS, dtype=np.float64)
    N.check(N.lib().mm_aortic_diameter_optimization(_engine(engine).handle, N._ptr(i), i.shape[0], N._ptr(r),
                                                    r.shape[0], N._ptr(centerline.points), len(centerline),
                                                    C.byref(best), N._ptr(d)), "find_aortic_scaling")
    return (best.value, d) if return_distances else best.value


def find_aortic_wall_scaling_raw(cl_aorta: Centerline, ref_pt_coronary, aortic_pts) -> float:
    """The binding ``find_aortic_wall_scaling`` (ccta_py.rs:467-481, host only)."""
    r = np.ascontiguousarray(np.asarray(ref_pt_coronary, dtype=np.float64).reshape(3))
    a = _p3(aortic_pts)
    out = C.c_double(0.0)
    N.check(N.lib().mm_wall_diameter_optimization(N._ptr(cl_aorta.points), len(cl_aorta), N._ptr(r), N._ptr(a),
                                                  a.shape[0], C.byref(out)), "find_aortic_wall_scaling")
    return out.value


def find_distal_and_proximal_scaling(geometry: G.FlatGeometry, centerline: Centerline, results: dict,
                                     dist_range: int = 3, prox_range: int = 2,
                                     engine: Optional[N.Engine] = None) -> Tuple[float, float]:
    """multimodars/ccta/scaling.py:84-145 with the frame list given as a FlatGeometry: the lumen
    points of the last ``dist_range`` / first ``prox_range`` frames are the references and a quarter
    of the anomalous points is compared on either side."""
    F = geometry.n_frames
    dist_pts = geometry.lumen[geometry.lumen_off[max(F - dist_range, 0)]:geometry.lumen_off[F]]
    prox_pts = geometry.lumen[geometry.lumen_off[0]:geometry.lumen_off[min(prox_range, F)]]
    n_section = int(math.ceil(0.25 * len(results["anomalous_points"])))
    return find_proximal_distal_scaling(results["anomalous_points"], n_section, n_section, centerline, prox_pts,
                                        dist_pts, engine=engine)


def _extract_wall_from_frames(geometry: G.FlatGeometry):
    """multimodars/ccta/scaling.py:239-297: the straight (coronary-side) half of the Wall contour, point
    indices below n/2, of the LAST frame that carries an aortic thickness; None if there is none."""
    from . import frames as FR
    fr = FR.to_frames(geometry)
    half = len(fr[0].lumen) // 2
    ref = None
    for f in fr:
        if f.lumen.aortic_thickness is None:
            continue
        wall = f.extras.get("wall")
        if wall is None:
            raise ValueError(f"No Wall extras found for frame {f.id}")
        if len(wall) == 0:
            raise ValueError(f"Empty Wall extras for frame {f.id}")
        ref = wall.points[:half].copy()
    return ref


def find_aorta_scaling(geometry: G.FlatGeometry, cl_aorta: Centerline, results: dict,
                       engine: Optional[N.Engine] = None) -> float:
    """multimodars/ccta/scaling.py:148-189: the removed RCA points are scaled radially about the aortic
    centerline against the straight wall of the intravascular frames."""
    ref = _extract_wall_from_frames(geometry)
    if ref is None:
        raise ValueError("No aortic wall points found in frames for scaling reference")
    return find_aortic_scaling(results["rca_removed_points"], ref, cl_aorta, engine=engine)


def find_aortic_wall_scaling(geometry_or_centerline, cl_aorta=None, results=None, aortic_pts=None):
    """Both spellings of the reference: the wrapper ``find_aortic_wall_scaling(frames, cl_aorta, results)``
    (multimodars/ccta/scaling.py:192-236: reference point = the point at index n/4 of the first lumen with
    an elliptic ratio < 1.3, scored against ``results["aorta_points"]``) when the first argument is a
    geometry, and the binding ``find_aortic_wall_scaling(cl_aorta, ref_pt_coronary, aortic_pts)``
    (ccta_py.rs:467-481) when it is a centerline."""
    if isinstance(geometry_or_centerline, Centerline):
        pts = aortic_pts if aortic_pts is not None else results
        return find_aortic_wall_scaling_raw(geometry_or_centerline, cl_aorta, pts)
    from .api import _elliptic_ratio
    g = geometry_or_centerline
    ref_point = None
    for i in range(g.n_frames):
        lum = g.frame_lumen(i)
        if _elliptic_ratio(lum) < 1.3:
            ref_point = lum[lum.shape[0] // 4].copy()
            break
    if ref_point is None:
        raise ValueError("No coronary reference point found")
    return find_aortic_wall_scaling_raw(cl_aorta, ref_point, results["aorta_points"])


def clean_outlier_points(points_to_cleanup, reference_points, neighborhood_radius: float, min_neigbor_ratio: float,
                         engine: Optional[N.Engine] = None):
    """ccta_py.rs:345-358 -> ``clean_up_non_section_points`` (scale_coronary.rs:342-409): a point of
    ``points_to_cleanup`` most of whose neighbours (within ``neighborhood_radius``) are reference points joins
    the reference set.  Returns (cleaned points, reference points + the moved ones in input order).  The
    neighbour counts run on the device in exact f64."""
    c, r = _p3(points_to_cleanup), _p3(reference_points)
    mv = np.zeros(c.shape[0], dtype=np.uint8)
    N.check(N.lib().mm_clean_outlier_points(_engine(engine).handle, N._ptr(c), c.shape[0], N._ptr(r), r.shape[0],
                                            float(neighborhood_radius), float(min_neigbor_ratio), N._ptr(mv)),
            "clean_outlier_points")
    return c[mv == 0].copy(), np.concatenate([r, c[mv == 1]], axis=0)


def find_points_by_cl_region(centerline: Centerline, frames, points, engine: Optional[N.Engine] = None,
                             cl_frame_index=None, return_labels: bool = False):
    """ccta_py.rs:304-319 -> ``find_points_by_cl_region_rs`` (scale_coronary.rs:263-312): the points whose
    closest centerline point lies within the imaged section (within the mean frame spacing of a frame
    centroid) are ``between``; the rest is proximal or distal of the last frame's centroid; two density
    clean-ups then move stray proximal / distal points into ``between``.  ``frames``: a FlatGeometry (its
    frame centroids are used) or an (F, 3) array of centroids.  Returns (proximal, distal, between) as
    (n, 3) arrays in the reference's order."""
    cen = frames.centroids if isinstance(frames, G.FlatGeometry) else np.asarray(frames, dtype=np.float64)
    cen = np.ascontiguousarray(cen.reshape(-1, 3))
    p = _p3(points)
    fi = None if cl_frame_index is None else np.ascontiguousarray(cl_frame_index, dtype=np.uint32)
    if fi is not None and fi.shape[0] != len(centerline):
        raise ValueError("cl_frame_index: one entry per centerline point")
    lab = np.zeros(p.shape[0], dtype=np.uint8)
    N.check(N.lib().mm_find_points_by_cl_region(_engine(engine).handle, N._ptr(centerline.points), N._ptr(fi),
                                                len(centerline), N._ptr(cen), cen.shape[0], N._ptr(p), p.shape[0],
                                                N._ptr(lab)), "find_points_by_cl_region")
    out = (p[lab == 0].copy(), p[lab == 1].copy(), np.concatenate([p[lab == 2], p[lab == 3], p[lab == 4]], axis=0))
    return out + (lab,) if return_labels else out
