"""ctypes front-end of ``mm_oracle_ccta.c`` -- TEST INFRASTRUCTURE ONLY (see mm_oracle_ccta.h).
CCTA diameter search of the reference (src/ccta/adjust_mesh/scale_coronary.rs) restated on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import oracle as O
from .oracle_cl import _cl, _v3

_ready = False


def lib():
    global _ready
    L = O.lib()
    if _ready:
        return L
    P, D, Z = C.c_void_p, C.c_double, C.c_size_t
    L.orc_nn_min_sq.restype = None
    L.orc_nn_min_sq.argtypes = [P, Z, P, Z, P]
    L.orc_symmetric_nn_distance.restype = D
    L.orc_symmetric_nn_distance.argtypes = [P, Z, P, Z]
    L.orc_diameter_morphing.restype = None
    L.orc_diameter_morphing.argtypes = [P, Z, P, Z, D, P]
    L.orc_find_region_points.restype = Z
    L.orc_find_region_points.argtypes = [P, Z, P, Z, Z, P, P]
    L.orc_aortic_diameter_optimization.restype = D
    L.orc_aortic_diameter_optimization.argtypes = [P, Z, P, Z, P, Z, P]
    L.orc_diameter_optimization.restype = None
    L.orc_diameter_optimization.argtypes = [P, Z, Z, Z, P, Z, P, Z, P, Z, C.POINTER(D), C.POINTER(D)]
    L.orc_clean_outlier_points.restype = None
    L.orc_clean_outlier_points.argtypes = [P, Z, P, Z, D, D, P]
    L.orc_find_points_by_cl_region.restype = None
    L.orc_find_points_by_cl_region.argtypes = [P, P, Z, P, Z, P, Z, P]
    L.orc_wall_diameter_optimization.restype = D
    L.orc_wall_diameter_optimization.argtypes = [P, Z, P, P, Z]
    _ready = True
    return L


def _p3(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 3))


def nn_min_sq(a, b) -> np.ndarray:
    a, b = _p3(a), _p3(b)
    out = np.zeros(a.shape[0])
    lib().orc_nn_min_sq(O._p(a), a.shape[0], O._p(b), b.shape[0], O._p(out))
    return out


def symmetric_nn_distance(a, b) -> float:
    a, b = _p3(a), _p3(b)
    return lib().orc_symmetric_nn_distance(O._p(a), a.shape[0], O._p(b), b.shape[0])


def diameter_morphing(cl, points, adjustment_mm) -> np.ndarray:
    cl, p = _cl(cl), _p3(points)
    out = np.zeros_like(p)
    lib().orc_diameter_morphing(O._p(cl), cl.shape[0], O._p(p), p.shape[0], float(adjustment_mm), O._p(out))
    return out


def find_region_points(anomalous, reference, n_points):
    a, r = _p3(anomalous), _p3(reference)
    sel, rem = np.zeros_like(a), np.zeros_like(a)
    k = lib().orc_find_region_points(O._p(a), a.shape[0], O._p(r), r.shape[0], int(n_points), O._p(sel), O._p(rem))
    return sel[:k].copy(), rem[: a.shape[0] - k].copy()


def aortic_diameter_optimization(intramural, reference, cl):
    """Returns (best scaling, the 41 distances)."""
    i, r, cl = _p3(intramural), _p3(reference), _cl(cl)
    d = np.zeros(41)
    best = lib().orc_aortic_diameter_optimization(O._p(i), i.shape[0], O._p(r), r.shape[0], O._p(cl), cl.shape[0], O._p(d))
    return best, d


def diameter_optimization(anomalous, n_proximal, n_distal, cl, proximal_reference, distal_reference):
    a, pr, dr, cl = _p3(anomalous), _p3(proximal_reference), _p3(distal_reference), _cl(cl)
    pb, db = C.c_double(0), C.c_double(0)
    lib().orc_diameter_optimization(O._p(a), a.shape[0], int(n_proximal), int(n_distal), O._p(cl), cl.shape[0],
                                    O._p(pr), pr.shape[0], O._p(dr), dr.shape[0], C.byref(pb), C.byref(db))
    return pb.value, db.value


def wall_diameter_optimization(cl, ref_pt, aortic) -> float:
    cl, r, a = _cl(cl), _v3(ref_pt), _p3(aortic)
    return lib().orc_wall_diameter_optimization(O._p(cl), cl.shape[0], O._p(r), O._p(a), a.shape[0])


def clean_outlier_points(points_to_cleanup, reference_points, neighborhood_radius, min_neigbor_ratio):
    """clean_up_non_section_points (scale_coronary.rs:342-409) -> (cleaned points, augmented reference)."""
    c, r = _p3(points_to_cleanup), _p3(reference_points)
    mv = np.zeros(c.shape[0], dtype=np.uint8)
    lib().orc_clean_outlier_points(O._p(c), c.shape[0], O._p(r), r.shape[0], float(neighborhood_radius),
                                   float(min_neigbor_ratio), O._p(mv))
    return c[mv == 0].copy(), np.concatenate([r, c[mv == 1]], axis=0)


def find_points_by_cl_region(cl, frame_centroids, points, cl_frame_index=None):
    """find_points_by_cl_region_rs (scale_coronary.rs:263-312) -> (proximal, distal, between, labels)."""
    cl, p = _cl(cl), _p3(points)
    cen = np.ascontiguousarray(np.asarray(frame_centroids, dtype=np.float64).reshape(-1, 3))
    fi = None if cl_frame_index is None else np.ascontiguousarray(cl_frame_index, dtype=np.uint32)
    lab = np.zeros(p.shape[0], dtype=np.uint8)
    lib().orc_find_points_by_cl_region(O._p(cl), None if fi is None else O._p(fi), cl.shape[0], O._p(cen), cen.shape[0],
                                       O._p(p), p.shape[0], O._p(lab))
    between = np.concatenate([p[lab == 2], p[lab == 3], p[lab == 4]], axis=0)
    return p[lab == 0].copy(), p[lab == 1].copy(), between, lab
