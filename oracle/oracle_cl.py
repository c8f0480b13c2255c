"""ctypes front-end of ``mm_oracle_cl.c`` -- TEST INFRASTRUCTURE ONLY (see mm_oracle_cl.h).

Centerline placement, three-point search and Hausdorff refinement of the reference
(src/intravascular/centerline_align/*.rs) restated on the CPU.  Only tests/,
``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import oracle as O

# orc_clpoint (64 bytes)
CL_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("tx", "<f8"), ("ty", "<f8"), ("tz", "<f8"),
                     ("radius", "<f8"), ("branch_id", "<u4"), ("pad_", "<u4")])


class _ClGeom(C.Structure):
    _fields_ = [("g", C.POINTER(O._Geometry)), ("has_lumen_centroid", C.c_void_p),
                ("lumen_centroid", C.c_void_p), ("n_extra_kinds", C.c_int32), ("extra_kind_off", C.c_void_p),
                ("lumen_aortic", C.c_void_p), ("wall_aortic", C.c_void_p), ("wall_kind1", C.c_int32)]


class _FrameTf(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("r", C.c_double * 9), ("pivot", C.c_double * 3)]


_ready = False


def lib():
    global _ready
    L = O.lib()
    if _ready:
        return L
    P, D, Z = C.c_void_p, C.c_double, C.c_size_t
    L.orc_centerline_from_points.restype = C.c_int
    L.orc_centerline_from_points.argtypes = [P, Z, P]
    L.orc_cl_find_ref_idx.restype = Z
    L.orc_cl_find_ref_idx.argtypes = [P, Z, P]
    L.orc_preprocess_centerline.restype = C.c_int64
    L.orc_preprocess_centerline.argtypes = [P, Z, C.POINTER(O._Geometry), P, Z, C.POINTER(D)]
    L.orc_sort_contour_points.restype = None
    L.orc_sort_contour_points.argtypes = [P, Z]
    L.orc_rotate_geometry.restype = None
    L.orc_rotate_geometry.argtypes = [C.POINTER(_ClGeom), D]
    L.orc_newell_normal.restype = None
    L.orc_newell_normal.argtypes = [P, Z, P, P]
    L.orc_align_frame.restype = None
    L.orc_align_frame.argtypes = [P, Z, C.c_int, P, P, C.POINTER(_FrameTf)]
    L.orc_tf_apply.restype = O._Point
    L.orc_tf_apply.argtypes = [C.POINTER(_FrameTf), O._Point]
    L.orc_apply_transformations.restype = Z
    L.orc_apply_transformations.argtypes = [P, C.c_int, P, Z, P]
    L.orc_rotation_from_axis_angle.restype = None
    L.orc_rotation_from_axis_angle.argtypes = [P, D, P]
    L.orc_rotate_contour_around_centroid.restype = None
    L.orc_rotate_contour_around_centroid.argtypes = [P, Z, C.c_int, P, D]
    L.orc_best_rotation_three_point.restype = D
    L.orc_best_rotation_three_point.argtypes = [P, Z, C.c_int, P, C.c_uint32, P, P, P, D, P]
    L.orc_refine_alignment_hausdorff.restype = C.c_int
    L.orc_refine_alignment_hausdorff.argtypes = [P, C.c_int, P, Z, Z, D, P, Z, D, D, Z, C.POINTER(D),
                                                 C.POINTER(Z), C.POINTER(D), P, Z, C.POINTER(Z)]
    L.orc_align_walls.restype = None
    L.orc_align_walls.argtypes = [P, C.c_int, C.c_int]
    L.orc_align_three_point.restype = C.c_int
    L.orc_align_three_point.argtypes = [P, Z, P, C.c_int, C.c_uint32, P, P, P, D, C.c_int, C.POINTER(D), C.POINTER(D)]
    L.orc_align_manual.restype = C.c_int
    L.orc_align_manual.argtypes = [P, Z, P, C.c_int, D, P, C.c_int, C.POINTER(D), C.POINTER(D)]
    L.orc_align_combined.restype = C.c_int
    L.orc_align_combined.argtypes = [P, Z, P, C.c_int, C.c_uint32, P, P, P, P, Z, D, D, Z, C.c_int,
                                     C.POINTER(D), C.POINTER(D), C.POINTER(Z)]
    _ready = True
    return L


def _v3(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(3))


def _cl(cl) -> np.ndarray:
    cl = np.ascontiguousarray(cl)
    assert cl.dtype == CL_DTYPE
    return cl


class _GeomPack:
    """Keeps the C structs of a list of OracleGeometry alive for one call."""

    def __init__(self, geoms: Sequence[O.OracleGeometry]):
        self.gs = [g._c() for g in geoms]
        self.cls = []
        for g, cg in zip(geoms, self.gs):
            c = _ClGeom()
            c.g = C.pointer(cg)
            c.has_lumen_centroid = O._p(g.has_lumen_centroid)
            c.lumen_centroid = O._p(g.lumen_centroids)
            c.n_extra_kinds = int(g.n_extra_kinds)
            c.extra_kind_off = O._p(g.extra_kind_off)
            for name in ("lumen_aortic", "wall_aortic"):
                a = getattr(g, name)
                if a is not None:
                    assert a.dtype == np.uint8 and a.flags.c_contiguous, name
                    setattr(c, name, O._p(a))
            c.wall_kind1 = int(g.wall_kind1)
            self.cls.append(c)
        self.arr = (C.POINTER(_ClGeom) * len(geoms))(*[C.pointer(c) for c in self.cls])

    @property
    def ptr(self):
        return C.cast(self.arr, C.c_void_p)


def with_lumen_centroids(g: O.OracleGeometry) -> O.OracleGeometry:
    """Sets Frame.lumen.centroid = Some(mean of points) (contour.rs:213-224), as PyContour does."""
    F = g.n_frames
    g.has_lumen_centroid = np.ones(F, dtype=np.uint8)
    lc = np.zeros((F, 3))
    for i in range(F):
        s = [0.0, 0.0, 0.0]
        for p in g.frame_lumen(i):
            s[0] += p[0]; s[1] += p[1]; s[2] += p[2]
        n = float(g.lumen_off[i + 1] - g.lumen_off[i])
        lc[i] = [s[0] / n, s[1] / n, s[2] / n]
    g.lumen_centroids = lc
    return g


# ---------------------------------------------------------------------------------------
def centerline_from_points(points) -> np.ndarray:
    p = O._pts(points)
    out = np.zeros(p.shape[0], dtype=CL_DTYPE)
    rc = lib().orc_centerline_from_points(O._p(p), p.shape[0], O._p(out))
    if rc:
        raise RuntimeError("centerline needs at least two points")
    return out


def make_centerline(xyz, tangents, radius=None, branch_id=None) -> np.ndarray:
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(tangents, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(xyz.shape[0], dtype=CL_DTYPE)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out["tx"], out["ty"], out["tz"] = t[:, 0], t[:, 1], t[:, 2]
    if radius is not None:
        out["radius"] = radius
    if branch_id is not None:
        out["branch_id"] = branch_id
    return out


def find_ref_idx(cl, ref) -> int:
    cl = _cl(cl); r = _v3(ref)
    return int(lib().orc_cl_find_ref_idx(O._p(cl), cl.shape[0], O._p(r)))


def preprocess_centerline(cl, g: O.OracleGeometry):
    cl = _cl(cl)
    sp = C.c_double(0.0)
    cg = g._c()
    n = lib().orc_preprocess_centerline(O._p(cl), cl.shape[0], C.byref(cg), None, 0, C.byref(sp))
    if n < 0:
        raise RuntimeError({-1: "Centerline has no branch-0 points", -3: "Reference mesh has no frames"}.get(n, str(n)))
    out = np.zeros(n, dtype=CL_DTYPE)
    lib().orc_preprocess_centerline(O._p(cl), cl.shape[0], C.byref(cg), O._p(out), n, C.byref(sp))
    return out, sp.value


def sort_contour_points(points) -> np.ndarray:
    p = O._pts(points).copy()
    lib().orc_sort_contour_points(O._p(p), p.shape[0])
    return p


def rotate_geometry(g: O.OracleGeometry, angle: float):
    pk = _GeomPack([g])
    lib().orc_rotate_geometry(C.byref(pk.cls[0]), angle)


def newell_normal(points, centroid) -> np.ndarray:
    p = O._pts(points); c = _v3(centroid)
    out = np.zeros(3)
    lib().orc_newell_normal(O._p(p), p.shape[0], O._p(c), O._p(out))
    return out


def align_frame(points, centroid, clpoint):
    """Returns (translation(3), rotation(3,3), pivot(3))."""
    p = O._pts(points)
    c = None if centroid is None else _v3(centroid)
    clp = np.ascontiguousarray(np.asarray(clpoint, dtype=CL_DTYPE).reshape(1))
    tf = _FrameTf()
    lib().orc_align_frame(O._p(p), p.shape[0], 0 if c is None else 1, O._p(c), O._p(clp), C.byref(tf))
    return np.array(tf.t[:]), np.array(tf.r[:]).reshape(3, 3), np.array(tf.pivot[:])


def tf_apply(translation, rotation, pivot, point) -> np.ndarray:
    tf = _FrameTf()
    tf.t[:] = list(map(float, translation))
    tf.r[:] = list(map(float, np.asarray(rotation, dtype=np.float64).reshape(9)))
    tf.pivot[:] = list(map(float, pivot))
    q = lib().orc_tf_apply(C.byref(tf), O._Point(*map(float, point)))
    return np.array([q.x, q.y, q.z])


def rotation_from_axis_angle(axis, angle) -> np.ndarray:
    a = _v3(axis); r = np.zeros(9)
    lib().orc_rotation_from_axis_angle(O._p(a), float(angle), O._p(r))
    return r.reshape(3, 3)


def rotate_contour_around_centroid(points, centroid, angle) -> np.ndarray:
    p = O._pts(points).copy()
    c = None if centroid is None else _v3(centroid)
    lib().orc_rotate_contour_around_centroid(O._p(p), p.shape[0], 0 if c is None else 1, O._p(c), float(angle))
    return p


def apply_transformations(geoms: Sequence[O.OracleGeometry], cl, ref_pt) -> int:
    cl = _cl(cl); r = _v3(ref_pt)
    pk = _GeomPack(geoms)
    return int(lib().orc_apply_transformations(pk.ptr, len(geoms), O._p(cl), cl.shape[0], O._p(r)))


def best_rotation_three_point(points, centroid, index_reference, p_main, p_ccw, p_cw, angle_step, clpoint) -> float:
    p = O._pts(points)
    c = None if centroid is None else _v3(centroid)
    clp = np.ascontiguousarray(np.asarray(clpoint, dtype=CL_DTYPE).reshape(1))
    a, b, d = _v3(p_main), _v3(p_ccw), _v3(p_cw)
    return lib().orc_best_rotation_three_point(O._p(p), p.shape[0], 0 if c is None else 1, O._p(c),
                                               int(index_reference), O._p(a), O._p(b), O._p(d),
                                               float(angle_step), O._p(clp))


def refine_alignment_hausdorff(geoms: Sequence[O.OracleGeometry], cl, initial_cl_ref_idx, initial_rotation, points,
                               angle_search_range, angle_step, index_search_range):
    """Returns (best_angle, best_cl_idx, min_hausdorff, costs of all evaluated candidates)."""
    cl = _cl(cl); pts = O._pts(points)
    pk = _GeomPack(geoms)
    ba, bi, mh, ne = C.c_double(0), C.c_size_t(0), C.c_double(0), C.c_size_t(0)
    import math
    n_ang = int(math.floor(2.0 * angle_search_range / angle_step)) + 3
    cap = (2 * int(index_search_range) + 1) * n_ang
    costs = np.zeros(cap, dtype=np.float64)
    rc = lib().orc_refine_alignment_hausdorff(pk.ptr, len(geoms), O._p(cl), cl.shape[0], int(initial_cl_ref_idx),
                                              float(initial_rotation), O._p(pts), pts.shape[0],
                                              float(angle_search_range), float(angle_step), int(index_search_range),
                                              C.byref(ba), C.byref(bi), C.byref(mh), O._p(costs), cap, C.byref(ne))
    if rc:
        raise RuntimeError(f"oracle refine error {rc}")
    assert ne.value <= cap
    return ba.value, int(bi.value), mh.value, costs[: ne.value].copy()


_ERR = {-1: "Centerline has no branch-0 points", -3: "Reference mesh has no frames",
        -4: "Couldn't find ref frame idx", -5: "missing reference point",
        -7: "at most two geometries"}


def align_walls(geoms, anomalous=True):
    """align_walls (align.rs:589-595), in place.  PARITY UNPINNED (no reference test)."""
    pk = _GeomPack(geoms)
    lib().orc_align_walls(pk.ptr, len(geoms), int(anomalous))


def align_three_point(cl, geoms, ref_point_index, p_main, p_ccw, p_cw, angle_step, align_wall_anomalous=False):
    cl = _cl(cl); pk = _GeomPack(geoms)
    a, b, d = _v3(p_main), _v3(p_ccw), _v3(p_cw)
    sp, rot = C.c_double(0), C.c_double(0)
    rc = lib().orc_align_three_point(O._p(cl), cl.shape[0], pk.ptr, len(geoms), int(ref_point_index), O._p(a),
                                     O._p(b), O._p(d), float(angle_step), int(align_wall_anomalous),
                                     C.byref(sp), C.byref(rot))
    if rc:
        raise RuntimeError(_ERR.get(rc, str(rc)))
    return sp.value, rot.value


def align_manual(cl, geoms, rotation_angle_deg, ref_pt, align_wall_anomalous=False):
    cl = _cl(cl); pk = _GeomPack(geoms)
    r = _v3(ref_pt)
    sp, rot = C.c_double(0), C.c_double(0)
    rc = lib().orc_align_manual(O._p(cl), cl.shape[0], pk.ptr, len(geoms), float(rotation_angle_deg), O._p(r),
                                int(align_wall_anomalous), C.byref(sp), C.byref(rot))
    if rc:
        raise RuntimeError(_ERR.get(rc, str(rc)))
    return sp.value, rot.value


def align_combined(cl, geoms, ref_point_index, p_main, p_ccw, p_cw, points, angle_step, refine_angle_range,
                   refine_index_range, align_wall_anomalous=False):
    cl = _cl(cl); pk = _GeomPack(geoms); pts = O._pts(points)
    a, b, d = _v3(p_main), _v3(p_ccw), _v3(p_cw)
    sp, rot, ri = C.c_double(0), C.c_double(0), C.c_size_t(0)
    rc = lib().orc_align_combined(O._p(cl), cl.shape[0], pk.ptr, len(geoms), int(ref_point_index), O._p(a),
                                  O._p(b), O._p(d), O._p(pts), pts.shape[0], float(angle_step),
                                  float(refine_angle_range), int(refine_index_range), int(align_wall_anomalous),
                                  C.byref(sp), C.byref(rot), C.byref(ri))
    if rc:
        raise RuntimeError(_ERR.get(rc, str(rc)))
    return sp.value, rot.value, int(ri.value)
