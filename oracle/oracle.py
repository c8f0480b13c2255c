"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module -- as the checker / reported baseline, never as the product path.
The C sources restate the reference (yungselm/multimoda-rs) line by line; see
``mm_oracle.h`` for the file:line map and the parity status (pinned by the reference's
own known-answer tests, ``tests/test_oracle_kat.py``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmm_oracle.so")


def build(force: bool = False) -> str:
    """Compile ``libmm_oracle.so`` with gcc (building the checker is not using it)."""
    deps = [os.path.join(_HERE, f) for f in ("mm_oracle.c", "mm_oracle.h", "mm_oracle_cl.c", "mm_oracle_cl.h",
                                             "mm_oracle_ccta.c", "mm_oracle_ccta.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in deps
    )
    if force or stale:
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:   # ranks of one node build once
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                subprocess.check_call(["make", "-C", _HERE, "libmm_oracle.so"], stdout=subprocess.DEVNULL)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


class _Point(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class _AlignLog(C.Structure):
    _fields_ = [
        ("contour_id", C.c_uint32),
        ("matched_to", C.c_uint32),
        ("rot_deg", C.c_double),
        ("tx", C.c_double),
        ("ty", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
    ]


class _Geometry(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32),
        ("id", C.c_void_p),
        ("lumen_id", C.c_void_p),
        ("orig_frame", C.c_void_p),
        ("centroid", C.c_void_p),
        ("lumen_off", C.c_void_p),
        ("lumen", C.c_void_p),
        ("has_catheter", C.c_int32),
        ("cath_off", C.c_void_p),
        ("cath", C.c_void_p),
        ("extra_off", C.c_void_p),
        ("extra", C.c_void_p),
        ("has_ref", C.c_void_p),
        ("ref", C.c_void_p),
        ("lumen_centroid", C.c_void_p),
    ]


_COST_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    path = _LIB_PATH
    if os.environ.get("MM_ORACLE_ASAN"):    # sanitizer build of the checker (make libmm_oracle_asan.so; LD_PRELOAD libasan)
        path = os.path.join(_HERE, "libmm_oracle_asan.so")
    L = C.CDLL(path)
    P = C.c_void_p
    L.orc_hausdorff_xy.restype = C.c_double
    L.orc_hausdorff_xy.argtypes = [P, P, C.c_size_t, P, P, C.c_size_t]
    L.orc_hausdorff.restype = C.c_double
    L.orc_hausdorff.argtypes = [P, C.c_size_t, P, C.c_size_t]
    L.orc_directed_hausdorff.restype = C.c_double
    L.orc_directed_hausdorff.argtypes = [P, C.c_size_t, P, C.c_size_t]
    L.orc_search_angles.restype = C.c_size_t
    L.orc_search_angles.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                    P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.orc_search_range.restype = C.c_double
    L.orc_search_range.argtypes = [_COST_FN, P, C.c_double, C.c_double, C.c_int, C.c_double,
                                   C.c_double, C.c_int]
    L.orc_cost_within.restype = C.c_double
    L.orc_cost_within.argtypes = [P, C.c_size_t, P, C.c_size_t, C.c_double, C.c_double, C.c_double]
    L.orc_cost_between.restype = C.c_double
    L.orc_cost_between.argtypes = [P, C.c_size_t, P, C.c_size_t, C.c_double, C.c_double, C.c_double]
    L.orc_find_best_rotation.restype = C.c_double
    L.orc_find_best_rotation.argtypes = [P, C.c_size_t, P, C.c_size_t, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_int, C.c_int]
    L.orc_bruteforce_rotation.restype = C.c_double
    L.orc_bruteforce_rotation.argtypes = [P, C.c_size_t, P, C.c_size_t, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_int]
    L.orc_count_evals.restype = C.c_size_t
    L.orc_count_evals.argtypes = [C.c_double, C.c_double, C.c_int]
    L.orc_downsample.restype = C.c_size_t
    L.orc_downsample.argtypes = [P, C.c_size_t, C.c_size_t, P]
    L.orc_align_within_chain.restype = C.c_int
    L.orc_align_within_chain.argtypes = [C.POINTER(_Geometry), C.c_double, C.c_double, C.c_int,
                                         C.c_size_t, P, C.c_int]
    L.orc_align_between.restype = C.c_int
    L.orc_align_between.argtypes = [C.POINTER(_Geometry), C.POINTER(_Geometry), C.c_double,
                                    C.c_double, C.c_size_t, C.POINTER(C.c_double), C.c_int]
    L.orc_extract_between_points.restype = C.c_size_t
    L.orc_extract_between_points.argtypes = [C.POINTER(_Geometry), C.c_size_t, P, C.c_size_t]
    L.orc_frame_translate.restype = None
    L.orc_frame_translate.argtypes = [C.POINTER(_Geometry), C.c_int32, C.c_double, C.c_double, C.c_double]
    L.orc_frame_rotate.restype = None
    L.orc_frame_rotate.argtypes = [C.POINTER(_Geometry), C.c_int32, C.c_double, C.c_double, C.c_double]
    L.orc_catheter_lumen_vec.restype = C.c_size_t
    L.orc_catheter_lumen_vec.argtypes = [C.POINTER(_Geometry), C.c_int32, C.c_size_t, C.c_int,
                                         C.c_size_t, P]
    L.orc_refine_angles.restype = C.c_size_t
    L.orc_refine_angles.argtypes = [C.c_double, C.c_double, C.c_double, P, C.c_size_t]
    L.orc_filter_points_in_region.restype = C.c_size_t
    L.orc_filter_points_in_region.argtypes = [P, C.c_size_t, P, P, P, C.c_size_t]
    L.orc_refine_downsample_count.restype = C.c_size_t
    L.orc_refine_downsample_count.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
    _lib = L
    return L


def _pts(a) -> np.ndarray:
    """(N,2|3) array-like -> C-contiguous (N,3) f64 (z=0 if absent)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 3)
    if a.shape[1] == 2:
        a = np.concatenate([a, np.zeros((a.shape[0], 1))], axis=1)
    return np.ascontiguousarray(a)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------
# metric / search
# ---------------------------------------------------------------------------------------
def hausdorff(a, b) -> float:
    a, b = _pts(a), _pts(b)
    return lib().orc_hausdorff(_p(a), a.shape[0], _p(b), b.shape[0])


def directed_hausdorff(a, b) -> float:
    a, b = _pts(a), _pts(b)
    return lib().orc_directed_hausdorff(_p(a), a.shape[0], _p(b), b.shape[0])


def search_angles(step_deg, range_deg, center=None, limes_deg=None):
    """Returns (angles ndarray, degenerate flag, early value)."""
    if limes_deg is None:
        limes_deg = range_deg
    deg = C.c_int(0)
    early = C.c_double(0.0)
    hc, c = (0, 0.0) if center is None else (1, float(center))
    n = lib().orc_search_angles(step_deg, range_deg, hc, c, limes_deg, None, 0, C.byref(deg), C.byref(early))
    out = np.empty(n, dtype=np.float64)
    if n:
        lib().orc_search_angles(step_deg, range_deg, hc, c, limes_deg, _p(out), n, C.byref(deg), C.byref(early))
    return out, bool(deg.value), early.value


def search_range(cost, step_deg, range_deg, center=None, limes_deg=180.0, n_threads=1) -> float:
    cb = _COST_FN(lambda ang, _ctx: float(cost(ang)))
    hc, c = (0, 0.0) if center is None else (1, float(center))
    # python callbacks are not thread-safe to call concurrently without the GIL dance; keep 1 thread
    return lib().orc_search_range(cb, None, step_deg, range_deg, hc, c, limes_deg, 1)


def cost_within(ref, tgt, angle, cx, cy) -> float:
    r, t = _pts(ref), _pts(tgt)
    return lib().orc_cost_within(_p(r), r.shape[0], _p(t), t.shape[0], angle, cx, cy)


def cost_between(ref, tgt, angle, cx, cy) -> float:
    r, t = _pts(ref), _pts(tgt)
    return lib().orc_cost_between(_p(r), r.shape[0], _p(t), t.shape[0], angle, cx, cy)


def costs_over_angles(ref, tgt, angles, cx, cy, between=False) -> np.ndarray:
    f = cost_between if between else cost_within
    return np.array([f(ref, tgt, float(a), cx, cy) for a in angles], dtype=np.float64)


def find_best_rotation(ref, tgt, step_deg, range_deg, cx, cy, between=False, n_threads=1) -> float:
    r, t = _pts(ref), _pts(tgt)
    return lib().orc_find_best_rotation(_p(r), r.shape[0], _p(t), t.shape[0], step_deg, range_deg,
                                        cx, cy, int(between), n_threads)


def bruteforce_rotation(ref, tgt, step_deg, range_deg, cx, cy, n_threads=1) -> float:
    r, t = _pts(ref), _pts(tgt)
    return lib().orc_bruteforce_rotation(_p(r), r.shape[0], _p(t), t.shape[0], step_deg, range_deg,
                                         cx, cy, n_threads)


def count_evals(step_deg, range_deg, bruteforce) -> int:
    return int(lib().orc_count_evals(step_deg, range_deg, int(bruteforce)))


def downsample(points, n) -> np.ndarray:
    p = _pts(points)
    out = np.empty((max(min(p.shape[0], n), 0), 3), dtype=np.float64)
    k = lib().orc_downsample(_p(p), p.shape[0], n, _p(out))
    return out[:k]


# ---------------------------------------------------------------------------------------
# flat geometry (mirror of orc_geometry)
# ---------------------------------------------------------------------------------------
@dataclass
class OracleGeometry:
    """numpy-backed mirror of ``orc_geometry``; arrays are mutated in place by the chain."""

    ids: np.ndarray            # (F,) u32
    lumen_ids: np.ndarray      # (F,) u32
    orig_frames: np.ndarray    # (F,) u32
    centroids: np.ndarray      # (F,3) f64
    lumen_off: np.ndarray      # (F+1,) i64
    lumen: np.ndarray          # (N,3) f64
    cath_off: Optional[np.ndarray] = None
    cath: Optional[np.ndarray] = None
    extra_off: Optional[np.ndarray] = None
    extra: Optional[np.ndarray] = None
    has_ref: Optional[np.ndarray] = None   # (F,) u8
    ref: Optional[np.ndarray] = None       # (F,3) f64
    label: str = ""
    _keep: list = field(default_factory=list, repr=False)
    # centerline placement only (orc_clgeom): the lumen contour's own centroid and the extras layout
    has_lumen_centroid: Optional[np.ndarray] = None   # (F,) u8
    lumen_centroids: Optional[np.ndarray] = None      # (F,3) f64
    n_extra_kinds: int = 0
    extra_kind_off: Optional[np.ndarray] = None       # (F*K+1,) i64
    lumen_aortic: Optional[np.ndarray] = None         # (lumen points,) u8  ContourPoint.aortic, follows its point in sorts
    wall_aortic: Optional[np.ndarray] = None          # (wall points,) u8
    wall_kind1: int = 0                               # 1 + index of the Wall contour among the K kinds; 0 = none

    @property
    def n_frames(self) -> int:
        return int(self.ids.shape[0])

    @staticmethod
    def from_frames(lumens, catheters=None, centroids=None, ids=None, orig_frames=None,
                    ref_points=None, label="") -> "OracleGeometry":
        """lumens: list of (n_i, 3) arrays. centroids default to the lumen mean
        (contour.rs:213-224 compute_centroid: sequential sums / n)."""
        F = len(lumens)
        lum = [_pts(l) for l in lumens]
        off = np.zeros(F + 1, dtype=np.int64)
        off[1:] = np.cumsum([l.shape[0] for l in lum])
        if centroids is None:
            cen = np.zeros((F, 3))
            for i, l in enumerate(lum):
                s = [0.0, 0.0, 0.0]
                for p in l:                      # sequential fold like the reference
                    s[0] += p[0]; s[1] += p[1]; s[2] += p[2]
                cen[i] = [s[0] / len(l), s[1] / len(l), s[2] / len(l)]
        else:
            cen = np.array(centroids, dtype=np.float64).reshape(F, 3).copy()
        g = OracleGeometry(
            ids=np.arange(F, dtype=np.uint32) if ids is None else np.asarray(ids, dtype=np.uint32).copy(),
            lumen_ids=np.arange(F, dtype=np.uint32) if ids is None else np.asarray(ids, dtype=np.uint32).copy(),
            orig_frames=(np.arange(F, dtype=np.uint32) if orig_frames is None
                         else np.asarray(orig_frames, dtype=np.uint32).copy()),
            centroids=np.ascontiguousarray(cen),
            lumen_off=off,
            lumen=np.ascontiguousarray(np.concatenate(lum, axis=0)) if F else np.zeros((0, 3)),
            label=label,
        )
        if catheters is not None:
            cat = [_pts(c) for c in catheters]
            coff = np.zeros(F + 1, dtype=np.int64)
            coff[1:] = np.cumsum([c.shape[0] for c in cat])
            g.cath_off = coff
            g.cath = np.ascontiguousarray(np.concatenate(cat, axis=0))
        g.has_ref = np.zeros(F, dtype=np.uint8)
        g.ref = np.zeros((F, 3), dtype=np.float64)
        if ref_points:
            for i, p in ref_points.items():
                g.has_ref[i] = 1
                g.ref[i] = p
        return g

    def copy(self) -> "OracleGeometry":
        cp = lambda a: None if a is None else a.copy()
        g = OracleGeometry(cp(self.ids), cp(self.lumen_ids), cp(self.orig_frames), cp(self.centroids),
                           cp(self.lumen_off), cp(self.lumen), cp(self.cath_off), cp(self.cath),
                           cp(self.extra_off), cp(self.extra), cp(self.has_ref), cp(self.ref), self.label)
        g.has_lumen_centroid, g.lumen_centroids = cp(self.has_lumen_centroid), cp(self.lumen_centroids)
        g.n_extra_kinds, g.extra_kind_off = self.n_extra_kinds, cp(self.extra_kind_off)
        g.lumen_aortic, g.wall_aortic, g.wall_kind1 = cp(self.lumen_aortic), cp(self.wall_aortic), self.wall_kind1
        return g

    def frame_lumen(self, i) -> np.ndarray:
        return self.lumen[self.lumen_off[i]:self.lumen_off[i + 1]]

    def frame_cath(self, i) -> np.ndarray:
        return self.cath[self.cath_off[i]:self.cath_off[i + 1]]

    def _c(self) -> _Geometry:
        for name in ("ids", "lumen_ids", "orig_frames"):
            a = getattr(self, name)
            assert a.dtype == np.uint32 and a.flags.c_contiguous
        assert self.centroids.dtype == np.float64 and self.centroids.flags.c_contiguous
        assert self.lumen_off.dtype == np.int64 and self.lumen.flags.c_contiguous
        g = _Geometry()
        g.n_frames = self.n_frames
        g.id = _p(self.ids)
        g.lumen_id = _p(self.lumen_ids)
        g.orig_frame = _p(self.orig_frames)
        g.centroid = _p(self.centroids)
        g.lumen_off = _p(self.lumen_off)
        g.lumen = _p(self.lumen)
        g.has_catheter = 1 if self.cath_off is not None else 0
        g.cath_off = _p(self.cath_off)
        g.cath = _p(self.cath)
        g.extra_off = _p(self.extra_off)
        g.extra = _p(self.extra)
        g.has_ref = _p(self.has_ref)
        g.ref = _p(self.ref)
        if self.lumen_centroids is not None and (self.has_lumen_centroid is None or bool(np.all(self.has_lumen_centroid))):
            assert self.lumen_centroids.dtype == np.float64 and self.lumen_centroids.flags.c_contiguous
            g.lumen_centroid = _p(self.lumen_centroids)        # Frame.lumen.centroid, tracked through translations
        return g


_ERR = {-1: "Geometry contains no frames", -2: "Lumen contours have no points",
        -3: "sample_size must be > 0"}


def align_within_chain(g: OracleGeometry, step_deg, range_deg, bruteforce, sample_size, n_threads=1):
    """align_frames_in_geometry lines 24-134 (in place). Returns list of 7-tuples
    (id, matched_to, rot_deg, tx, ty, cx, cy) like binding/functions.rs:26-40."""
    F = g.n_frames
    logs = (_AlignLog * max(F - 1, 1))()
    cg = g._c()
    rc = lib().orc_align_within_chain(C.byref(cg), step_deg, range_deg, int(bruteforce),
                                      int(sample_size), C.cast(logs, C.c_void_p), n_threads)
    if rc != 0:
        raise RuntimeError(_ERR.get(rc, f"oracle error {rc}"))
    return [(l.contour_id, l.matched_to, l.rot_deg, l.tx, l.ty, l.cx, l.cy) for l in logs[: F - 1]]


def align_between(a: OracleGeometry, b: OracleGeometry, rot_deg, step_rot_deg, sample_size, n_threads=1) -> float:
    best = C.c_double(0.0)
    ca, cb = a._c(), b._c()
    rc = lib().orc_align_between(C.byref(ca), C.byref(cb), rot_deg, step_rot_deg, int(sample_size),
                                 C.byref(best), n_threads)
    if rc != 0:
        raise RuntimeError(f"oracle align_between error {rc}")
    return best.value


def extract_between_points(g: OracleGeometry, sample_size) -> np.ndarray:
    cg = g._c()
    n = lib().orc_extract_between_points(C.byref(cg), int(sample_size), None, 0)
    out = np.empty((n, 3), dtype=np.float64)
    lib().orc_extract_between_points(C.byref(cg), int(sample_size), _p(out), n)
    return out


def frame_translate(g: OracleGeometry, i, dx, dy, dz):
    cg = g._c()
    lib().orc_frame_translate(C.byref(cg), i, dx, dy, dz)


def frame_rotate(g: OracleGeometry, i, angle, cx, cy):
    cg = g._c()
    lib().orc_frame_rotate(C.byref(cg), i, angle, cx, cy)


def catheter_lumen_vec(g: OracleGeometry, i, sample_size) -> np.ndarray:
    """Sets exactly as align_within.rs:45-59,173-191 builds them for frame i."""
    import math
    len0 = int(g.lumen_off[1] - g.lumen_off[0])
    ratio = float(sample_size) / float(len0)
    has_sc, sc = 0, 0
    if g.cath_off is not None:
        has_sc = 1
        sc = int(math.ceil(float(int(g.cath_off[1] - g.cath_off[0])) * ratio))
    cap = int(g.lumen_off[i + 1] - g.lumen_off[i]) + (int(g.cath_off[i + 1] - g.cath_off[i]) if has_sc else 0)
    out = np.empty((cap + 1, 3), dtype=np.float64)
    cg = g._c()
    n = lib().orc_catheter_lumen_vec(C.byref(cg), i, int(sample_size), has_sc, sc, _p(out))
    return out[:n].copy()


# ---------------------------------------------------------------------------------------
# refine_alignment_hausdorff helpers (align_algorithms.rs:339-451)
# ---------------------------------------------------------------------------------------
def refine_angles(initial, search_range, step) -> np.ndarray:
    n = lib().orc_refine_angles(initial, search_range, step, None, 0)
    out = np.empty(n, dtype=np.float64)
    if n:
        lib().orc_refine_angles(initial, search_range, step, _p(out), n)
    return out


def filter_points_in_region(points_xyz, start_xyz, end_xyz) -> np.ndarray:
    pts = _pts(points_xyz)
    s = _pts(np.asarray(start_xyz, dtype=np.float64).reshape(1, 3))
    e = _pts(np.asarray(end_xyz, dtype=np.float64).reshape(1, 3))
    idx = np.empty(pts.shape[0], dtype=np.int64)
    m = lib().orc_filter_points_in_region(_p(pts), pts.shape[0], _p(s), _p(e), _p(idx), pts.shape[0])
    return idx[:m].copy()


def refine_downsample_count(n_filtered, n_points_per_frame, n_frames) -> int:
    return int(lib().orc_refine_downsample_count(n_filtered, n_points_per_frame, n_frames))


def refine_select(candidate_sets):
    """Evaluation + selection of refine_alignment_hausdorff (:431-437): candidates in
    enumeration order, strict `<` keeps the first minimum. candidate_sets: list of
    (filtered_ccta_points, flat_geometry_points); returns (costs, best index or -1)."""
    costs = np.array([hausdorff(a, b) for a, b in candidate_sets], dtype=np.float64)
    best, best_cost = -1, np.finfo(np.float64).max
    for i, c in enumerate(costs):
        if c < best_cost:
            best, best_cost = i, c
    return costs, best
