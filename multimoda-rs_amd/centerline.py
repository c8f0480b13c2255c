"""Centerline placement path, mirroring the reference's Python entry points
``align_three_point`` / ``align_manual`` / ``align_combined`` (multimodars/_processing.py:1010-1300,
binding src/intravascular/binding/align.rs:83-460, implementation
src/intravascular/centerline_align/{align.rs, align_algorithms.rs, preprocessing.rs}).

Same argument names, meaning, defaults and error behaviour; geometries are ``FlatGeometry`` /
``GeometryPair`` containers instead of the PyO3 value classes and are returned as transformed
copies (the reference clones at the boundary too).  The three-point sweep and the frame placement
are host f64 (csrc/mm_centerline.cpp); every Hausdorff evaluation of ``align_combined``'s
refinement grid runs on the GPU; ``align_wall_anomalous=True`` applies the wall twist compensation
(align.rs:381-595, mm_align_walls) after the placement; ``write=True`` writes the OBJ / MTL /
texture files (export.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from . import geometry as G
from .io import EXTRA_KINDS

# mm_clpoint (include/mm_centerline.h)
CL_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("tx", "<f8"), ("ty", "<f8"), ("tz", "<f8"),
                     ("radius", "<f8"), ("branch_id", "<u4"), ("pad_", "<u4")])
assert CL_DTYPE.itemsize == 64


class Centerline:
    """types/native/centerline.rs ``Centerline``: points with unit tangents, radius and branch id."""

    def __init__(self, points: np.ndarray):
        points = np.ascontiguousarray(points)
        if points.dtype != CL_DTYPE or points.ndim != 1:
            raise TypeError("Centerline expects a 1-d array of CL_DTYPE records")
        self.points = points

    def __len__(self) -> int:
        return int(self.points.shape[0])

    @staticmethod
    def from_contour_points(xyz) -> "Centerline":
        """Centerline::from_contour_points (centerline.rs:14-42): forward-difference unit tangents."""
        a = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64).reshape(-1, 3))
        out = np.zeros(a.shape[0], dtype=CL_DTYPE)
        N.check(N.lib().mm_centerline_from_points(N._ptr(a), a.shape[0], N._ptr(out)), "from_contour_points")
        return Centerline(out)

    @staticmethod
    def from_arrays(xyz, tangents, radius=None, branch_id=None) -> "Centerline":
        a = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
        t = np.asarray(tangents, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(a.shape[0], dtype=CL_DTYPE)
        out["x"], out["y"], out["z"] = a[:, 0], a[:, 1], a[:, 2]
        out["tx"], out["ty"], out["tz"] = t[:, 0], t[:, 1], t[:, 2]
        if radius is not None:
            out["radius"] = radius
        if branch_id is not None:
            out["branch_id"] = branch_id
        return Centerline(out)

    def xyz(self) -> np.ndarray:
        return np.stack([self.points["x"], self.points["y"], self.points["z"]], axis=1)

    @property
    def branch_start_indices(self):
        """Centerline.branch_start_indices (centerline.rs): first point of every branch."""
        return [s for s, _e in self._branch_runs()]

    # -- branches (centerline.rs: points of one branch are contiguous, branch 0 = main vessel) -------
    def _branch_runs(self):
        b = self.points["branch_id"]
        if len(b) == 0:
            return []
        starts = [0] + [i for i in range(1, len(b)) if b[i] != b[i - 1]]
        return [(s, e) for s, e in zip(starts, starts[1:] + [len(b)])]

    def get_branch(self, branch_id: int) -> "Centerline":
        """py_centerline.rs:211-232: the points of one branch as a single-branch centerline (branch 0)."""
        sel = self.points[self.points["branch_id"] == branch_id].copy()
        if sel.shape[0] == 0:
            raise ValueError(f"branch_id {branch_id} not found in centerline")
        sel["branch_id"] = 0
        return Centerline(sel)

    def mean_spacing(self) -> float:
        """centerline.rs:304-320: mean distance of consecutive points of branch 0 (1.0 if fewer than two)."""
        runs = self._branch_runs()
        if not runs or runs[0][1] - runs[0][0] < 2:
            return 1.0
        xyz = self.xyz()[runs[0][0]:runs[0][1]]
        d = xyz[1:] - xyz[:-1]
        dist = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        return float(np.add.accumulate(dist)[-1]) / float(len(dist))

    def resample(self, spacing_mm: float) -> "Centerline":
        """Centerline::resample (centerline.rs:717-796): every branch resampled to even arc-length spacing by
        linear interpolation (samples at 0, s, 2s, ... and the end point), branch ids renumbered in order,
        tangents recomputed as normalised forward differences inside a branch (:377-391).  Returns a new
        centerline, like the Python binding (py_centerline.rs:287-296)."""
        if len(self) == 0 or spacing_mm <= 1e-12:
            return Centerline(self.points.copy())
        out = []
        for new_id, (s0, s1) in enumerate(self._branch_runs()):
            pts = self.points[s0:s1]
            xyz = np.stack([pts["x"], pts["y"], pts["z"]], axis=1)
            n = len(pts)
            if n < 2:
                res = pts.copy()
            else:
                cum = [0.0]
                for i in range(1, n):
                    d = xyz[i - 1] - xyz[i]                               # distance_to(other): self - other
                    cum.append(cum[-1] + math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
                total = cum[-1]
                if total < 1e-12:
                    res = pts.copy()
                else:
                    targets, s = [], 0.0
                    while s < total:
                        targets.append(s)
                        s += spacing_mm
                    targets.append(total)
                    res = np.zeros(len(targets), dtype=CL_DTYPE)
                    seg = 0
                    for k, t in enumerate(targets):
                        while seg < n - 2 and cum[seg + 1] < t:
                            seg += 1
                        a, b = cum[seg], cum[seg + 1]
                        frac = 0.0 if abs(b - a) < 1e-12 else (t - a) / (b - a)
                        p0, p1 = xyz[seg], xyz[seg + 1]
                        res[k]["x"] = p0[0] + frac * (p1[0] - p0[0])
                        res[k]["y"] = p0[1] + frac * (p1[1] - p0[1])
                        res[k]["z"] = p0[2] + frac * (p1[2] - p0[2])
                        res[k]["radius"] = pts["radius"][seg] + frac * (pts["radius"][seg + 1] - pts["radius"][seg])
            res["branch_id"] = new_id
            out.append(res)
        pts = np.concatenate(out)
        n = len(pts)
        for i in range(n):                                                # recompute_tangents
            if i + 1 < n and pts["branch_id"][i] == pts["branch_id"][i + 1]:
                d = np.array([pts["x"][i + 1] - pts["x"][i], pts["y"][i + 1] - pts["y"][i], pts["z"][i + 1] - pts["z"][i]])
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = d / math.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
            elif i > 0 and pts["branch_id"][i - 1] == pts["branch_id"][i]:
                t = np.array([pts["tx"][i - 1], pts["ty"][i - 1], pts["tz"][i - 1]])
            else:
                t = np.zeros(3)
            pts["tx"][i], pts["ty"][i], pts["tz"][i] = t
        return Centerline(pts)

    def find_reference_cl_point_idx(self, reference_point) -> int:
        r = np.ascontiguousarray(np.asarray(reference_point, dtype=np.float64).reshape(3))
        return int(N.lib().mm_centerline_find_ref_idx(N._ptr(self.points), len(self), N._ptr(r)))


def numpy_to_centerline(arr) -> Centerline:
    """multimodars/_converters.py:605-686: (N,3) array -> centerline; NaNs are interpolated along the
    index axis, an all-NaN column or fewer than two points raise ValueError."""
    arr = np.asarray(arr, dtype=float)
    if arr.ndim != 2 or arr.shape[1] != 3:
        raise ValueError("Input must be a (N,3) array")
    n = arr.shape[0]
    if n == 0:
        raise ValueError("Input array must contain at least one point")
    if np.isnan(arr).any():
        idx = np.arange(n)
        fixed = arr.copy()
        for col in range(3):
            ok = ~np.isnan(arr[:, col])
            if ok.sum() == 0:
                raise ValueError(f"All values are NaN for coordinate column {col}; cannot build centerline.")
            if ok.sum() < n:
                fixed[:, col] = np.interp(idx, idx[ok], arr[ok, col])
        arr = fixed
    if arr.shape[0] < 2:
        raise ValueError("Centerline must contain at least two points after cleaning/interpolation.")
    return Centerline.from_contour_points(arr)


def read_centerline_vtp(file_path: str) -> Centerline:
    """``read_centerline_vtp`` (multimodars/_processing.py:1355-1385, src/intravascular/io/input.rs:259-461): an
    ASCII VTK PolyData (``.vtp``) centerline, e.g. a VMTK export.  Every VTK line is a branch; branches are numbered
    by descending ARC LENGTH (branch 0 = the geometrically longest line, not the one with most points); tangents are
    normalised forward differences inside a branch, the last point of a branch repeats its predecessor's (zero for a
    single-point branch or coincident points); radii from ``MaximumInscribedSphereRadius`` if its length matches,
    else 0.  Binary or appended-data files are refused with the reference's messages."""
    from .io import _RE_F64, _RE_U32
    try:
        with open(file_path, "rb") as fh:
            raw = fh.read()
    except OSError as e:
        raise RuntimeError(f"cannot open {file_path!r}: {e}") from e
    if any(b < 0x09 or 0x0D < b < 0x20 for b in raw[:512]):
        raise RuntimeError(f"{file_path!r} appears to be a binary VTP file; only ASCII-format VTP is supported. "
                           "Re-export from your software with 'ASCII' data mode.")
    try:
        xml = raw.decode("utf-8")
    except UnicodeDecodeError as e:
        raise RuntimeError(f"{file_path!r}: not valid UTF-8") from e
    for fmt in ('format="binary"', 'format="appended"'):
        if fmt in xml:
            raise RuntimeError(f"{file_path!r}: binary-encoded DataArrays detected ({fmt}); only ASCII format is "
                               "supported. Re-export with 'ASCII' data mode.")

    def section(tag):                                         # extract_section (:295-306)
        start = xml.find("<" + tag)
        if start < 0:
            raise RuntimeError(f"VTP: <{tag}> section not found")
        end = xml.find(f"</{tag}>", start)
        if end < 0:
            raise RuntimeError(f"VTP: </{tag}> not found")
        return xml[start:end + len(tag) + 3]

    def dataarray_text(sec, name):                            # :308-329
        pos = sec.find(f'Name="{name}"')
        if pos < 0:
            raise RuntimeError(f'VTP: DataArray Name="{name}" not found')
        da = sec.rfind("<DataArray", 0, pos)
        if da < 0:
            raise RuntimeError(f'VTP: no <DataArray before Name="{name}"')
        gt = sec.find(">", da)
        if gt < 0:
            raise RuntimeError(f'VTP: unclosed <DataArray Name="{name}">')
        close = sec.find("</DataArray>", gt + 1)
        if close < 0:
            raise RuntimeError(f'VTP: no </DataArray> for Name="{name}"')
        text = sec[gt + 1:close].strip()
        lt = text.find("<")                                   # <InformationKey> nodes inside the Points array
        return (text if lt < 0 else text[:lt]).strip()

    def nums(text, rx, conv):                                 # parse_nums (:331-343), Rust's number grammars
        out = []
        for tok in text.split():
            if not rx.match(tok):
                raise RuntimeError(f"VTP: bad number '{tok}'")
            out.append(conv(tok))
        return out

    pts_raw = nums(dataarray_text(section("Points"), "Points"), _RE_F64, float)
    if len(pts_raw) % 3:
        raise RuntimeError(f"VTP: Points array length {len(pts_raw)} not divisible by 3")
    coords = np.asarray(pts_raw, dtype=np.float64).reshape(-1, 3)
    n_pts = coords.shape[0]
    try:
        radii = nums(dataarray_text(section("PointData"), "MaximumInscribedSphereRadius"), _RE_F64, float)
    except RuntimeError:
        radii = []
    if len(radii) != n_pts:
        radii = [0.0] * n_pts
    lines = section("Lines")
    connectivity = nums(dataarray_text(lines, "connectivity"), _RE_U32, int)
    offsets = nums(dataarray_text(lines, "offsets"), _RE_U32, int)
    if not offsets:
        raise RuntimeError("VTP: Lines section is empty (no branches)")
    if offsets[-1] != len(connectivity):
        raise RuntimeError(f"VTP: last offset ({offsets[-1]}) != connectivity length ({len(connectivity)})")
    branches = [connectivity[a:b] for a, b in zip([0] + offsets[:-1], offsets)]
    for br in branches:
        for i in br:
            if i >= n_pts:
                raise RuntimeError(f"VTP: connectivity index {i} out of range ({n_pts} points)")

    def arc_length(br):                                       # :383-393, sequential f64 sum
        total = 0.0
        for a, b in zip(br[:-1], br[1:]):
            d = coords[b] - coords[a]
            total += math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        return total
    lengths = [arc_length(br) for br in branches]
    order = sorted(range(len(branches)), key=lambda k: -lengths[k])      # longest first (ties: file order)
    out = np.zeros(len(connectivity), dtype=CL_DTYPE)
    k = 0
    for branch_id, vi in enumerate(order):
        br = branches[vi]
        for li, pi in enumerate(br):
            x, y, z = coords[pi]
            if li + 1 < len(br):
                d = coords[br[li + 1]] - coords[pi]
                nrm = math.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
                t = d / nrm if nrm > 1e-12 else np.zeros(3)
            elif li > 0:
                t = np.array([out["tx"][k - 1], out["ty"][k - 1], out["tz"][k - 1]])
            else:
                t = np.zeros(3)
            out[k] = (x, y, z, t[0], t[1], t[2], radii[pi], branch_id, 0)
            k += 1
    return Centerline(out)


def preprocess_centerline(centerline: Centerline, ref_mesh: G.FlatGeometry) -> Tuple[Centerline, float]:
    """preprocessing.rs:16-108 -> (resampled centerline, spacing in mm)."""
    cg = ref_mesh.c_struct()
    sp = C.c_double(0.0)
    L = N.lib()
    n = L.mm_centerline_preprocess(N._ptr(centerline.points), len(centerline), C.byref(cg), None, 0, C.byref(sp))
    if n < 0:
        N.check(int(n), "preprocess_centerline")
    out = np.zeros(int(n), dtype=CL_DTYPE)
    L.mm_centerline_preprocess(N._ptr(centerline.points), len(centerline), C.byref(cg), N._ptr(out), int(n), C.byref(sp))
    return Centerline(out), sp.value


def with_lumen_centroids(g: G.FlatGeometry) -> G.FlatGeometry:
    """Contour::compute_centroid (contour.rs:213-224) for every lumen: what a PyContour carries."""
    F = g.n_frames
    lc = N.contour_centroids(g.lumen, g.lumen_off)          # the same sequential sums, all contours in one call
    g.has_lumen_centroid = np.ones(F, dtype=np.uint8)
    g.lumen_centroids = lc
    return g


class _ClPack:
    """ctypes views (mm_cl_geometry) of one or two FlatGeometry objects, kept alive for one call."""

    def __init__(self, geoms: Sequence[G.FlatGeometry]):
        self.keep = []
        self._flags = []
        self.gs = [g.c_struct() for g in geoms]
        self.cls = []
        for g, cg in zip(geoms, self.gs):
            c = N.MMClGeometry()
            c.g = C.pointer(cg)
            if g.lumen_centroids is not None:
                hl = g.has_lumen_centroid if g.has_lumen_centroid is not None else np.ones(g.n_frames, dtype=np.uint8)
                if hl.dtype != np.uint8 or hl.shape != (g.n_frames,) or g.lumen_centroids.shape != (g.n_frames, 3) \
                        or g.lumen_centroids.dtype != np.float64 or not g.lumen_centroids.flags.c_contiguous:
                    raise ValueError("FlatGeometry.lumen_centroids: expected C-contiguous float64 (F,3) + uint8 (F,)")
                self.keep.append(hl)
                c.has_lumen_centroid = N._ptr(hl)
                c.lumen_centroid = N._ptr(g.lumen_centroids)
            counts = g.meta.get("extra_counts")
            if g.extra_off is not None and counts:
                kinds = [k for k in EXTRA_KINDS if k in counts]
                per = np.stack([np.asarray(counts[k], dtype=np.int64) for k in kinds], axis=1)   # (F, K)
                ko = np.zeros(per.size + 1, dtype=np.int64)
                ko[1:] = np.cumsum(per.reshape(-1))
                if int(ko[-1]) != int(g.extra_off[-1]):
                    raise ValueError("meta['extra_counts'] does not match the extras blob")
                self.keep.append(ko)
                c.n_extra_kinds = len(kinds)
                c.extra_kind_off = N._ptr(ko)
                if "wall" in kinds and int(np.sum(counts["wall"])) > 0:
                    c.wall_kind1 = kinds.index("wall") + 1
            # ContourPoint.aortic: the flags follow their points through every sort inside the library
            for key, n_pts, field in (("lumen_aortic", g.lumen.shape[0], "lumen_aortic"),
                                      ("wall_aortic", int(np.sum(counts["wall"])) if counts and "wall" in counts else 0,
                                       "wall_aortic")):
                fl = g.meta.get(key)
                if fl is None or n_pts == 0:
                    continue
                fl = np.ascontiguousarray(fl, dtype=np.uint8).reshape(-1)
                if fl.shape[0] != n_pts:
                    raise ValueError(f"meta['{key}'] does not match the contour points")
                self.keep.append(fl)
                self._flags.append((g, key, fl))
                setattr(c, field, N._ptr(fl))
            self.cls.append(c)
        self.arr = (C.POINTER(N.MMClGeometry) * len(geoms))(*[C.pointer(c) for c in self.cls])

    @property
    def ptr(self):
        return C.cast(self.arr, C.c_void_p)

    def commit_flags(self) -> None:
        """Write the (possibly permuted) per-point aortic flags back into the geometries' meta."""
        for g, key, fl in self._flags:
            g.meta[key] = fl.astype(bool)


def _v3(p) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(3))


def _unpack(geometry):
    from .api import GeometryPair
    if isinstance(geometry, GeometryPair):
        a, b = geometry.geom_a.copy(), geometry.geom_b.copy()
        return [a, b], lambda: GeometryPair(a, b, geometry.label)
    if isinstance(geometry, G.FlatGeometry):
        a = geometry.copy()
        return [a], lambda: a
    raise TypeError("geometry must be a FlatGeometry or a GeometryPair")     # binding/align.rs:151


def _process_and_write(write: bool, result, case_name: str, output_dir: str, interpolation_steps: int,
                       watertight: bool, contour_types):
    """Processable::process_and_write (align.rs:19-61): a pair goes through process_case, a single
    geometry through write_single_geometry."""
    if not write:
        return
    from . import export as EX
    from . import frames as FR
    kinds = EX.DEFAULT_CONTOUR_TYPES if contour_types is None else contour_types
    try:
        if hasattr(result, "geom_a"):
            EX.process_case(case_name, FR.to_frames(result.geom_a), FR.to_frames(result.geom_b), output_dir,
                            interpolation_steps, watertight, kinds)
        else:
            EX.write_single_geometry(case_name, result, output_dir, watertight, kinds)
    except RuntimeError as e:
        raise RuntimeError(f"Failed to write obj: {e}") from e


def _py_walls() -> bool:
    """MM_PY_POSTPROC=1: the wall twist compensation runs in postproc.align_walls (the Python checker) instead of
    mm_align_walls."""
    return bool(os.environ.get("MM_PY_POSTPROC"))


def _align_walls(geoms: Sequence[G.FlatGeometry], anomalous: bool) -> None:
    """align_walls (align.rs:589-595) through the Python checker (postproc.align_walls); the product path is
    mm_align_walls inside mm_align_three_point / _manual / _combined (their align_wall_anomalous argument)."""
    if not anomalous or not _py_walls() or geoms[0].n_frames < 2:
        return
    from . import frames as FR
    from .api import _checker
    PP = _checker("postproc")
    for g in geoms:
        fr = PP.align_walls(FR.to_frames(g), True)
        h = FR.from_frames(fr, g.label, g.meta)
        g.extra_off, g.extra = h.extra_off, h.extra        # only wall points move


def align_walls(geometry, anomalous: bool = True):
    """``align_walls`` (align.rs:589-595) on a FlatGeometry or a GeometryPair -> a new object (mm_align_walls)."""
    geoms, rebuild = _unpack(geometry)
    pk = _ClPack(geoms)
    N.check(N.lib().mm_align_walls(pk.ptr, len(geoms), int(bool(anomalous))), "align_walls")
    pk.commit_flags()
    return rebuild()


def _ref_point_index(g: G.FlatGeometry) -> int:
    return int(g.meta.get("ref_point_index", 0))      # Frame.reference_point.point_index (0 from file, build.rs:406)


# ---- building blocks ---------------------------------------------------------------------
def rotate_geometry(g: G.FlatGeometry, angle_rad: float) -> None:
    """Geometry::rotate_geometry (geometry.rs:241-250), in place."""
    pk = _ClPack([g])
    N.check(N.lib().mm_rotate_geometry(C.byref(pk.cls[0]), float(angle_rad)), "rotate_geometry")
    pk.commit_flags()


def apply_transformations(geoms: Sequence[G.FlatGeometry], centerline: Centerline, ref_pt) -> int:
    """apply_transformations (align_algorithms.rs:511-535), in place; returns the frames placed."""
    pk = _ClPack(geoms)
    r = _v3(ref_pt)
    n = N.lib().mm_apply_transformations(pk.ptr, len(geoms), N._ptr(centerline.points), len(centerline), N._ptr(r))
    if n < 0:
        N.check(int(n), "apply_transformations")
    return int(n)


def best_rotation_three_point(lumen_xyz, centroid, index_reference: int, main_ref_pt, counterclockwise_ref_pt,
                              clockwise_ref_pt, angle_step: float, centerline_point) -> float:
    """best_rotation_three_point (align_algorithms.rs:263-336); angles in radians."""
    p = np.ascontiguousarray(np.asarray(lumen_xyz, dtype=np.float64).reshape(-1, 3))
    c = None if centroid is None else _v3(centroid)
    clp = np.ascontiguousarray(np.asarray(centerline_point, dtype=CL_DTYPE).reshape(1))
    a, b, d = _v3(main_ref_pt), _v3(counterclockwise_ref_pt), _v3(clockwise_ref_pt)
    out = C.c_double(0.0)
    N.check(N.lib().mm_best_rotation_three_point(N._ptr(p), p.shape[0], 0 if c is None else 1, N._ptr(c),
                                                 int(index_reference), N._ptr(a), N._ptr(b), N._ptr(d),
                                                 float(angle_step), N._ptr(clp), C.byref(out)),
            "best_rotation_three_point")
    return out.value


def refine_alignment_hausdorff(engine: N.Engine, geoms: Sequence[G.FlatGeometry], centerline: Centerline,
                               initial_cl_ref_idx: int, initial_rotation: float, points, angle_search_range: float,
                               angle_step: float, index_search_range: int, return_costs: bool = True):
    """refine_alignment_hausdorff (align_algorithms.rs:339-451) with the grid scored on the GPU.
    Returns (best_angle, best_cl_ref_idx, min_hausdorff, costs of every evaluated candidate).
    return_costs=False asks for the winner only (what align_combined does): candidates that a lower
    bound rules out are then not evaluated, and the last element is an empty array."""
    pk = _ClPack(geoms)
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    n_ang = int(math.floor(2.0 * angle_search_range / angle_step)) + 3 if angle_step > 0 else 1
    cap = (2 * int(index_search_range) + 1) * n_ang if return_costs else 0
    costs = np.zeros(cap, dtype=np.float64)
    ba, mh, bi, ne = C.c_double(0.0), C.c_double(0.0), C.c_int64(0), C.c_int64(0)
    N.check(N.lib().mm_refine_alignment_hausdorff(engine.handle, pk.ptr, len(geoms), N._ptr(centerline.points),
                                                  len(centerline), int(initial_cl_ref_idx), float(initial_rotation),
                                                  N._ptr(pts), pts.shape[0], float(angle_search_range),
                                                  float(angle_step), int(index_search_range), C.byref(ba),
                                                  C.byref(bi), C.byref(mh), N._ptr(costs) if return_costs else None, cap,
                                                  C.byref(ne)),
            "refine_alignment_hausdorff")
    return ba.value, int(bi.value), mh.value, costs[: min(int(ne.value), cap)].copy()


# ---- the reference's entry points ----------------------------------------------------------
def align_three_point(centerline: Centerline, geometry, main_ref_pt, counterclockwise_ref_pt, clockwise_ref_pt,
                      angle_step_deg: float = 1.0, write: bool = False, watertight: bool = True,
                      interpolation_steps: int = 0, output_dir: str = "output/aligned", contour_types=None,
                      case_name: str = "None", align_wall_anomalous: bool = False):
    """multimodars/_processing.py:1010-1103 -> (geometry, spacing_mm, total_rotation_deg)."""
    geoms, rebuild = _unpack(geometry)
    pk = _ClPack(geoms)
    a, b, d = _v3(main_ref_pt), _v3(counterclockwise_ref_pt), _v3(clockwise_ref_pt)
    sp, rot = C.c_double(0.0), C.c_double(0.0)
    N.check(N.lib().mm_align_three_point(N._ptr(centerline.points), len(centerline), pk.ptr, len(geoms),
                                         _ref_point_index(geoms[0]), N._ptr(a), N._ptr(b), N._ptr(d),
                                         math.radians(angle_step_deg), int(bool(align_wall_anomalous) and not _py_walls()),
                                         C.byref(sp), C.byref(rot)),
            "align_three_point")
    pk.commit_flags()
    _align_walls(geoms, align_wall_anomalous)                                   # align.rs:105-107 (checker only)
    out = rebuild()
    _process_and_write(write, out, case_name, output_dir, interpolation_steps, watertight, contour_types)  # :109-121
    return out, sp.value, rot.value * (180.0 / math.pi)


def align_manual(centerline: Centerline, geometry, rotation_angle_deg: float, ref_point, write: bool = False,
                 watertight: bool = True, interpolation_steps: int = 0, output_dir: str = "output/aligned",
                 contour_types=None, case_name: str = "None", align_wall_anomalous: bool = False):
    """multimodars/_processing.py:1106-1188 -> (geometry, spacing_mm, total_rotation_deg)."""
    geoms, rebuild = _unpack(geometry)
    pk = _ClPack(geoms)
    r = _v3(ref_point)
    sp, rot = C.c_double(0.0), C.c_double(0.0)
    N.check(N.lib().mm_align_manual(N._ptr(centerline.points), len(centerline), pk.ptr, len(geoms),
                                    float(rotation_angle_deg), N._ptr(r), int(bool(align_wall_anomalous) and not _py_walls()),
                                    C.byref(sp), C.byref(rot)),
            "align_manual")
    pk.commit_flags()
    _align_walls(geoms, align_wall_anomalous)                                   # align.rs:147-149 (checker only)
    out = rebuild()
    _process_and_write(write, out, case_name, output_dir, interpolation_steps, watertight, contour_types)  # :151-163
    return out, sp.value, rot.value * (180.0 / math.pi)


def align_combined(centerline: Centerline, geometry, main_ref_pt, counterclockwise_ref_pt, clockwise_ref_pt, points,
                   angle_step_deg: float = 1.0, angle_range_deg: float = 15.0, index_range: int = 2,
                   write: bool = False, watertight: bool = True, interpolation_steps: int = 0,
                   output_dir: str = "output/aligned", contour_types=None, case_name: str = "None",
                   align_wall_anomalous: bool = False, engine: Optional[N.Engine] = None):
    """multimodars/_processing.py:1191-1300 -> (geometry, spacing_mm, total_rotation_deg).  The
    Hausdorff refinement grid ((2*index_range+1) x angles) is scored on the GPU."""
    if engine is None:
        from .api import default_engine
        engine = default_engine()
    geoms, rebuild = _unpack(geometry)
    pk = _ClPack(geoms)
    a, b, d = _v3(main_ref_pt), _v3(counterclockwise_ref_pt), _v3(clockwise_ref_pt)
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    sp, rot, ri, ne = C.c_double(0.0), C.c_double(0.0), C.c_int64(0), C.c_int64(0)
    N.check(N.lib().mm_align_combined(engine.handle, N._ptr(centerline.points), len(centerline), pk.ptr, len(geoms),
                                      _ref_point_index(geoms[0]), N._ptr(a), N._ptr(b), N._ptr(d), N._ptr(pts),
                                      pts.shape[0], math.radians(angle_step_deg), math.radians(angle_range_deg),
                                      int(index_range), int(bool(align_wall_anomalous) and not _py_walls()),
                                      C.byref(sp), C.byref(rot), C.byref(ri), C.byref(ne)),
            "align_combined")
    pk.commit_flags()
    _align_walls(geoms, align_wall_anomalous)                                   # align.rs:266-268 (checker only)
    out = rebuild()
    first = out.geom_a if hasattr(out, "geom_a") else out
    first.meta["refined_cl_ref_idx"] = int(ri.value)
    first.meta["refine_evals"] = int(ne.value)
    _process_and_write(write, out, case_name, output_dir, interpolation_steps, watertight, contour_types)  # align.rs:270-282
    return out, sp.value, rot.value * (180.0 / math.pi)
