"""Host-side bookkeeping around the hot path, restated from the reference:

* hole filling after the chain            src/intravascular/processing/align_within.rs:301-674
* aortic flags, wall synthesis            align_within.rs:316-328, src/intravascular/processing/wall.rs
* 3-frame smoothing of lumen / EEM / wall src/types/native/geometry.rs:165-239
* geometry integrity check                src/intravascular/io/integrity_check.rs:8-256
* pair post-processing (z-resampling, trimming to the common frame range, wall thickness
  equalisation)                           src/intravascular/processing/postprocessing.rs:12-476
* wall twist compensation                 src/intravascular/centerline_align/align.rs:381-595

Everything here is O(points) f64 arithmetic on the frame-list model (frames.py), in the reference's
operation order; none of it enters a search.  The reference's own tests for these functions are
restated in tests/test_postproc.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from multimoda_rs_amd._libm import sincos
from multimoda_rs_amd.frames import Contour, Frame

F64_EPS = 2.220446049250313e-16
EXTRA_ORDER = ("eem", "calcification", "sidebranch", "catheter", "wall")     # postprocessing.rs:243-250


# ======================================================================================
# hole filling (align_within.rs:330-674)
# ======================================================================================
def _median(values: Sequence[float]) -> float:
    v = sorted(values)
    n = len(v)
    if n == 0:
        return 0.0
    return v[n // 2] if n % 2 == 1 else (v[n // 2 - 1] + v[n // 2]) / 2.0


def detect_holes(frames: List[Frame]) -> Tuple[bool, float]:
    """align_within.rs:345-368 -> (has_hole, baseline spacing = median |dz|)."""
    dz = [abs(frames[i].centroid[2] - frames[i - 1].centroid[2]) for i in range(1, len(frames))]
    if not dz:
        return False, 0.0
    baseline = _median(dz)
    if baseline <= F64_EPS:
        return False, baseline
    return any(d >= 1.5 * baseline for d in dz), baseline


def _opt2(a, b, both, one=lambda x: x):
    if a is not None and b is not None:
        return both(a, b)
    if a is not None:
        return one(a)
    if b is not None:
        return one(b)
    return None


def _avg_contour(c1: Contour, c2: Contour, cid: int, original_frame: int) -> Contour:
    """align_within.rs:476-498."""
    n = min(len(c1), len(c2))
    pts = (c1.points[:n] + c2.points[:n]) / 2.0
    cen = _opt2(c1.centroid, c2.centroid, lambda a, b: ((a[0] + b[0]) / 2.0, (a[1] + b[1]) / 2.0, (a[2] + b[2]) / 2.0))
    return Contour(cid, original_frame, pts, cen,
                   _opt2(c1.aortic_thickness, c2.aortic_thickness, lambda x, y: (x + y) / 2.0),
                   _opt2(c1.pulmonary_thickness, c2.pulmonary_thickness, lambda x, y: (x + y) / 2.0),
                   c1.kind, c1.aortic[:n] | c2.aortic[:n])


def _interp_contour(c1: Contour, c2: Contour, t: float, cid: int, original_frame: int) -> Contour:
    """fill_frame_gap (align_within.rs:575-601): p1 + (p2 - p1) * t."""
    n = min(len(c1), len(c2))
    pts = c1.points[:n] + (c2.points[:n] - c1.points[:n]) * t
    cen = _opt2(c1.centroid, c2.centroid,
                lambda a, b: (a[0] + (b[0] - a[0]) * t, a[1] + (b[1] - a[1]) * t, a[2] + (b[2] - a[2]) * t))
    return Contour(cid, original_frame, pts, cen,
                   _opt2(c1.aortic_thickness, c2.aortic_thickness, lambda x, y: x + (y - x) * t),
                   _opt2(c1.pulmonary_thickness, c2.pulmonary_thickness, lambda x, y: x + (y - x) * t),
                   c1.kind, c1.aortic[:n] | c2.aortic[:n])


def _merge_extras(f1: Frame, f2: Frame, both) -> Dict[str, Contour]:
    out: Dict[str, Contour] = {}
    for k in list(f1.extras) + list(f2.extras):
        if k in out:
            continue
        a, b = f1.extras.get(k), f2.extras.get(k)
        out[k] = both(a, b) if a is not None and b is not None else (a if a is not None else b).clone()
    return out


def fix_one_frame_hole(f1: Frame, f2: Frame) -> Frame:
    """align_within.rs:500-543: the averaged frame; it carries no reference point."""
    cen = [(f1.centroid[k] + f2.centroid[k]) / 2.0 for k in range(3)]
    lumen = _avg_contour(f1.lumen, f2.lumen, f2.lumen.id, f2.lumen.original_frame)
    extras = _merge_extras(f1, f2, lambda a, b: _avg_contour(a, b, b.id, b.original_frame))
    return Frame(f2.id, cen, lumen, extras, None)


def create_interpolated_frame(f1: Frame, f2: Frame, t: float) -> Frame:
    """align_within.rs:603-653."""
    cen = [f1.centroid[k] + (f2.centroid[k] - f1.centroid[k]) * t for k in range(3)]
    lumen = _interp_contour(f1.lumen, f2.lumen, t, f2.lumen.id, f2.lumen.original_frame)
    extras = _merge_extras(f1, f2, lambda a, b: _interp_contour(a, b, t, b.id, b.original_frame))
    r1, r2 = f1.reference_point, f2.reference_point
    ref = _opt2(r1, r2, lambda a, b: a + (b - a) * t, lambda a: a.copy())
    return Frame(f2.id, cen, lumen, extras, ref)


def insert_frame(frames: List[Frame], frame: Frame, idx: Optional[int] = None) -> None:
    """Geometry::insert_frame (geometry.rs:285-323): insert, then renumber Frame.id and contour ids."""
    if idx is None:
        z = frame.centroid[2]
        idx = next((i for i, f in enumerate(frames) if f.centroid[2] > z), len(frames))
    frames.insert(idx, frame)
    for i, f in enumerate(frames):
        f.id = i
        f.lumen.id = i
        for c in f.extras.values():
            c.id = i


def fill_holes(frames: List[Frame]) -> List[Frame]:
    """align_within.rs:376-449, in place (and returned).  Gaps of 1.5..2.5 baselines get one averaged
    frame, 2.5..3.5 two interpolated ones, larger ones floor(ratio - 1) interpolated frames."""
    hole, baseline = detect_holes(frames)
    if not hole:
        return frames
    if baseline <= F64_EPS:
        raise RuntimeError("Baseline spacing is zero or too small to decide.")
    i = 1
    while i < len(frames):
        prev, curr = frames[i - 1].clone(), frames[i].clone()
        ratio = abs(curr.centroid[2] - prev.centroid[2]) / baseline
        if ratio < 1.5:
            i += 1
        elif ratio < 2.5:
            insert_frame(frames, fix_one_frame_hole(prev, curr), i)
            i += 2
        elif ratio < 3.5:
            insert_frame(frames, create_interpolated_frame(prev, curr, 1.0 / 3.0), i)
            insert_frame(frames, create_interpolated_frame(prev, curr, 2.0 / 3.0), i + 1)
            i += 3
        else:
            missing = int(max(math.floor(ratio - 1.0), 1.0))
            for k in range(1, missing + 1):
                insert_frame(frames, create_interpolated_frame(prev, curr, k / (missing + 1)), i + k - 1)
            i += missing + 1
    return frames


# ======================================================================================
# aortic flags and walls (align_within.rs:316-328, wall.rs)
# ======================================================================================
def assign_aortic(frames: List[Frame]) -> None:
    """align_within.rs:316-328: the second half of every lumen (index >= len/2) is the aortic side."""
    for f in frames:
        n = len(f.lumen)
        if n:
            f.lumen.aortic = np.arange(n) >= n // 2


def offset_contour(contour: Contour, distance: float, point_range: Optional[Tuple[int, int]] = None) -> Contour:
    """wall.rs:52-100: every point (or those with point_index in the inclusive range) moves
    ``distance`` away from the contour's freshly computed centroid."""
    c = contour.clone()
    c.compute_centroid()
    cx, cy, cz = c.centroid
    pts = c.points.copy()
    n = len(c)
    sel = np.ones(n, dtype=bool) if point_range is None else \
        (np.arange(n) >= point_range[0]) & (np.arange(n) <= point_range[1])
    dx, dy, dz = c.points[:, 0] - cx, c.points[:, 1] - cy, c.points[:, 2] - cz
    ln = np.sqrt(dx * dx + dy * dy + dz * dz)
    ok = sel & (ln > F64_EPS)
    safe = np.where(ok, ln, 1.0)
    pts[:, 0] = np.where(ok, c.points[:, 0] + (dx / safe) * distance, c.points[:, 0])
    pts[:, 1] = np.where(ok, c.points[:, 1] + (dy / safe) * distance, c.points[:, 1])
    pts[:, 2] = np.where(ok, c.points[:, 2] + (dz / safe) * distance, c.points[:, 2])
    return Contour(c.id, c.original_frame, pts, c.centroid, c.aortic_thickness, c.pulmonary_thickness, "wall",
                   c.aortic.copy())


def _rust_round(x: float) -> int:
    """f64::round: half away from zero."""
    return int(math.floor(x + 0.5)) if x >= 0.0 else -int(math.floor(-x + 0.5))


def create_aortic_wall(contour: Contour) -> Contour:
    """wall.rs:109-213: the coronary half (indices 0..n/2) is the lumen offset by 1 mm, the aortic
    half a rectangle of the measured thickness built from three straight segments."""
    n = len(contour)
    first_quarter, half = n // 4, n // 2
    third_quarter = first_quarter * 3
    P = contour.points
    if contour.aortic_thickness is None:
        raise RuntimeError("aortic_thickness must be present for this contour")
    outer_x = P[third_quarter, 0] + contour.aortic_thickness
    z = P[third_quarter, 2]
    up_mid = (P[0, 0], P[0, 1] + 1.0)
    up_right = (outer_x, up_mid[1])
    low_mid = (P[half, 0], P[half, 1] - 1.0)
    low_right = (outer_x, low_mid[1])
    dist_up = abs(up_right[0] - up_mid[0])
    dist_right = abs(up_right[1] - low_right[1])
    dist_low = abs(low_right[0] - low_mid[0])
    total = dist_up + dist_right + dist_low
    n_up = _rust_round(dist_up / total * half)
    n_mid = _rust_round(dist_right / total * half)
    n_low = half - n_up - n_mid
    if n_low < 0:
        raise RuntimeError("attempt to subtract with overflow")        # usize arithmetic panics in the reference
    right: List[Tuple[float, float]] = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(n_low):                                         # low_mid -> low_right
            t = np.float64(i) / np.float64(n_low - 1)
            right.append((float(low_mid[0] + t * (low_right[0] - low_mid[0])), low_mid[1]))
        for i in range(n_mid):                                         # low_right -> up_right
            t = np.float64(i) / np.float64(n_mid - 1)
            right.append((low_right[0], float(low_right[1] + t * (up_right[1] - low_right[1]))))
        for i in range(n_up):                                          # up_right -> up_mid
            t = np.float64(i) / np.float64(max(n_up, 1) - 1)
            right.append((float(up_right[0] - t * (up_right[0] - up_mid[0])), up_right[1]))
    left = offset_contour(contour, 1.0, (0, half))
    left_len = half + 1 if n % 2 else half                             # wall.rs:171-176
    left_len = min(left_len, n)
    if left_len + len(right) > n:
        raise RuntimeError(f"Index out of bounds: {left_len + len(right) - 1} >= {n}")
    pts = np.zeros((left_len + len(right), 3), dtype=np.float64)
    pts[:left_len] = left.points[:left_len]
    aortic = np.zeros(pts.shape[0], dtype=bool)
    aortic[:left_len] = left.aortic[:left_len]
    for i, (x, y) in enumerate(right):
        pts[left_len + i] = (x, y, z)
        aortic[left_len + i] = contour.aortic[left_len + i]
    return Contour(contour.id, contour.original_frame, pts, contour.centroid, contour.aortic_thickness,
                   contour.pulmonary_thickness, "wall", aortic)


def create_wall_frames(frames: List[Frame], anomalous: bool, with_pulmonary: bool = False) -> List[Frame]:
    """wall.rs:7-34: a Wall contour per frame -- from the lumen (anomalous or no EEM) or the EEM,
    offset by 1 mm, or the aortic-wall construction when a thickness was measured."""
    if with_pulmonary:
        raise NotImplementedError("not yet implemented")               # todo!() in the reference
    out = []
    for f in frames:
        src = f.lumen if anomalous or "eem" not in f.extras else f.extras["eem"]
        wall = offset_contour(src, 1.0, None) if src.aortic_thickness is None else create_aortic_wall(src)
        g = f.clone()
        g.extras["wall"] = wall
        out.append(g)
    return out


def offset_contours_batched(P: np.ndarray, distance: float) -> np.ndarray:
    """offset_contour (wall.rs:52-100) for F contours of equal length at once: P is (F, m, 3).  The
    arithmetic per element is that of the per-contour version (sequential centroid sums, the same
    elementwise operations), so the result is bit-identical; only the Python loop over frames is gone."""
    F, m, _ = P.shape
    c = centroids_batched(P)                                                  # compute_centroid per contour
    d = P - c[:, None, :]
    ln = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    ok = ln > F64_EPS
    safe = np.where(ok, ln, 1.0)
    out = P.copy()
    for k in range(3):
        out[..., k] = np.where(ok, P[..., k] + (d[..., k] / safe) * distance, P[..., k])
    return out


def smooth_batched(P: np.ndarray) -> np.ndarray:
    """The lumen / EEM / wall part of smooth_frames for F contours of equal length: (F, m, 3) ->
    (F, m, 3), x and y averaged over (previous, current, next) frame with the ends mirrored."""
    prev = np.concatenate([P[:1], P[:-1]], axis=0)
    nxt = np.concatenate([P[1:], P[-1:]], axis=0)
    out = P.copy()
    out[..., 0] = (prev[..., 0] + P[..., 0] + nxt[..., 0]) / 3.0
    out[..., 1] = (prev[..., 1] + P[..., 1] + nxt[..., 1]) / 3.0
    return out


def centroids_batched(P: np.ndarray) -> np.ndarray:
    """compute_centroid (contour.rs:213-224) of F contours of equal length: sequential sums / m."""
    from multimoda_rs_amd import _native as N
    F, m, _ = P.shape
    return N.contour_centroids(np.ascontiguousarray(P).reshape(F * m, 3), np.arange(F + 1, dtype=np.int64) * m)


# ======================================================================================
# smoothing (geometry.rs:165-239)
# ======================================================================================
def smooth_frames(frames: List[Frame]) -> List[Frame]:
    """3-frame moving average of x, y for the lumen and, where all three frames have them, the EEM
    and wall contours; z, flags and the frame centroid stay, contour centroids are recomputed."""
    n = len(frames)
    out = []
    for i, cur in enumerate(frames):
        prev, nxt = frames[max(i - 1, 0)], frames[min(i + 1, n - 1)]
        m = len(cur.lumen)

        def smooth(c: Contour, p: Contour, q: Contour) -> Contour:
            if len(c) < m or len(p) < m or len(q) < m:
                raise RuntimeError("index out of bounds")              # the reference indexes 0..point_count
            pts = c.points[:m].copy()
            pts[:, 0] = (p.points[:m, 0] + c.points[:m, 0] + q.points[:m, 0]) / 3.0
            pts[:, 1] = (p.points[:m, 1] + c.points[:m, 1] + q.points[:m, 1]) / 3.0
            s = Contour(c.id, c.original_frame, pts, None, c.aortic_thickness, c.pulmonary_thickness, c.kind,
                        c.aortic[:m].copy())
            s.compute_centroid()
            return s

        f = cur.clone()
        f.lumen = smooth(cur.lumen, prev.lumen, nxt.lumen)
        for k in ("eem", "wall"):
            if k in cur.extras and k in prev.extras and k in nxt.extras:
                f.extras[k] = smooth(cur.extras[k], prev.extras[k], nxt.extras[k])
        out.append(f)
    return out


# ======================================================================================
# integrity check (io/integrity_check.rs)
# ======================================================================================
def find_proximal_end_idx(frames: List[Frame]) -> int:
    """geometry.rs:42-60."""
    n = len(frames)
    if n == 0:
        return 0
    if n == 1:
        return frames[0].lumen.id
    return frames[0].lumen.id if frames[0].lumen.original_frame > frames[-1].lumen.original_frame else frames[-1].lumen.id


def find_ref_frame_idx(frames: List[Frame]) -> int:
    """geometry.rs:62-69."""
    for f in frames:
        if f.reference_point is not None:
            return f.id
    raise RuntimeError("No reference point found in any frame")


def check_geometry_integrity(frames: List[Frame]) -> None:
    """integrity_check.rs:8-33; raises RuntimeError with the reference's message.  The
    reference-point original-frame check (:186-198) needs ContourPoint.frame_index, which this model
    does not carry, and is skipped."""
    if not frames:
        raise RuntimeError("Geometry has no frames")
    for i, f in enumerate(frames):                                     # :35-46
        if f.id != i:
            raise RuntimeError(f"Frame IDs are not consecutive. Expected ID {i}, found ID {f.id}")
    for i, f in enumerate(frames):                                     # :49-81
        lc = f.lumen.centroid
        if lc is None:
            n = len(f.lumen)
            lc = (0.0, 0.0, 0.0) if n == 0 else tuple(float(f.lumen.points[:, k].sum() / n) for k in range(3))
        if not all(abs(f.centroid[k] - lc[k]) < 1e-6 for k in range(3)):
            raise RuntimeError(f"Frame centroid does not match lumen centroid in frame {i} (ID {f.id}). "
                               f"Frame: {tuple(f.centroid)}, Lumen: {lc}")
    for i, f in enumerate(frames):                                     # :84-104
        if len(f.lumen) == 0:
            raise RuntimeError(f"Lumen contour has no points in frame {i} (ID {f.id})")
        if f.lumen.kind != "lumen":
            raise RuntimeError(f"Lumen contour has incorrect type in frame {i} (ID {f.id}). Expected Lumen, found {f.lumen.kind}")
    n_ref = sum(1 for f in frames if f.reference_point is not None)    # :107-118
    if n_ref != 1:
        raise RuntimeError(f"Expected exactly one reference point, found {n_ref}")
    expected: Dict[str, int] = {}                                      # :121-166
    for i, f in enumerate(frames):
        for kind, c in [("lumen", f.lumen)] + [(c.kind, c) for c in f.extras.values()]:
            if kind in expected and len(c) != expected[kind]:
                name = "Lumen" if kind == "lumen" else kind.capitalize() + " contour"
                raise RuntimeError(f"{name} point count mismatch in frame {i} (ID {f.id}). "
                                   f"Expected {expected[kind]}, found {len(c)}")
            expected.setdefault(kind, len(c))
    for i, f in enumerate(frames):                                     # :169-200
        for kind, c in f.extras.items():
            if c.original_frame != f.lumen.original_frame:
                raise RuntimeError(f"Original frame mismatch in frame {i} (ID {f.id}). Lumen has original_frame "
                                   f"{f.lumen.original_frame}, {kind} has original_frame {c.original_frame}")
    prox = find_proximal_end_idx(frames)                               # :203-221
    zs = [f.centroid[2] for f in frames]
    min_idx = min(range(len(zs)), key=lambda k: (zs[k], k))
    if prox != min_idx:
        raise RuntimeError(f"Proximal end index is {prox}, but frame with minimum z is {min_idx} (z={zs[min_idx]}).")
    if zs[0] > zs[-1]:                                                 # :224-234
        raise RuntimeError(f"First frame has higher z-coords {zs[0]} than last frame {zs[-1]}")


# ======================================================================================
# pair post-processing (postprocessing.rs)
# ======================================================================================
def get_avg_z_diff(frames: List[Frame]) -> float:
    """postprocessing.rs:100-113."""
    if len(frames) < 2:
        return 0.0
    s = 0.0
    for i in range(1, len(frames)):
        s += frames[i].centroid[2] - frames[i - 1].centroid[2]
    return s / (len(frames) - 1)


def check_same_sample_rate(frames_a: List[Frame], frames_b: List[Frame], tol: float) -> Tuple[bool, float, float]:
    """postprocessing.rs:89-98 (signed difference, as in the reference)."""
    da, db = get_avg_z_diff(frames_a), get_avg_z_diff(frames_b)
    return (da - db) < tol, da, db


def resample_by_diff(frames: List[Frame], diff: float) -> List[Frame]:
    """postprocessing.rs:116-140: smallest z first, then z_i = z_0 + i * diff."""
    fr = [f.clone() for f in frames]
    if fr:
        zs = [f.centroid[2] for f in fr]
        k = min(range(len(zs)), key=lambda i: (zs[i], i))              # min_by: first minimum
        if k:
            fr = fr[k:] + fr[:k]
    start = fr[0].centroid[2]
    for i in range(1, len(fr)):
        fr[i].set_z(start + i * diff)
    return fr


def predict_z_positions(ref_z: float, start_z: float, stop_z: float, z_diff: float) -> List[float]:
    """postprocessing.rs:142-195."""
    out: List[float] = []
    if not math.isfinite(z_diff) or z_diff == 0.0:
        return out
    eps = 1e-9
    if abs(ref_z - start_z) > eps and abs(ref_z - stop_z) > eps:
        cur = ref_z
        while cur >= start_z - eps:
            out.append(cur)
            cur -= z_diff
            if not math.isfinite(cur):
                break
        out.sort()
        cur = ref_z + z_diff
        while cur <= stop_z + eps:
            out.append(cur)
            cur += z_diff
            if not math.isfinite(cur):
                break
    else:
        cur = start_z
        if stop_z >= start_z and z_diff > 0.0:
            while cur <= stop_z + eps:
                out.append(cur)
                cur += z_diff
                if not math.isfinite(cur):
                    break
        elif stop_z <= start_z and z_diff < 0.0:
            while cur >= stop_z - eps:
                out.append(cur)
                cur += z_diff
                if not math.isfinite(cur):
                    break
    return out


def blend_contour(c1: Contour, c2: Contour, t: float) -> Contour:
    """postprocessing.rs:302-340: x, y = p1 + t (p2 - p1); z, flags, ids from c1; thickness and
    centroid interpolated only when both sides have them."""
    n = min(len(c1), len(c2))
    pts = c1.points[:n].copy()
    pts[:, 0] = c1.points[:n, 0] + t * (c2.points[:n, 0] - c1.points[:n, 0])
    pts[:, 1] = c1.points[:n, 1] + t * (c2.points[:n, 1] - c1.points[:n, 1])
    both = lambda a, b, fn: fn(a, b) if a is not None and b is not None else None
    cen = both(c1.centroid, c2.centroid,
               lambda a, b: (a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2])))
    return Contour(c1.id, c1.original_frame, pts, cen,
                   both(c1.aortic_thickness, c2.aortic_thickness, lambda a, b: a + t * (b - a)),
                   both(c1.pulmonary_thickness, c2.pulmonary_thickness, lambda a, b: a + t * (b - a)),
                   c1.kind, c1.aortic[:n].copy())


def new_frames_by_sample_rate(frames: List[Frame], z_coords: Sequence[float]) -> List[Frame]:
    """postprocessing.rs:197-300: frames at the given z positions, existing ones reused (|dz| < 1e-9),
    the others blended from their two neighbours; ids renumbered, z written through."""
    zc = sorted(z_coords)
    max_z = frames[-1].centroid[2]
    out: List[Frame] = []
    for z in zc:
        if z > max_z:
            break
        hit = next((f for f in frames if abs(f.centroid[2] - z) < 1e-9), None)
        if hit is not None:
            out.append(hit.clone())
            continue
        pair = next(((a, b) for a, b in zip(frames, frames[1:]) if a.centroid[2] <= z <= b.centroid[2]), None)
        if pair is None:
            raise RuntimeError("Cannot find frames to interpolate between")
        lo, up = pair
        t = (z - lo.centroid[2]) / (up.centroid[2] - lo.centroid[2])
        extras = {k: blend_contour(lo.extras[k], up.extras[k], t) for k in EXTRA_ORDER
                  if k in lo.extras and k in up.extras}
        out.append(Frame(lo.id, [lo.centroid[0] + t * (up.centroid[0] - lo.centroid[0]),
                                 lo.centroid[1] + t * (up.centroid[1] - lo.centroid[1]), z],
                         blend_contour(lo.lumen, up.lumen, t), extras, None))
    out.sort(key=lambda f: f.centroid[2])                              # stable, like sort_by
    for i, f in enumerate(out):
        f.id = i
        f.lumen.id = i
        z = f.centroid[2]
        f.lumen.points[:, 2] = z
        if f.lumen.centroid is not None:
            f.lumen.centroid = (f.lumen.centroid[0], f.lumen.centroid[1], z)
        for c in f.extras.values():
            c.id = i
            c.points[:, 2] = z
        if f.reference_point is not None:
            f.reference_point[2] = z
    return out


def trim_pair(frames_a: List[Frame], frames_b: List[Frame]) -> Tuple[List[Frame], List[Frame]]:
    """trim_geom_pair (postprocessing.rs:342-409): the same number of frames before and after the
    reference frame in both geometries; ids renumbered."""
    def ref_or_zero(fr):
        try:
            return find_ref_frame_idx(fr)
        except RuntimeError:
            return 0

    ra, rb = ref_or_zero(frames_a), ref_or_zero(frames_b)
    before = min(ra, rb)
    after = min(len(frames_a) - ra, len(frames_b) - rb)

    def cut(fr, r):
        s, e = r - before, r + after
        sel = fr[s:e] if s < e <= len(fr) else fr
        out = [f.clone() for f in sel]
        for i, f in enumerate(out):
            f.id = i
            f.lumen.id = i
            for c in f.extras.values():
                c.id = i
        return out

    return cut(frames_a, ra), cut(frames_b, rb)


def adjust_walls_anomalous_pair(frames_a: List[Frame], frames_b: List[Frame]) -> Tuple[List[Frame], List[Frame]]:
    """postprocessing.rs:411-476: frame by frame the two lumens get the mean of their aortic
    thicknesses (or the one that exists), then the walls are rebuilt."""
    oa, ob = [], []
    for fa, fb in zip(frames_a, frames_b):
        fa2, fb2 = fa.clone(), fb.clone()
        ta, tb = fa.lumen.aortic_thickness, fb.lumen.aortic_thickness
        if ta is not None or tb is not None:
            th = (ta + tb) / 2.0 if ta is not None and tb is not None else (ta if ta is not None else tb)
            fa2.lumen.aortic_thickness = th
            fb2.lumen.aortic_thickness = th
        oa.append(fa2)
        ob.append(fb2)
    return create_wall_frames(oa, True, False), create_wall_frames(ob, True, False)


def postprocess_pair(frames_a: List[Frame], frames_b: List[Frame], tol: float, anomalous: bool
                     ) -> Tuple[List[Frame], List[Frame]]:
    """postprocess_geom_pair (postprocessing.rs:12-87)."""
    same, da, db = check_same_sample_rate(frames_a, frames_b, tol)
    ia, ib = find_ref_frame_idx(frames_a), find_ref_frame_idx(frames_b)
    ref_z_a, ref_z_b = frames_a[ia].centroid[2], frames_b[ib].centroid[2]

    def span(fr):
        z0, zn = fr[0].centroid[2], fr[-1].centroid[2]
        return (z0, zn) if z0 < zn else (zn, z0)

    if same:
        mean = (da + db) / 2.0
        ra, rb = resample_by_diff(frames_a, mean), resample_by_diff(frames_b, mean)
    elif da < db:
        start, stop = span(frames_b)
        rb = new_frames_by_sample_rate(frames_b, predict_z_positions(ref_z_b, start, stop, da))
        ra = resample_by_diff(frames_a, da)
    else:
        start, stop = span(frames_a)
        ra = new_frames_by_sample_rate(frames_a, predict_z_positions(ref_z_a, start, stop, db))
        rb = resample_by_diff(frames_b, db)
    # :70-76 -- the reference indexes the ORIGINAL pair with the resampled geometries' reference indices
    ja, jb = find_ref_frame_idx(ra), find_ref_frame_idx(rb)
    translation = frames_a[ja].centroid[2] - frames_b[jb].centroid[2]
    for f in ra:
        f.translate(0.0, 0.0, translation)
    ta, tb = trim_pair(ra, rb)
    if anomalous:
        ta, tb = adjust_walls_anomalous_pair(ta, tb)
    return ta, tb


# ======================================================================================
# wall twist compensation (centerline_align/align.rs:381-595)
# ======================================================================================
def _v_norm(v):
    return math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


def _v_dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _v_cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _v_angle(a, b):
    n1, n2 = _v_norm(a), _v_norm(b)
    if n1 == 0.0 or n2 == 0.0:
        return 0.0
    return math.acos(max(-1.0, min(1.0, _v_dot(a, b) / (n1 * n2))))


def _axis_angle(axis, angle):
    """Rotation3::from_axis_angle(&Unit::new_normalize(axis), angle), row-major 3x3."""
    if angle == 0.0:
        return ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))
    n = _v_norm(axis)
    ux, uy, uz = axis[0] / n, axis[1] / n, axis[2] / n
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    s, c = sincos(angle)
    omc = 1.0 - c
    return ((sqx + (1.0 - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s),
            (ux * uy * omc + uz * s, sqy + (1.0 - sqy) * c, uy * uz * omc - ux * s),
            (ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (1.0 - sqz) * c))


def _mat_vec(r, v):
    """Rotation3 * Vector3 (nalgebra gemv: column 0, then += column 1, += column 2)."""
    out = []
    for i in range(3):
        y = r[i][0] * v[0]
        y = r[i][1] * v[1] + y
        y = r[i][2] * v[2] + y
        out.append(y)
    return tuple(out)


def _lumen_normal(f: Frame):
    """align.rs:440-461: Newell normal of the lumen about the FRAME centroid."""
    c, p = f.centroid, f.lumen.points
    n = len(f.lumen)
    if n < 3:
        return (0.0, 0.0, 1.0)
    nx = ny = nz = 0.0
    for i in range(n):
        cur, nxt = p[i], p[(i + 1) % n]
        nx += (cur[1] - c[1]) * (nxt[2] - c[2]) - (cur[2] - c[2]) * (nxt[1] - c[1])
        ny += (cur[2] - c[2]) * (nxt[0] - c[0]) - (cur[0] - c[0]) * (nxt[2] - c[2])
        nz += (cur[0] - c[0]) * (nxt[1] - c[1]) - (cur[1] - c[1]) * (nxt[0] - c[0])
    nn = _v_norm((nx, ny, nz))
    return (nx / nn, ny / nn, nz / nn) if nn > 1e-12 else (0.0, 0.0, 1.0)


def _aortic_direction(wall: Contour, frame_centroid):
    """align.rs:385-407."""
    idx = np.nonzero(wall.aortic)[0]
    if idx.size == 0:
        return None
    n = float(idx.size)
    cx = float(sum(wall.points[i, 0] for i in idx)) / n
    cy = float(sum(wall.points[i, 1] for i in idx)) / n
    cz = float(sum(wall.points[i, 2] for i in idx)) / n
    d = (cx - frame_centroid[0], cy - frame_centroid[1], cz - frame_centroid[2])
    return None if _v_norm(d) < 1e-9 else d


def _major_axis(wall: Contour):
    """align.rs:410-437: direction between the farthest pair of points (first maximum)."""
    p = wall.points
    n = len(wall)
    if n < 2:
        return None
    best, fa, fb = 0.0, 0, 0
    for i in range(n):
        d = p[i + 1:] - p[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        if d2.size:
            j = int(np.argmax(d2))                                     # first of equal maxima, like `>`
            if d2[j] > best:
                best, fa, fb = float(d2[j]), i, i + 1 + j
    d = (p[fb, 0] - p[fa, 0], p[fb, 1] - p[fa, 1], p[fb, 2] - p[fa, 2])
    return None if _v_norm(d) < 1e-9 else tuple(float(v) for v in d)


def _project_normalized(v, t):
    """align.rs:465-472."""
    k = _v_dot(v, t)
    p = (v[0] - t[0] * k, v[1] - t[1] * k, v[2] - t[2] * k)
    n = _v_norm(p)
    return None if n < 1e-9 else (p[0] / n, p[1] / n, p[2] / n)


def _parallel_transport(v, t_from, t_to):
    """align.rs:476-492."""
    ang = _v_angle(t_from, t_to)
    if ang < 1e-9:
        return v
    axis = _v_cross(t_from, t_to)
    if _v_norm(axis) < 1e-9:
        if abs(t_from[0]) < 0.9:
            perp = (1.0 - t_from[0] * t_from[0], 0.0 - t_from[1] * t_from[0], 0.0 - t_from[2] * t_from[0])
        else:
            perp = (0.0 - t_from[0] * t_from[1], 1.0 - t_from[1] * t_from[1], 0.0 - t_from[2] * t_from[1])
        n = _v_norm(perp)
        perp = (perp[0] / n, perp[1] / n, perp[2] / n)
        return _mat_vec(_axis_angle(perp, math.pi), v)
    return _mat_vec(_axis_angle(axis, ang), v)


def _signed_angle(frm, to, axis):
    """align.rs:495-497."""
    return math.atan2(_v_dot(_v_cross(frm, to), axis), _v_dot(frm, to))


def align_walls(frames: List[Frame], anomalous: bool = True) -> List[Frame]:
    """align_walls_on_geometry (align.rs:507-584), in place: every frame's Wall contour is rotated about
    the lumen normal so that its aortic side (or major axis) follows the direction of frame 0
    parallel-transported along the vessel.  Lumen and other contours stay."""
    if not anomalous or len(frames) < 2:
        return frames
    f0 = frames[0]
    t0 = _lumen_normal(f0)
    w0 = f0.extras.get("wall")
    if w0 is None:
        return frames
    d0 = _aortic_direction(w0, f0.centroid) or _major_axis(w0)
    u = _project_normalized(d0, t0) if d0 is not None else None
    if u is None:
        return frames
    for i in range(1, len(frames)):
        t_prev, t_cur = _lumen_normal(frames[i - 1]), _lumen_normal(frames[i])
        u = _parallel_transport(u, t_prev, t_cur)
        pu = _project_normalized(u, t_cur)
        if pu is None:
            continue
        u = pu
        f = frames[i]
        w = f.extras.get("wall")
        if w is None:
            continue
        d = _aortic_direction(w, f.centroid)
        has_aortic = d is not None
        if d is None:
            d = _major_axis(w)
            if d is None:
                continue
        v = _project_normalized(d, t_cur)
        if v is None:
            continue
        if has_aortic:
            ang = _signed_angle(v, u, t_cur)
        else:
            a1 = _signed_angle(v, u, t_cur)
            a2 = _signed_angle((-v[0], -v[1], -v[2]), u, t_cur)
            ang = a1 if abs(a1) <= abs(a2) else a2
        if abs(ang) < 1e-6:
            continue
        r = _axis_angle(t_cur, ang)
        c = f.centroid
        for k in range(len(w)):
            q = w.points[k]
            rot = _mat_vec(r, (q[0] - c[0], q[1] - c[1], q[2] - c[2]))
            w.points[k] = (c[0] + rot[0], c[1] + rot[1], c[2] + rot[2])
    return frames
