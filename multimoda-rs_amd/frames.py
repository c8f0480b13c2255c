"""Frame-list mirror of the reference's ``Geometry`` / ``Frame`` / ``Contour`` value types
(src/types/native/{geometry,frame,contour}.rs) for the host-side bookkeeping around the hot path:
hole filling, wall synthesis, smoothing, post-processing (postproc.py).  The GPU path works on the
flat CSR ``FlatGeometry``; ``to_frames`` / ``from_frames`` convert both ways without touching a
coordinate.

Conventions: ``point_index`` == position in the contour (what ``sort_contour_points`` leaves,
contour.rs:401-404); ``frame_index`` of points is not stored (it is ``Frame.id`` everywhere on this
path, geometry.rs:299-318).  Contour kinds are lower-case strings: lumen, eem, calcification,
sidebranch, catheter, wall.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .geometry import FlatGeometry, contour_centroid
from .io import EXTRA_KINDS

Triple = Tuple[float, float, float]


@dataclass
class Contour:
    """types/native/contour.rs:35-44."""
    id: int
    original_frame: int
    points: np.ndarray                       # (n, 3) f64
    centroid: Optional[Triple] = None
    aortic_thickness: Optional[float] = None
    pulmonary_thickness: Optional[float] = None
    kind: str = "lumen"
    aortic: Optional[np.ndarray] = None      # (n,) bool ContourPoint.aortic; None = all False

    def __post_init__(self):
        self.points = np.ascontiguousarray(np.asarray(self.points, dtype=np.float64).reshape(-1, 3))
        if self.aortic is None:
            self.aortic = np.zeros(self.points.shape[0], dtype=bool)

    def __len__(self) -> int:
        return int(self.points.shape[0])

    def compute_centroid(self) -> None:
        """contour.rs:213-224: sequential sums / n; None for an empty contour."""
        self.centroid = None if len(self) == 0 else contour_centroid(self.points)

    def clone(self) -> "Contour":
        return Contour(self.id, self.original_frame, self.points.copy(), self.centroid, self.aortic_thickness,
                       self.pulmonary_thickness, self.kind, self.aortic.copy())


@dataclass
class Frame:
    """types/native/frame.rs:8-15."""
    id: int
    centroid: List[float]
    lumen: Contour
    extras: Dict[str, Contour] = field(default_factory=dict)
    reference_point: Optional[np.ndarray] = None     # (3,) f64
    reference_aortic: bool = False

    def clone(self) -> "Frame":
        return Frame(self.id, list(self.centroid), self.lumen.clone(), {k: v.clone() for k, v in self.extras.items()},
                     None if self.reference_point is None else self.reference_point.copy(), self.reference_aortic)

    def set_z(self, z: float) -> None:
        """Frame::set_value(None, None, None, Some(z)) (frame.rs:96-116)."""
        self.lumen.points[:, 2] = z
        if self.lumen.centroid is not None:
            self.lumen.centroid = (self.lumen.centroid[0], self.lumen.centroid[1], z)
        for c in self.extras.values():
            c.points[:, 2] = z
            if c.centroid is not None:
                c.centroid = (c.centroid[0], c.centroid[1], z)
        if self.reference_point is not None:
            self.reference_point[2] = z
        self.centroid[2] = z

    def translate(self, dx: float, dy: float, dz: float) -> None:
        """Frame::translate (frame.rs:18-38): contour centroids are recomputed from the points, the
        frame centroid is shifted."""
        d = np.array([dx, dy, dz])
        self.lumen.points += d
        self.lumen.compute_centroid()
        for c in self.extras.values():
            c.points += d
            c.compute_centroid()
        if self.reference_point is not None:
            self.reference_point += d
        self.centroid[0] += dx
        self.centroid[1] += dy
        self.centroid[2] += dz


# --------------------------------------------------------------------------------------
def to_frames(g: FlatGeometry) -> List[Frame]:
    """FlatGeometry -> frame list (copies).  Contour centroids: the lumen's from
    ``g.lumen_centroids`` (None where absent); extras get the mean of their points."""
    counts = g.meta.get("extra_counts") or {}
    a_th = g.meta.get("aortic_thickness") or [None] * g.n_frames
    p_th = g.meta.get("pulmonary_thickness") or [None] * g.n_frames
    lum_aortic = g.meta.get("lumen_aortic")
    wall_aortic = g.meta.get("wall_aortic")
    frames = []
    for i in range(g.n_frames):
        lo, hi = int(g.lumen_off[i]), int(g.lumen_off[i + 1])
        cen = None
        if g.lumen_centroids is not None and (g.has_lumen_centroid is None or g.has_lumen_centroid[i]):
            cen = tuple(float(v) for v in g.lumen_centroids[i])
        lumen = Contour(int(g.lumen_ids[i]), int(g.orig_frames[i]), g.lumen[lo:hi].copy(), cen, a_th[i], p_th[i],
                        "lumen", None if lum_aortic is None else np.asarray(lum_aortic[lo:hi], dtype=bool).copy())
        extras: Dict[str, Contour] = {}
        if g.extra_off is not None:
            e = int(g.extra_off[i])
            for k in EXTRA_KINDS:
                n = int(counts[k][i]) if k in counts else 0
                if n:
                    c = Contour(int(g.ids[i]), int(g.orig_frames[i]), g.extra[e:e + n].copy(), None, None, None, k)
                    if k == "wall":
                        c.aortic_thickness, c.pulmonary_thickness = a_th[i], p_th[i]
                        if wall_aortic is not None:
                            w0 = int(np.sum(counts["wall"][:i]))
                            c.aortic = np.asarray(wall_aortic[w0:w0 + n], dtype=bool).copy()
                    c.compute_centroid()
                    extras[k] = c
                    e += n
        if g.cath_off is not None:
            c = Contour(int(g.ids[i]), int(g.orig_frames[i]), g.cath[int(g.cath_off[i]):int(g.cath_off[i + 1])].copy(),
                        None, None, None, "catheter")
            c.compute_centroid()
            extras["catheter"] = c
        ref = g.ref[i].copy() if g.has_ref is not None and g.has_ref[i] else None
        frames.append(Frame(int(g.ids[i]), [float(v) for v in g.centroids[i]], lumen, extras, ref))
    return frames


def from_frames(frames: List[Frame], label: str = "", meta: Optional[dict] = None) -> FlatGeometry:
    """Frame list -> FlatGeometry.  A catheter is emitted only if every frame has one (the chain's
    rule, align_within.rs:45-59 reads frames[0] and indexes all)."""
    F = len(frames)
    has_cath = F > 0 and all("catheter" in f.extras for f in frames)
    g = FlatGeometry.from_frames(
        [f.lumen.points for f in frames],
        catheters=[f.extras["catheter"].points for f in frames] if has_cath else None,
        centroids=[f.centroid for f in frames] if F else None,
        ids=[f.id for f in frames], orig_frames=[f.lumen.original_frame for f in frames],
        ref_points={i: f.reference_point for i, f in enumerate(frames) if f.reference_point is not None}, label=label)
    g.lumen_ids = np.array([f.lumen.id for f in frames], dtype=np.uint32)
    counts = {k: np.array([len(f.extras[k]) if k in f.extras else 0 for f in frames], dtype=np.int64) for k in EXTRA_KINDS}
    if any(int(c.sum()) for c in counts.values()):
        off = np.zeros(F + 1, dtype=np.int64)
        chunks = []
        for i, f in enumerate(frames):
            tot = 0
            for k in EXTRA_KINDS:
                if k in f.extras and len(f.extras[k]):
                    chunks.append(f.extras[k].points)
                    tot += len(f.extras[k])
            off[i + 1] = off[i] + tot
        g.extra_off = off
        g.extra = np.ascontiguousarray(np.concatenate(chunks, axis=0))
    g.meta = dict(meta or {})
    g.meta["extra_counts"] = counts
    g.meta["aortic_thickness"] = [f.lumen.aortic_thickness for f in frames]
    g.meta["pulmonary_thickness"] = [f.lumen.pulmonary_thickness for f in frames]
    if F and any(f.lumen.aortic.any() for f in frames):
        g.meta["lumen_aortic"] = np.concatenate([f.lumen.aortic for f in frames])
    else:
        g.meta.pop("lumen_aortic", None)
    if F and any("wall" in f.extras and f.extras["wall"].aortic.any() for f in frames):
        g.meta["wall_aortic"] = np.concatenate([f.extras["wall"].aortic for f in frames if "wall" in f.extras])
    else:
        g.meta.pop("wall_aortic", None)
    if F and any(f.lumen.centroid is not None for f in frames):
        g.has_lumen_centroid = np.array([f.lumen.centroid is not None for f in frames], dtype=np.uint8)
        g.lumen_centroids = np.array([f.lumen.centroid if f.lumen.centroid is not None else (0.0, 0.0, 0.0) for f in frames],
                                     dtype=np.float64).reshape(F, 3)
    return g
