"""numpy converters of the reference's Python layer (multimodars/_converters.py): ``to_array`` (:19-92,
95-201) and ``numpy_to_geometry`` (:440-602), for this package's containers.  Rows are
``[frame_index, x, y, z]``; for geometries the frame index is ``Frame.id``."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .frames import Contour, Frame, from_frames, to_frames
from .geometry import FlatGeometry
from .io import InputData

_LAYERS = ("lumen", "eem", "calcification", "sidebranch", "catheter", "wall", "reference")


def _geometry_to_numpy(g: FlatGeometry) -> Dict[str, np.ndarray]:
    """_converters.py:124-148."""
    out = {k: [] for k in _LAYERS}
    for f in to_frames(g):
        fid = float(f.id)
        out["lumen"].append(np.column_stack([np.full(len(f.lumen), fid), f.lumen.points]))
        for k, c in f.extras.items():
            if len(c):
                out[k].append(np.column_stack([np.full(len(c), fid), c.points]))
        if f.reference_point is not None:
            out["reference"].append(np.array([[fid, *f.reference_point]]))
    return {k: (np.vstack(v).astype(float) if v else np.zeros((0, 4), dtype=float)) for k, v in out.items()}


def to_array(generic):
    """_converters.py:19-92: Centerline -> (N, 4); FlatGeometry -> dict of (M, 4) arrays per layer;
    GeometryPair -> (dict, dict); InputData -> dict of arrays and metadata."""
    from .api import GeometryPair
    from .centerline import Centerline
    if isinstance(generic, Centerline):
        xyz = generic.xyz()
        return np.column_stack([np.arange(len(generic), dtype=float), xyz])
    if isinstance(generic, FlatGeometry):
        return _geometry_to_numpy(generic)
    if isinstance(generic, GeometryPair):
        return _geometry_to_numpy(generic.geom_a), _geometry_to_numpy(generic.geom_b)
    if isinstance(generic, InputData):
        z = np.zeros((0, 4), dtype=float)
        return {"lumen": generic.lumen, "eem": generic.eem if generic.eem is not None else z,
                "calcification": generic.calcification if generic.calcification is not None else z,
                "sidebranch": generic.sidebranch if generic.sidebranch is not None else z,
                "reference": np.asarray(generic.ref_point, dtype=float).reshape(1, 4), "diastole": generic.diastole,
                "label": generic.label}
    raise TypeError(f"Unsupported type for to_array: {type(generic)}")


def numpy_to_geometry(lumen_arr, eem_arr=None, catheter_arr=None, wall_arr=None, reference_arr=None,
                      label: str = "") -> FlatGeometry:
    """_converters.py:440-602: frames grouped by frame index (ascending), contour centroid = np.mean of the
    points, Frame.centroid = the lumen centroid; the first reference row becomes the reference point of
    EVERY frame (the reference does that, :571-572, 592)."""
    def num(a):
        if a is None:
            return np.zeros((0, 4), dtype=float)
        a = np.asarray(a)
        if a.ndim == 1 and a.dtype.names:
            a = np.vstack([a[n] for n in a.dtype.names]).T
        return np.asarray(a, dtype=float)

    lum, eem, cath, wall, ref = num(lumen_arr), num(eem_arr), num(catheter_arr), num(wall_arr), num(reference_arr)
    if lum.size == 0:
        raise ValueError("lumen_arr cannot be empty")
    gref = None
    if ref.size > 0:
        row = ref[:4] if ref.ndim == 1 else ref[0, :4]
        gref = np.array([float(row[1]), float(row[2]), float(row[3])])
    frames_ids = set()
    for a in (lum, eem, cath, wall):
        if a.size:
            frames_ids.update(a[:, 0].astype(int).tolist())

    def contour(a, fid, kind) -> Optional[Contour]:
        if a.size == 0:
            return None
        pts = a[a[:, 0].astype(int) == fid]
        if len(pts) == 0:
            return None
        cen = (float(np.mean(pts[:, 1])), float(np.mean(pts[:, 2])), float(np.mean(pts[:, 3])))
        return Contour(fid, fid, pts[:, 1:4].copy(), cen, None, None, kind)

    frames = []
    for fid in sorted(frames_ids):
        lc = contour(lum, fid, "lumen")
        if lc is None:
            continue
        extras = {}
        for kind, a in (("eem", eem), ("catheter", cath), ("wall", wall)):
            c = contour(a, fid, kind)
            if c is not None:
                extras[kind] = c
        frames.append(Frame(fid, list(lc.centroid), lc, extras, None if gref is None else gref.copy()))
    return from_frames(frames, label, {})
