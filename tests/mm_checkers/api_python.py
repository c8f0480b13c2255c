"""The Python side of multimoda_rs_amd.api's post-steps (checker of mm_frames_finish_within / mm_frames_postprocess_pair)."""
from __future__ import annotations

import numpy as np

from multimoda_rs_amd import api as _api
from multimoda_rs_amd import frames as FR
from multimoda_rs_amd import geometry as G
from multimoda_rs_amd.api import (GeometryPair, TOLERANCE, _angle_ref_point_to_right, _detect_holes, _elliptic_ratio, _replace,
                                  _rotate_geometry)
from multimoda_rs_amd.centerline import with_lumen_centroids

from . import postproc as PP
from .postproc_flat import postprocess_pair_regular


def finish_within_batched(g: G.FlatGeometry, anomalous: bool, smooth: bool) -> bool:
    """The regular case of align_within.rs:144-158 without a Python loop over frames: every lumen (and
    EEM, if present) has the same number of points and no other extras kind is present.  Aortic flags,
    wall contours and smoothing are computed on (F, m, 3) arrays with the per-element arithmetic of
    postproc.assign_aortic / create_wall_frames / smooth_frames (frames with a measured thickness still
    build their aortic wall one by one); the results are bit-identical (tests/test_postproc.py).
    Returns False, leaving g untouched, when the geometry is not regular."""
    F = g.n_frames
    cnt = np.diff(g.lumen_off)
    if F == 0 or not np.all(cnt == cnt[0]) or cnt[0] == 0:
        return False
    m = int(cnt[0])
    counts = g.meta.get("extra_counts") or {}
    has_eem = "eem" in counts and int(np.sum(counts["eem"])) > 0
    if any(int(np.sum(c)) for k, c in counts.items() if k != "eem"):
        return False
    if has_eem and not np.all(counts["eem"] == m):
        return False
    if (g.extra_off is not None) != has_eem:
        return False
    L = g.lumen.reshape(F, m, 3)
    E = g.extra.reshape(F, m, 3) if has_eem else None
    a_th = g.meta.get("aortic_thickness") or [None] * F
    p_th = g.meta.get("pulmonary_thickness") or [None] * F
    aortic = np.zeros((F, m), dtype=bool)
    if anomalous:
        aortic[:, m // 2:] = True                                              # assign_aortic
    src = L if anomalous or not has_eem else E                                 # wall.rs:13-19
    src_aortic = aortic if src is L else np.zeros((F, m), dtype=bool)
    W = PP.offset_contours_batched(src, 1.0)                                   # offset_contour(.., 1.0, None)
    w_aortic = src_aortic.copy()
    # frames whose source contour carries a measured aortic thickness get the aortic-wall construction;
    # only the lumen carries thicknesses (contour.rs:128-141), the EEM never does
    if src is L:
        for i in range(F):
            if a_th[i] is not None:
                c = FR.Contour(int(g.lumen_ids[i]), int(g.orig_frames[i]), L[i].copy(), None, a_th[i], p_th[i], "lumen",
                               aortic[i].copy())
                w = PP.create_aortic_wall(c)
                if len(w) != m:
                    return False
                W[i] = w.points
                w_aortic[i] = w.aortic
    if smooth:
        L = PP.smooth_batched(L)
        W = PP.smooth_batched(W)
        if has_eem:
            E = PP.smooth_batched(E)
    g.lumen = np.ascontiguousarray(L.reshape(F * m, 3))
    blob = np.concatenate([E, W], axis=1) if has_eem else W
    g.extra = np.ascontiguousarray(blob.reshape(-1, 3))
    g.extra_off = np.arange(F + 1, dtype=np.int64) * blob.shape[1]
    meta = dict(g.meta)
    from multimoda_rs_amd.io import EXTRA_KINDS
    meta["extra_counts"] = {k: (np.full(F, m, dtype=np.int64) if k == "wall" or (k == "eem" and has_eem)
                                else np.zeros(F, dtype=np.int64)) for k in EXTRA_KINDS}
    if aortic.any():
        meta["lumen_aortic"] = aortic.reshape(-1)
    else:
        meta.pop("lumen_aortic", None)
    if w_aortic.any():
        meta["wall_aortic"] = w_aortic.reshape(-1)
    else:
        meta.pop("wall_aortic", None)
    g.meta = meta
    if smooth or g.lumen_centroids is None:              # smooth_frames recomputes the contour centroid (geometry.rs:204)
        g.has_lumen_centroid = np.ones(F, dtype=np.uint8)
        g.lumen_centroids = np.ascontiguousarray(PP.centroids_batched(L))
    return True


def finish_within_python(g: G.FlatGeometry, ref_idx: int, smooth: bool) -> bool:
    """align_within.rs:136-160 after the chain, in Python (the native path: api._finish_within)."""
    tracked = g.lumen_centroids is not None and (g.has_lumen_centroid is None or bool(np.all(g.has_lumen_centroid)))
    if not tracked:
        with_lumen_centroids(g)
    hole, _ = _detect_holes(g)
    if hole:                                                                   # :136
        fr = PP.fill_holes(FR.to_frames(g))
        _replace(g, FR.from_frames(fr, g.label, g.meta))
    if ref_idx >= g.n_frames:
        raise RuntimeError("reference frame index out of range")
    lum = g.frame_lumen(ref_idx)
    a_th = g.meta.get("aortic_thickness")
    p_th = g.meta.get("pulmonary_thickness")
    anomalous = (_elliptic_ratio(lum) > 2.0 or (a_th is not None and a_th[ref_idx] is not None)
                 or (p_th is not None and p_th[ref_idx] is not None))        # align_within.rs:249-254
    _rotate_geometry(g, _angle_ref_point_to_right(g, ref_idx, anomalous))      # :139-142 (contour centroids untouched)
    if not finish_within_batched(g, anomalous, smooth):
        fr = FR.to_frames(g)
        if anomalous:
            PP.assign_aortic(fr)                                               # :144-148
        fr = PP.create_wall_frames(fr, anomalous, False)                       # :150-154
        if smooth:
            fr = PP.smooth_frames(fr)                                          # :156-158
        meta = dict(g.meta)
        _replace(g, FR.from_frames(fr, g.label, meta))
    g.meta["anomalous"] = bool(anomalous)
    g.meta["lumen_centroid_fresh"] = bool(smooth)        # geometry.rs:204
    g.meta["lumen_centroid_tracked"] = bool(tracked)
    return bool(anomalous)


def maybe_postprocess_python(pair: GeometryPair, anomalous: bool) -> GeometryPair:
    """postprocess_geom_pair (postprocessing.rs:12-87) in Python (the native path: api._maybe_postprocess)."""
    try:
        fast = postprocess_pair_regular(pair.geom_a, pair.geom_b, TOLERANCE, anomalous)
        if fast is not None:
            return GeometryPair(fast[0], fast[1], pair.label)
        fa, fb = PP.postprocess_pair(FR.to_frames(pair.geom_a), FR.to_frames(pair.geom_b), TOLERANCE, anomalous)
    except RuntimeError as e:
        raise RuntimeError(f"Failed postprocessing of {pair.label}: {e}") from e
    a = FR.from_frames(fa, pair.geom_a.label, pair.geom_a.meta)
    b = FR.from_frames(fb, pair.geom_b.label, pair.geom_b.meta)
    return GeometryPair(a, b, pair.label)
