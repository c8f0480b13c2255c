"""Shared test helpers."""
from __future__ import annotations

import numpy as np


def to_oracle(orc, g):
    """FlatGeometry -> OracleGeometry (deep copy; same CSR layout)."""
    cp = lambda a: None if a is None else a.copy()
    og = orc.OracleGeometry(cp(g.ids), cp(g.lumen_ids), cp(g.orig_frames), cp(g.centroids), cp(g.lumen_off),
                            cp(g.lumen), cp(g.cath_off), cp(g.cath), cp(g.extra_off), cp(g.extra),
                            cp(g.has_ref), cp(g.ref), g.label)
    # centerline placement extras (orc_clgeom)
    if getattr(g, "lumen_centroids", None) is not None:
        og.lumen_centroids = g.lumen_centroids.copy()
        og.has_lumen_centroid = (g.has_lumen_centroid.copy() if g.has_lumen_centroid is not None
                                 else np.ones(g.n_frames, dtype=np.uint8))
    counts = g.meta.get("extra_counts") if hasattr(g, "meta") else None
    if g.extra_off is not None and counts:
        kinds = [k for k in ("eem", "calcification", "sidebranch", "wall") if k in counts]
        per = np.stack([np.asarray(counts[k], dtype=np.int64) for k in kinds], axis=1)
        ko = np.zeros(per.size + 1, dtype=np.int64)
        ko[1:] = np.cumsum(per.reshape(-1))
        og.n_extra_kinds, og.extra_kind_off = len(kinds), ko
        if "wall" in kinds and int(np.sum(counts["wall"])) > 0:
            og.wall_kind1 = kinds.index("wall") + 1
    for key in ("lumen_aortic", "wall_aortic"):                     # ContourPoint.aortic, per point
        fl = g.meta.get(key) if hasattr(g, "meta") else None
        if fl is not None:
            setattr(og, key, np.ascontiguousarray(fl, dtype=np.uint8).copy())
    return og


def geoms_equal(g, og) -> bool:
    """Bit-for-bit equality of everything the chain mutates.  NaN coordinates (a degenerate wall: 0/0 in the
    reference's own arithmetic, wall.rs:151-168) must sit at the same positions on both sides."""
    eq = lambda a, b: np.array_equal(a, b, equal_nan=True)
    ok = eq(g.lumen, og.lumen) and eq(g.centroids, og.centroids)
    if g.cath is not None:
        ok = ok and eq(g.cath, og.cath)
    if g.extra is not None:
        ok = ok and eq(g.extra, og.extra)
    if g.ref is not None:
        ok = ok and eq(g.ref, og.ref)
    if getattr(g, "lumen_centroids", None) is not None and og.lumen_centroids is not None:
        ok = ok and eq(g.lumen_centroids, og.lumen_centroids)
    return bool(ok)


def to_oracle_cl(ocl, centerline):
    """multimoda_rs_amd.Centerline -> oracle CL array (same 64-byte record layout)."""
    a = np.zeros(len(centerline), dtype=ocl.CL_DTYPE)
    for f in ocl.CL_DTYPE.names:
        a[f] = centerline.points[f]
    return a


def blob(rng, n, radius=2.5, centre=(4.5, 4.5), noise=0.05):
    """A noisy closed contour with n points (roughly IVUS-lumen sized)."""
    t = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = radius * (1 + 0.15 * np.cos(2 * t + rng.uniform(0, 6)) + 0.07 * np.sin(3 * t)) + rng.normal(0, noise, n)
    return np.stack([centre[0] + r * np.cos(t), centre[1] + 0.8 * r * np.sin(t)], axis=1)
