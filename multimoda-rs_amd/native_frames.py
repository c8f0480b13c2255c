"""FlatGeometry <-> the library's frame list (``mm_frames``, include/mm_build.h, csrc/mm_frames.cpp): the post-steps of
``align_frames_in_geometry`` and ``postprocess_geom_pair`` behind the C ABI.  ``postproc.py`` / ``postproc_flat.py``
hold the same logic in Python and stay as the checker (tests/test_native_frames.py compares them bit for bit)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from .geometry import FlatGeometry
from .io import EXTRA_KINDS


class NativeFrames:
    """Owns one ``mm_frames`` handle."""

    def __init__(self, g: FlatGeometry):
        self.label, self.meta = g.label, dict(g.meta)
        F = g.n_frames
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a

        st = N.MMFlatGeometry()
        gs = g.c_struct()
        keep.append(gs)
        st.g = gs
        counts = g.meta.get("extra_counts") or {}
        ec = np.zeros((F, 4), dtype=np.int64)
        for q, k in enumerate(EXTRA_KINDS):
            if k in counts:
                ec[:, q] = np.asarray(counts[k], dtype=np.int64)
        if g.extra_off is None:
            ec[:] = 0
        st.extra_counts = N._ptr(arr(ec, np.int64))
        if g.lumen_centroids is not None:
            st.lumen_centroid = N._ptr(arr(g.lumen_centroids, np.float64))
            if g.has_lumen_centroid is not None:
                st.has_lumen_centroid = N._ptr(arr(g.has_lumen_centroid, np.uint8))
        for name, key in (("aortic", "aortic_thickness"), ("pulmonary", "pulmonary_thickness")):
            vals = g.meta.get(key) or [None] * F
            has = arr([v is not None for v in vals], np.uint8)
            th = arr([0.0 if v is None else float(v) for v in vals], np.float64)
            setattr(st, f"{name}_thickness", N._ptr(th))
            setattr(st, f"has_{name}", N._ptr(has))
        la, wa = g.meta.get("lumen_aortic"), g.meta.get("wall_aortic")
        if la is not None:
            st.lumen_aortic = N._ptr(arr(la, np.uint8))
        if wa is not None:
            st.wall_aortic = N._ptr(arr(wa, np.uint8))
        self._h = C.c_void_p()
        N.check(N.lib().mm_frames_from_flat(C.byref(st), C.byref(self._h)), "mm_frames_from_flat")

    def finish_within(self, ref_idx: int, smooth: bool) -> bool:
        an = C.c_int(0)
        N.check(N.lib().mm_frames_finish_within(self._h, int(ref_idx), int(bool(smooth)), C.byref(an)), "finish_within")
        return bool(an.value)

    def postprocess_pair(self, other: "NativeFrames", tol: float, anomalous: bool) -> None:
        N.check(N.lib().mm_frames_postprocess_pair(self._h, other._h, float(tol), int(bool(anomalous))), "postprocess_pair")

    def to_flat(self) -> FlatGeometry:
        L = N.lib()
        F, nl, nc, ne, nw = C.c_int32(0), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        N.check(L.mm_frames_dims(self._h, C.byref(F), C.byref(nl), C.byref(nc), C.byref(ne), C.byref(nw)), "mm_frames_dims")
        F, nl, nc, ne, nw = F.value, nl.value, nc.value, ne.value, nw.value
        g = FlatGeometry(ids=np.zeros(F, np.uint32), lumen_ids=np.zeros(F, np.uint32), orig_frames=np.zeros(F, np.uint32),
                         centroids=np.zeros((F, 3)), lumen_off=np.zeros(F + 1, np.int64), lumen=np.empty((nl, 3)),
                         has_ref=np.zeros(F, np.uint8), ref=np.zeros((F, 3)), label=self.label)
        if nc:
            g.cath_off, g.cath = np.zeros(F + 1, np.int64), np.empty((nc, 3))    # (the point arrays are written in full)
        if ne:
            g.extra_off, g.extra = np.zeros(F + 1, np.int64), np.empty((ne, 3))
        ec = np.zeros((F, 4), dtype=np.int64)
        has_lc, lc = np.zeros(F, np.uint8), np.zeros((F, 3))
        a_th, p_th, has_a, has_p = np.zeros(F), np.zeros(F), np.zeros(F, np.uint8), np.zeros(F, np.uint8)
        la, wa = np.zeros(max(nl, 1), np.uint8), np.zeros(max(nw, 1), np.uint8)
        st = N.MMFlatGeometry()
        p = N._ptr
        gs = N.MMGeometry()
        gs.id, gs.lumen_id, gs.orig_frame, gs.centroid = p(g.ids), p(g.lumen_ids), p(g.orig_frames), p(g.centroids)
        gs.lumen_off, gs.lumen, gs.cath_off, gs.cath = p(g.lumen_off), p(g.lumen), p(g.cath_off), p(g.cath)
        gs.extra_off, gs.extra, gs.has_ref, gs.ref = p(g.extra_off), p(g.extra), p(g.has_ref), p(g.ref)
        st.g = gs
        st.extra_counts, st.has_lumen_centroid, st.lumen_centroid = p(ec), p(has_lc), p(lc)
        st.aortic_thickness, st.has_aortic, st.pulmonary_thickness, st.has_pulmonary = p(a_th), p(has_a), p(p_th), p(has_p)
        st.lumen_aortic, st.wall_aortic = p(la), p(wa)
        N.check(L.mm_frames_export(self._h, C.byref(st)), "mm_frames_export")
        meta = dict(self.meta)
        meta["extra_counts"] = {k: ec[:, q].copy() for q, k in enumerate(EXTRA_KINDS)}
        meta["aortic_thickness"] = [float(a_th[i]) if has_a[i] else None for i in range(F)]
        meta["pulmonary_thickness"] = [float(p_th[i]) if has_p[i] else None for i in range(F)]
        if nl and la[:nl].any():
            meta["lumen_aortic"] = la[:nl].astype(bool)
        else:
            meta.pop("lumen_aortic", None)
        if nw and wa[:nw].any():
            meta["wall_aortic"] = wa[:nw].astype(bool)
        else:
            meta.pop("wall_aortic", None)
        g.meta = meta
        if F and has_lc.any():
            g.has_lumen_centroid, g.lumen_centroids = has_lc, lc
        return g

    def close(self):
        if self._h.value:
            N.lib().mm_frames_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def finish_within(g: FlatGeometry, ref_idx: int, smooth: bool) -> Tuple[FlatGeometry, bool]:
    """align_within.rs:136-160 on a FlatGeometry; returns (new geometry, anomalous)."""
    nf = NativeFrames(g)
    try:
        an = nf.finish_within(ref_idx, smooth)
        return nf.to_flat(), an
    finally:
        nf.close()


def stage_pair(a: FlatGeometry, b: FlatGeometry) -> Tuple[NativeFrames, NativeFrames]:
    """The library's own copies of two geometries (mm_frames_from_flat): from here on the FlatGeometry objects may change
    without the pair noticing -- a snapshot that costs nothing extra, because the post-processing starts with this copy."""
    fa = NativeFrames(a)
    try:
        return fa, NativeFrames(b)
    except BaseException:
        fa.close()
        raise


def postprocess_staged(fa: NativeFrames, fb: NativeFrames, tol: float, anomalous: bool) -> Tuple[FlatGeometry, FlatGeometry]:
    """postprocess_geom_pair (postprocessing.rs:12-87) on a staged pair; closes the handles."""
    try:
        fa.postprocess_pair(fb, tol, anomalous)
        return fa.to_flat(), fb.to_flat()
    finally:
        fa.close(); fb.close()


def postprocess_pair(a: FlatGeometry, b: FlatGeometry, tol: float, anomalous: bool) -> Tuple[FlatGeometry, FlatGeometry]:
    """postprocess_geom_pair (postprocessing.rs:12-87) on two FlatGeometry objects."""
    return postprocess_staged(*stage_pair(a, b), tol, anomalous)
