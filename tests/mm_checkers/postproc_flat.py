"""postprocess_geom_pair (src/intravascular/processing/postprocessing.rs:12-87) for the regular case,
without a Python loop over frames: both geometries have the same number of points in every lumen /
catheter / extras contour and the same z sampling rate (the ``same_sample_rate`` branch, :24-33 -- the
usual case for a diastolic / systolic pair of one pullback).  Works on (F, m, 3) views of the flat arrays
with the per-element arithmetic of postproc.postprocess_pair; results are bit-identical to the
frame-list version (tests/test_postproc.py), which remains the general path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from . import postproc as PP
from multimoda_rs_amd.frames import Contour
from multimoda_rs_amd.geometry import FlatGeometry
from multimoda_rs_amd.io import EXTRA_KINDS


class _Reg:
    """Per-frame views of a regular FlatGeometry."""

    def __init__(self, g: FlatGeometry):
        F = g.n_frames
        self.F = F
        self.label, self.meta = g.label, dict(g.meta)
        self.ids, self.lumen_ids, self.orig = g.ids.copy(), g.lumen_ids.copy(), g.orig_frames.copy()
        self.cen = g.centroids.copy()
        self.m = int(g.lumen_off[1] - g.lumen_off[0])
        self.L = g.lumen.reshape(F, self.m, 3).copy()
        self.C = None if g.cath_off is None else g.cath.reshape(F, -1, 3).copy()
        counts = g.meta.get("extra_counts") or {}
        self.kinds = [k for k in EXTRA_KINDS if k in counts and int(np.sum(counts[k])) > 0]
        self.kcount = {k: int(counts[k][0]) for k in self.kinds}
        self.X = None if g.extra_off is None else g.extra.reshape(F, -1, 3).copy()
        self.has_ref = g.has_ref.copy() if g.has_ref is not None else np.zeros(F, dtype=np.uint8)
        self.ref = g.ref.copy() if g.ref is not None else np.zeros((F, 3))
        self.lc = None if g.lumen_centroids is None else g.lumen_centroids.copy()
        self.a_th = list(g.meta.get("aortic_thickness") or [None] * F)
        self.p_th = list(g.meta.get("pulmonary_thickness") or [None] * F)
        la = g.meta.get("lumen_aortic")
        self.l_aortic = np.zeros((F, self.m), dtype=bool) if la is None else np.asarray(la, dtype=bool).reshape(F, self.m).copy()
        wa = g.meta.get("wall_aortic")
        mw = self.kcount.get("wall", 0)
        self.w_aortic = (np.zeros((F, mw), dtype=bool) if wa is None or mw == 0
                         else np.asarray(wa, dtype=bool).reshape(F, mw).copy())

    @staticmethod
    def regular(g: FlatGeometry) -> bool:
        F = g.n_frames
        if F == 0:
            return False
        cnt = np.diff(g.lumen_off)
        if not np.all(cnt == cnt[0]) or cnt[0] == 0:
            return False
        if g.cath_off is not None and not np.all(np.diff(g.cath_off) == np.diff(g.cath_off)[0]):
            return False
        counts = g.meta.get("extra_counts") or {}
        tot = 0
        for k, c in counts.items():
            if int(np.sum(c)) and not np.all(np.asarray(c) == c[0]):
                return False
            tot += int(c[0]) if int(np.sum(c)) else 0
        if g.extra_off is not None:
            if not np.all(np.diff(g.extra_off) == tot):
                return False
        elif tot:
            return False
        if g.lumen_centroids is not None and g.has_lumen_centroid is not None and not np.all(g.has_lumen_centroid):
            return False
        return True

    def per_frame(self):
        return ("ids", "lumen_ids", "orig", "cen", "L", "C", "X", "has_ref", "ref", "lc", "l_aortic", "w_aortic")

    def reorder(self, idx):
        for n in self.per_frame():
            v = getattr(self, n)
            if v is not None:
                setattr(self, n, v[idx])
        self.a_th = [self.a_th[i] for i in idx]
        self.p_th = [self.p_th[i] for i in idx]
        self.F = len(idx)

    def ref_frame_id(self) -> int:
        """find_ref_frame_idx (geometry.rs:62-69): Frame.id of the first frame with a reference point."""
        nz = np.nonzero(self.has_ref)[0]
        if nz.size == 0:
            raise RuntimeError("No reference point found in any frame")
        return int(self.ids[nz[0]])

    def avg_z_diff(self) -> float:
        if self.F < 2:
            return 0.0
        d = self.cen[1:, 2] - self.cen[:-1, 2]
        return float(np.add.accumulate(d)[-1]) / float(self.F - 1)

    def resample_by_diff(self, diff: float):
        """postprocessing.rs:116-140."""
        k = int(np.argmin(self.cen[:, 2]))                         # first minimum, like min_by
        if k:
            self.reorder(list(range(k, self.F)) + list(range(k)))
        start = float(self.cen[0, 2])
        z = start + np.arange(self.F, dtype=np.float64) * diff
        for arr in (self.L, self.C, self.X):
            if arr is not None and self.F > 1:
                arr[1:, :, 2] = z[1:, None]
        if self.F > 1:
            self.cen[1:, 2] = z[1:]
            if self.lc is not None:
                self.lc[1:, 2] = z[1:]
            sel = self.has_ref[1:] != 0
            self.ref[1:, 2] = np.where(sel, z[1:], self.ref[1:, 2])

    def translate_z(self, dz: float):
        """translate_geometry((0, 0, dz)) -> Frame::translate (frame.rs:18-38) for every frame."""
        d = np.array([0.0, 0.0, dz])
        for arr in (self.L, self.C, self.X):
            if arr is not None:
                arr += d
        self.lc = PP.centroids_batched(self.L)                     # lumen.compute_centroid()
        self.ref = np.where((self.has_ref != 0)[:, None], self.ref + d, self.ref)
        self.cen = self.cen + d

    def trim(self, s: int, e: int):
        if s < e <= self.F:
            self.reorder(list(range(s, e)))
        self.ids = np.arange(self.F, dtype=np.uint32)
        self.lumen_ids = np.arange(self.F, dtype=np.uint32)

    def kind_slice(self, kind: str) -> Tuple[int, int]:
        lo = 0
        for k in self.kinds:
            if k == kind:
                return lo, lo + self.kcount[k]
            lo += self.kcount[k]
        raise KeyError(kind)

    def rebuild_walls_anomalous(self):
        """create_wall_frames(frames, anomalous = true, false) (wall.rs:7-34): every wall from the lumen."""
        W = PP.offset_contours_batched(self.L, 1.0)
        wa = self.l_aortic.copy()
        for i in range(self.F):
            if self.a_th[i] is not None:
                c = Contour(int(self.lumen_ids[i]), int(self.orig[i]), self.L[i].copy(),
                            None if self.lc is None else tuple(self.lc[i]), self.a_th[i], self.p_th[i], "lumen",
                            self.l_aortic[i].copy())
                w = PP.create_aortic_wall(c)
                if len(w) != self.m:
                    raise ValueError("irregular aortic wall")
                W[i] = w.points
                wa[i] = w.aortic
        if "wall" in self.kinds:
            lo, hi = self.kind_slice("wall")
            self.X = np.concatenate([self.X[:, :lo], W, self.X[:, hi:]], axis=1)
        else:
            # EXTRA_KINDS order: wall is last
            self.X = W if self.X is None else np.concatenate([self.X, W], axis=1)
            self.kinds.append("wall")
        self.kcount["wall"] = self.m
        self.w_aortic = wa

    def to_flat(self) -> FlatGeometry:
        F = self.F
        g = FlatGeometry(ids=np.ascontiguousarray(self.ids, dtype=np.uint32),
                         lumen_ids=np.ascontiguousarray(self.lumen_ids, dtype=np.uint32),
                         orig_frames=np.ascontiguousarray(self.orig, dtype=np.uint32),
                         centroids=np.ascontiguousarray(self.cen), lumen_off=np.arange(F + 1, dtype=np.int64) * self.m,
                         lumen=np.ascontiguousarray(self.L.reshape(-1, 3)), label=self.label)
        if self.C is not None:
            g.cath_off = np.arange(F + 1, dtype=np.int64) * self.C.shape[1]
            g.cath = np.ascontiguousarray(self.C.reshape(-1, 3))
        if self.X is not None:
            g.extra_off = np.arange(F + 1, dtype=np.int64) * self.X.shape[1]
            g.extra = np.ascontiguousarray(self.X.reshape(-1, 3))
        g.has_ref = np.ascontiguousarray(self.has_ref, dtype=np.uint8)
        g.ref = np.ascontiguousarray(np.where((self.has_ref != 0)[:, None], self.ref, 0.0))
        meta = dict(self.meta)
        meta["extra_counts"] = {k: (np.full(F, self.kcount[k], dtype=np.int64) if k in self.kinds else np.zeros(F, dtype=np.int64))
                                for k in EXTRA_KINDS}
        meta["aortic_thickness"], meta["pulmonary_thickness"] = list(self.a_th), list(self.p_th)
        for key, arr in (("lumen_aortic", self.l_aortic), ("wall_aortic", self.w_aortic)):
            if arr.size and arr.any():
                meta[key] = arr.reshape(-1)
            else:
                meta.pop(key, None)
        g.meta = meta
        if self.lc is not None:
            g.has_lumen_centroid = np.ones(F, dtype=np.uint8)
            g.lumen_centroids = np.ascontiguousarray(self.lc)
        return g


def postprocess_pair_regular(a: FlatGeometry, b: FlatGeometry, tol: float, anomalous: bool
                             ) -> Optional[Tuple[FlatGeometry, FlatGeometry]]:
    """Returns None when the pair is not regular or does not take the same-sample-rate branch; the
    caller then uses postproc.postprocess_pair on the frame-list model."""
    if not (_Reg.regular(a) and _Reg.regular(b)):
        return None
    ra, rb = _Reg(a), _Reg(b)
    da, db = ra.avg_z_diff(), rb.avg_z_diff()
    if not ((da - db) < tol):                                      # postprocessing.rs:89-98
        return None
    ra.ref_frame_id(); rb.ref_frame_id()                           # :19-20 (errors if there is none)
    orig_za, orig_zb = a.centroids[:, 2].copy(), b.centroids[:, 2].copy()
    mean = (da + db) / 2.0
    ra.resample_by_diff(mean); rb.resample_by_diff(mean)           # :24-33
    ja, jb = ra.ref_frame_id(), rb.ref_frame_id()                  # :70-72
    if ja >= len(orig_za) or jb >= len(orig_zb):
        return None
    ra.translate_z(float(orig_za[ja]) - float(orig_zb[jb]))        # :73-76 (the ORIGINAL pair is indexed)

    def ref_or_zero(r):
        nz = np.nonzero(r.has_ref)[0]
        return int(r.ids[nz[0]]) if nz.size else 0

    ia, ib = ref_or_zero(ra), ref_or_zero(rb)                      # trim_geom_pair (:342-409)
    before = min(ia, ib)
    after = min(ra.F - ia, rb.F - ib)
    ra.trim(ia - before, ia + after); rb.trim(ib - before, ib + after)
    if anomalous:                                                  # adjust_walls_anomalous_geom_pair (:411-476)
        n = min(ra.F, rb.F)
        for i in range(n):
            ta, tb = ra.a_th[i], rb.a_th[i]
            if ta is not None or tb is not None:
                th = (ta + tb) / 2.0 if ta is not None and tb is not None else (ta if ta is not None else tb)
                ra.a_th[i] = th; rb.a_th[i] = th
        if ra.F != rb.F:
            return None                                            # zip() would drop frames: leave it to the general path
        try:
            ra.rebuild_walls_anomalous(); rb.rebuild_walls_anomalous()
        except ValueError:
            return None
    return ra.to_flat(), rb.to_flat()
