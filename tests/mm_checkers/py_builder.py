"""The geometry builder in Python (io/build.rs:9-205 restated; checker of mm_build_geometry, csrc/mm_build.cpp): the product's
``build_geometry_from_inputdata`` runs it only under MM_PY_BUILDER=1 (tests/test_refbuild.py, test_property_host.py)."""
from __future__ import annotations

from typing import Dict

import numpy as np

from multimoda_rs_amd.io import (FlatGeometry, InputData, _Frame, _group_by_frame, _integrity_error, _to_flat, contour_centroid, create_catheter_points, sort_contour_order, sort_contour_points)


def build_geometry_python(d: InputData, label: str = "", image_center=(4.5, 4.5), radius: float = 0.5,
                          n_points: int = 20, check_integrity: bool = True) -> FlatGeometry:
    """The same builder in Python (checker of the native one)."""

    # build.rs:37-71 shared original-frame -> sequential-id mapping
    originals = set(int(f) for f in np.unique(d.lumen[:, 0]))
    for arr in (d.eem, d.calcification, d.sidebranch):
        if arr is not None:
            originals.update(int(f) for f in np.unique(arr[:, 0]))
    ref_frame = int(d.ref_point[0])
    originals.add(ref_frame)
    mapping = {o: i for i, o in enumerate(sorted(originals))}

    meas = {}
    if d.record:
        for r in d.record:                                          # contour.rs:172-177
            meas[r.frame] = (r.measurement_1, r.measurement_2)

    frames: Dict[int, _Frame] = {}
    flags = None
    if d.lumen_aortic is not None and np.any(d.lumen_aortic):      # per-point aortic flags of the optional 5th column
        tagged = np.concatenate([d.lumen[:, :1], np.asarray(d.lumen_aortic, dtype=np.float64).reshape(-1, 1),
                                 np.zeros((d.lumen.shape[0], 2))], axis=1)
        flags = {o: v[:, 0] != 0.0 for o, v in _group_by_frame(tagged).items()}
    for orig, pts in sorted(_group_by_frame(d.lumen).items()):     # build.rs:74-129
        fid = mapping[orig]
        m = meas.get(orig, (None, None))
        fr = _Frame(id=fid, orig=orig, lumen=pts, centroid=list(contour_centroid(pts)), aortic=m[0], pulmonary=m[1],
                    lumen_aortic=None if flags is None else flags[orig])
        if mapping.get(ref_frame) == fid:
            fr.ref = [float(d.ref_point[1]), float(d.ref_point[2]), float(d.ref_point[3])]
        frames[fid] = fr
    for kind, arr in (("eem", d.eem), ("calcification", d.calcification), ("sidebranch", d.sidebranch)):
        if arr is None:
            continue
        for orig, pts in _group_by_frame(arr).items():             # build.rs:131-150
            fid = mapping[orig]
            if fid in frames:
                frames[fid].extras[kind] = pts
    if n_points > 0:                                               # build.rs:152-174
        frame_z = {fr.orig: float(fr.lumen[0, 2]) for fr in frames.values()}
        for orig, pts in create_catheter_points(frame_z, image_center, radius, n_points).items():
            fid = mapping[orig]
            if fid in frames:
                frames[fid].extras["catheter"] = pts

    flist = [frames[k] for k in sorted(frames)]                    # build.rs:176-177

    if d.record:                                                   # build.rs:184-186 -> geometry.rs:72-155
        phase = "D" if d.diastole else "S"
        filtered = [r.frame for r in d.record if r.phase == phase]
        by_orig = {fr.orig: fr for fr in flist}
        new, used = [], set()
        for o in filtered:
            if o in by_orig and o not in used:
                new.append(by_orig[o]); used.add(o)
        rest = sorted((fr for fr in flist if fr.orig not in used), key=lambda fr: fr.orig)
        flist = new + rest
        for i, fr in enumerate(flist):
            fr.id = i
            # geometry.rs:77-83,106-141: every z of the frame -- points, extras, reference point AND the frame
            # centroid -- becomes the z of the frame's first lumen point (the centroid's z was the rounded mean
            # of the points' z until here; ensure_proximal_at_position_zero below hands the centroid z's on to
            # all points, so leaving the mean in place put a 1e-13 rounding residue into every output z)
            z = float(fr.lumen[0, 2])
            fr.lumen[:, 2] = z
            for k in fr.extras:
                fr.extras[k][:, 2] = z
            if fr.ref is not None:
                fr.ref[2] = z
            fr.centroid[2] = z

    for fr in flist:                                               # build.rs:188-190
        if fr.lumen_aortic is not None:
            order = sort_contour_order(fr.lumen)
            fr.lumen_aortic = fr.lumen_aortic[order]
            fr.lumen = np.ascontiguousarray(fr.lumen[order])
        else:
            fr.lumen = sort_contour_points(fr.lumen)
        for k in list(fr.extras):
            fr.extras[k] = sort_contour_points(fr.extras[k])

    # build.rs:192 -> geometry.rs:325-381 ensure_proximal_at_position_zero
    n = len(flist)
    if n:
        if n == 1:
            prox = flist[0].id
        else:
            prox = flist[0].id if flist[0].orig > flist[-1].orig else flist[-1].id
        prox = min(prox, n - 1)
        if prox != 0:
            flist.reverse()
        zs = sorted(fr.centroid[2] for fr in flist)
        for i, fr in enumerate(flist):
            fr.id = i
            z = zs[i]
            fr.centroid[2] = z
            fr.lumen[:, 2] = z
            for k in fr.extras:
                fr.extras[k][:, 2] = z
            if fr.ref is not None:
                fr.ref[2] = z
    if n == 0:
        raise RuntimeError("Geometry has no frames")               # integrity_check.rs:9-11
    if check_integrity:                                            # build.rs:199
        why = _integrity_error(flist)
        if why:
            raise RuntimeError(f"build_geometry_from_inputdata: {why}")
    return _to_flat(flist, label or d.label)


