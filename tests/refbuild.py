"""Independent pure-Python restatement of the reference's geometry builder, for the tests only.

It exists so that the golden vectors (tests/make_golden.py) and the config-1 parity tests feed the ORACLE
with inputs that were NOT produced by the product's own reader/builder (multimoda_rs_amd.io): a bug there
would otherwise be invisible.  Written from the reference source, record by record,
with dicts and lists like the Rust code; nothing is imported from the product package.

Follows (paths relative to the reference checkout):
  src/intravascular/io/input.rs:62-146,149-194,212-258   process_directory, delimiter sniffing, readers
  src/intravascular/io/build.rs:9-205                     build_geometry_from_inputdata
  src/types/native/contour.rs:158-224,368-405             build_contour_with_mapping, compute_centroid,
                                                          sort_contour_points
  src/types/native/frame.rs:69-82,123-129,163-204         set_value(id), sort_frame_points, catheter circle
  src/types/native/geometry.rs:42-59,72-155,325-381       find_proximal_end_idx, reorder_frames,
                                                          ensure_proximal_at_position_zero
"""
from __future__ import annotations

import csv
import math
import os
import re

import numpy as np

_U32 = re.compile(r"^\+?[0-9]+$")


def _parse_point(row):
    """serde: ContourPoint { frame_index: u32, x, y, z: f64, aortic: bool (default) } from a headerless row
    (contour_point.rs:55-68; point_index is skipped).  None = the row does not deserialize."""
    if len(row) not in (4, 5):
        return None
    f = row[0]
    if not _U32.match(f) or int(f) > 0xFFFFFFFF:
        return None
    try:
        x, y, z = float(row[1]), float(row[2]), float(row[3])
    except ValueError:
        return None
    for s in row[1:4]:
        if "_" in s or s != s.strip():        # Python's float() accepts these, Rust's str::parse::<f64> does not
            return None
    aortic = False
    if len(row) == 5:
        if row[4] not in ("true", "false"):
            return None
        aortic = row[4] == "true"
    return {"frame": int(f), "x": x, "y": y, "z": z, "aortic": aortic}


def _delimiter(path):
    with open(path, "r", newline="") as fh:
        first = fh.readline()
    return "\t" if first.count("\t") > first.count(",") else ","      # input.rs:149-170


def read_contour_data(path):
    pts = []
    first_len = None
    with open(path, "r", newline="") as fh:
        for row in csv.reader(fh, delimiter=_delimiter(path)):
            if not row:
                continue                                              # the csv crate skips empty lines
            if first_len is None:                                     # csv::ReaderBuilder: flexible(false) -- a record of
                first_len = len(row)                                  # another length than the first is an Err row,
            if len(row) != first_len:                                 # "Skipping invalid row" (input.rs:191)
                continue
            p = _parse_point(row)
            if p is not None:                                         # "Skipping invalid record" (input.rs:186-190)
                pts.append(p)
    return pts


def read_reference_point(path):
    with open(path, "r", newline="") as fh:
        for row in csv.reader(fh, delimiter=_delimiter(path)):
            if not row:
                continue
            p = _parse_point(row)
            if p is None:
                raise ValueError("failed to deserialize first reference-point record")
            return p
    raise ValueError("reference-point file was empty")


def read_records(path):
    """Record { frame, phase, measurement_1, measurement_2 } by header name (record.rs, input.rs:235-250);
    csv::invalid_option: anything that does not parse becomes None."""
    def opt(s):
        try:
            return float(s) if s.strip() == s and "_" not in s and s != "" else None
        except ValueError:
            return None
    out = []
    with open(path, "r", newline="") as fh:
        rd = csv.reader(fh, delimiter=_delimiter(path))
        header = next(rd)
        col = {n: i for i, n in enumerate(header)}
        for row in rd:
            if not row:
                continue
            out.append({"frame": int(row[col["frame"]]), "phase": row[col["phase"]],
                        "m1": opt(row[col["measurement_1"]]), "m2": opt(row[col["measurement_2"]])})
    return out


def _centroid(points):
    sx = sy = sz = 0.0
    for p in points:                                                  # contour.rs:219-223, sequential fold
        sx, sy, sz = sx + p["x"], sy + p["y"], sz + p["z"]
    n = float(len(points))
    return [sx / n, sy / n, sz / n]


def _sort_contour_points(points):
    """contour.rs:368-405: stable sort by atan2 around the xy centroid, then the LAST point of maximal y
    (Iterator::max_by keeps the last of equal maxima) rotated to the front."""
    n = float(len(points))
    if n == 0.0:
        return points
    sx = sy = 0.0
    for p in points:
        sx, sy = sx + p["x"], sy + p["y"]
    cx, cy = sx / n, sy / n
    pts = sorted(points, key=lambda p: math.atan2(p["y"] - cy, p["x"] - cx))     # list.sort is stable, like slice::sort_by
    best = 0
    for i, p in enumerate(pts):
        if p["y"] >= pts[best]["y"]:
            best = i
    return pts[best:] + pts[:best]


def build_geometry(path, diastole, image_center=(4.5, 4.5), radius=0.5, n_points=20):
    """build_geometry_from_inputdata(None, Some(path), ..) -> dict of per-frame lists in final frame order:
    ids, orig_frames, lumens (n,3), catheters (n,3), centroids (3), ref_points {frame position: xyz}."""
    phase = "diastolic" if diastole else "systolic"
    lumen = read_contour_data(os.path.join(path, f"{phase}_contours.csv"))
    ref = read_reference_point(os.path.join(path, f"{phase}_reference_points.csv"))
    extras_in = {}                                                                 # input.rs:100-118: optional contour files
    for kind, stem in (("eem", "eem"), ("calcification", "calcium"), ("sidebranch", "branch")):
        f = os.path.join(path, f"{stem}_{phase}_contours.csv")
        if os.path.exists(f):
            extras_in[kind] = read_contour_data(f)
    rec_path = os.path.join(path, "combined_sorted_manual.csv")
    if not os.path.exists(rec_path):
        rec_path = os.path.join(path, "diastolic_systolic_records.csv")
    records = read_records(rec_path) if os.path.exists(rec_path) else None

    originals = {p["frame"] for p in lumen} | {ref["frame"]}                       # build.rs:36-68: ALL contour kinds
    for pts in extras_in.values():
        originals |= {p["frame"] for p in pts}
    originals = sorted(originals)
    mapping = {o: i for i, o in enumerate(originals)}

    groups = {}
    for p in lumen:                                                                # contour.rs:164-167
        groups.setdefault(p["frame"], []).append(p)
    frames = []
    for orig in sorted(groups):
        pts = groups[orig]
        fid = mapping[orig]
        fr = {"id": fid, "orig": orig, "lumen": pts, "centroid": _centroid(pts), "ref": None, "cath": None, "extras": {}}
        if mapping.get(ref["frame"]) == fid:                                       # build.rs:121-125
            fr["ref"] = dict(ref)
        frames.append(fr)

    by_id = {fr["id"]: fr for fr in frames}                                        # build.rs:131-150: extras join the frame of
    for kind, pts in extras_in.items():                                            # their (mapped) id; frames without a lumen
        g2 = {}                                                                    # do not exist, their extras are dropped
        for p in pts:
            g2.setdefault(p["frame"], []).append(p)
        for orig, cp in g2.items():
            fr = by_id.get(mapping[orig])
            if fr is not None:
                fr["extras"][kind] = cp

    if n_points > 0:                                                               # build.rs:152-174, frame.rs:163-204
        for fr in frames:
            z = fr["lumen"][0]["z"]            # first encountered z of the frame (all points of a frame share it)
            c = []
            for i in range(n_points):
                a = 2.0 * math.pi * float(i) / float(n_points)
                c.append({"frame": fr["orig"], "x": image_center[0] + radius * math.cos(a),
                          "y": image_center[1] + radius * math.sin(a), "z": z, "aortic": False})
            fr["cath"] = c
    frames.sort(key=lambda f: f["id"])

    if records is not None:                                                        # geometry.rs:72-144
        want = [r["frame"] for r in records if r["phase"] == ("D" if diastole else "S")]
        orig_z = {}
        for fr in frames:
            orig_z.setdefault(fr["orig"], fr["lumen"][0]["z"])
        by_orig = {fr["orig"]: fr for fr in frames}
        new = []
        for o in want:
            if o in by_orig:
                new.append(by_orig.pop(o))
        new.extend(sorted(by_orig.values(), key=lambda f: f["orig"]))
        for idx, fr in enumerate(new):
            z = orig_z.get(fr["orig"], float(idx))
            fr["id"] = idx
            for p in fr["lumen"]:
                p["z"] = z
            if fr["cath"] is not None:
                for p in fr["cath"]:
                    p["z"] = z
            for cp in fr["extras"].values():
                for p in cp:
                    p["z"] = z
            if fr["ref"] is not None:
                fr["ref"]["z"] = z
            fr["centroid"][2] = z
        frames = new

    for fr in frames:                                                              # build.rs:186-188
        fr["lumen"] = _sort_contour_points(fr["lumen"])
        if fr["cath"] is not None:
            fr["cath"] = _sort_contour_points(fr["cath"])
        fr["extras"] = {k: _sort_contour_points(v) for k, v in fr["extras"].items()}

    n = len(frames)                                                                # geometry.rs:325-381
    if n:
        if n == 1:
            prox = frames[0]["id"]
        else:
            prox = frames[0]["id"] if frames[0]["orig"] > frames[-1]["orig"] else frames[-1]["id"]
        prox = min(prox, n - 1)
        if prox != 0:
            frames.reverse()
        zs = sorted(fr["centroid"][2] for fr in frames)
        for idx, fr in enumerate(frames):
            fr["id"] = idx                                                         # + build.rs:192-195 set_value(Some(id))
            z = zs[idx]
            fr["centroid"][2] = z
            for p in fr["lumen"]:
                p["z"] = z
            if fr["cath"] is not None:
                for p in fr["cath"]:
                    p["z"] = z
            for cp in fr["extras"].values():
                for p in cp:
                    p["z"] = z
            if fr["ref"] is not None:
                fr["ref"]["z"] = z

    xyz = lambda pts: np.array([[p["x"], p["y"], p["z"]] for p in pts], dtype=np.float64).reshape(-1, 3)
    return {
        "ids": [fr["id"] for fr in frames],
        "orig_frames": [fr["orig"] for fr in frames],
        "lumens": [xyz(fr["lumen"]) for fr in frames],
        "catheters": [xyz(fr["cath"]) for fr in frames] if n_points > 0 else None,
        "centroids": [list(fr["centroid"]) for fr in frames],
        "ref_points": {i: [fr["ref"]["x"], fr["ref"]["y"], fr["ref"]["z"]] for i, fr in enumerate(frames) if fr["ref"] is not None},
        "extras": [{k: xyz(v) for k, v in fr["extras"].items()} for fr in frames],
    }


def oracle_geometry(orc, path, diastole, label="", **kw):
    """The oracle's geometry container filled from the independent builder."""
    b = build_geometry(path, diastole, **kw)
    og = orc.OracleGeometry.from_frames(b["lumens"], catheters=b["catheters"], centroids=b["centroids"], ids=b["ids"],
                                        orig_frames=b["orig_frames"], ref_points=b["ref_points"], label=label)
    # Frame.lumen.centroid as the builder leaves it: the value the frame centroid was copied from (build.rs:113-117),
    # its z rewritten together with the frame's (geometry.rs:119-121,361-363)
    og.lumen_centroids = og.centroids.copy()
    og.has_lumen_centroid = np.ones(og.n_frames, dtype=np.uint8)
    return og
