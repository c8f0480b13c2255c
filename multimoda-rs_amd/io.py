"""Input data and geometry builder: host-side restatement of the reference's
``io/input.rs`` (CSV readers) and ``io/build.rs:9-205`` (``build_geometry_from_inputdata``)
with the ``Geometry`` helpers it calls (``reorder_frames`` geometry.rs:72-155,
``sort_contour_points`` contour.rs:368-405, ``ensure_proximal_at_position_zero``
geometry.rs:325-381).  Pure host bookkeeping around the hot path;
needed so that ``mm.from_file_*`` / ``mm.from_array_*`` can be driven from the reference's
own inputs.  Scalars are Python floats (IEEE f64, no fused multiply-add) and ``math.*`` is
glibc, i.e. the arithmetic of the Rust code on linux-gnu.
"""
from __future__ import annotations

import csv
import math
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from ._libm import sincos

from .geometry import FlatGeometry, contour_centroid

RECORD_FILE_NAME = "combined_sorted_manual.csv"          # input.rs:12
RECORD_FILE_NAME_ALT = "diastolic_systolic_records.csv"  # input.rs:13
EXTRA_KINDS = ("eem", "calcification", "sidebranch", "wall")   # "wall" is synthesised (processing/wall.rs), never read


@dataclass
class Record:
    """types/native/record.rs"""
    frame: int
    phase: str
    measurement_1: Optional[float] = None
    measurement_2: Optional[float] = None


@dataclass
class InputData:
    """io/input.rs:28-37; point arrays are (N, 4) ``[frame_index, x, y, z]``
    (the ``numpy_to_inputdata`` contract of multimodars/_converters.py:204-437)."""
    lumen: np.ndarray
    ref_point: np.ndarray                      # (4,) [frame_index, x, y, z]
    diastole: bool = True
    label: str = ""
    eem: Optional[np.ndarray] = None
    calcification: Optional[np.ndarray] = None
    sidebranch: Optional[np.ndarray] = None
    record: Optional[List[Record]] = None
    lumen_aortic: Optional[np.ndarray] = None  # (N,) bool: ContourPoint.aortic of the lumen rows (optional 5th CSV column)


def _as_points4(arr, name: str) -> Optional[np.ndarray]:
    if arr is None:
        return None
    a = np.asarray(arr, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError(f"{name}: expected an (N, 4) array [frame_index, x, y, z]")
    return np.ascontiguousarray(a)


def numpy_to_inputdata(lumen_arr, ref_point, diastole: bool, record=None, eem_arr=None, calcification=None,
                       sidebranch=None, label: str = "") -> InputData:
    """Mirror of ``multimodars.numpy_to_inputdata`` (_converters.py:204-437)."""
    lum = _as_points4(lumen_arr, "lumen_arr")
    if lum is None or lum.shape[0] == 0:
        raise ValueError("lumen_arr is empty")
    rp = np.asarray(ref_point, dtype=np.float64).reshape(-1)
    if rp.shape[0] != 4:
        raise ValueError("ref_point: expected [frame_index, x, y, z]")
    recs = None
    if record is not None:
        recs = []
        for row in np.asarray(record, dtype=object).reshape(-1, 4):
            m1 = None if row[2] is None or (isinstance(row[2], float) and math.isnan(row[2])) else float(row[2])
            m2 = None if row[3] is None or (isinstance(row[3], float) and math.isnan(row[3])) else float(row[3])
            recs.append(Record(int(row[0]), str(row[1]), m1, m2))
    return InputData(lumen=lum, ref_point=rp, diastole=bool(diastole), label=label, eem=_as_points4(eem_arr, "eem_arr"),
                     calcification=_as_points4(calcification, "calcification"),
                     sidebranch=_as_points4(sidebranch, "sidebranch"), record=recs)


# ---------------------------------------------------------------------------------------
# CSV readers (io/input.rs)
# ---------------------------------------------------------------------------------------
def _detect_delimiter(path: str) -> str:
    """input.rs:149-170: tabs > commas on the first line -> tab, else comma."""
    with open(path, "r", newline="") as f:
        first = f.readline()
    return "\t" if first.count("\t") > first.count(",") else ","


def _read_numeric_table_native(path: str, delim: str) -> Optional[np.ndarray]:
    """The same fast path in the native library (``mm_parse_contour_table``, include/mm_hausdorff.h): one pass
    over the bytes, correctly rounded conversions, no interpreter lock while it runs -- the pullbacks' files are
    read by parallel threads.  None if the file is not of the regular form or the library is unavailable."""
    try:
        from . import _native as N
        lib = N.lib()
    except Exception:
        return None
    with open(path, "rb") as f:
        raw = f.read()
    if not raw:
        return None
    cap = raw.count(b"\n") + 1
    out = np.empty((cap, 4), dtype=np.float64)
    rows = lib.mm_parse_contour_table(raw, len(raw), delim.encode("ascii"), N._ptr(out), cap)
    if rows < 0 or rows > cap:
        return None
    return out[:rows]


def _read_numeric_table(path: str, delim: str) -> Optional[np.ndarray]:
    """Fast path of read_contour_data for the regular case -- every line is exactly four plain numbers, the
    first a u32 written in digits only (native parser, ``mm_parse_contour_table``).  Anything irregular returns
    None and the row reader decides record by record, like the reference."""
    return _read_numeric_table_native(path, delim)


# Rust's grammars for the two field types of ContourPoint (contour_point.rs:55-68): u32::from_str takes an
# optional '+' and decimal digits; f64::from_str takes [+-] digits [. digits] [e [+-] digits] (either digit
# run may be empty, not both) or inf / infinity / nan in any case.  No surrounding blanks, no underscores --
# Python's int() / float() accept both, so the fields are checked before they are converted.
_RE_U32 = re.compile(r"^\+?[0-9]+$")
_RE_F64 = re.compile(r"^[+-]?(?:(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?|[iI][nN][fF](?:[iI][nN][iI][tT][yY])?|[nN][aA][nN])$")


def _parse_contour_row(rec):
    """One headerless record -> (frame, x, y, z, aortic) as serde deserialises ContourPoint from it, or None
    where the reference skips the row ("Skipping invalid record", input.rs:186-190): fewer than four fields, a
    frame index that is not a u32, a coordinate that is not an f64, or a fifth field that is not true / false."""
    if len(rec) < 4:
        return None
    if not _RE_U32.match(rec[0]):
        return None
    fi = int(rec[0])
    if fi > 0xFFFFFFFF:
        return None
    for v in rec[1:4]:
        if not _RE_F64.match(v):
            return None
    aortic = False
    if len(rec) >= 5:
        if rec[4] not in ("true", "false"):
            return None
        aortic = rec[4] == "true"
    return float(fi), float(rec[1]), float(rec[2]), float(rec[3]), aortic


def read_contour_data(path: str, with_aortic: bool = False):
    """input.rs:172-194: headerless ``frame,x,y,z[,aortic]`` rows; invalid rows are skipped.
    Returns the (N, 4) array, with_aortic=True: (array, (N,) bool flags of the optional fifth column)."""
    delim = _detect_delimiter(path)
    fast = _read_numeric_table(path, delim)
    if fast is not None:
        return (fast, np.zeros(fast.shape[0], dtype=bool)) if with_aortic else fast
    rows, flags = [], []
    width = None
    with open(path, "r", newline="") as f:
        for rec in csv.reader(f, delimiter=delim):
            if not rec:
                continue                                   # the csv crate skips empty lines
            # csv::ReaderBuilder's default is flexible(false): a record whose field count differs from the FIRST
            # record's is Err(UnequalLengths) -> "Skipping invalid row" (input.rs:191), whatever it holds
            if width is None:
                width = len(rec)
            elif len(rec) != width:
                continue
            r = _parse_contour_row(rec)
            if r is not None:
                rows.append(r[:4]); flags.append(r[4])
    arr = np.array(rows, dtype=np.float64).reshape(-1, 4)
    return (arr, np.array(flags, dtype=bool)) if with_aortic else arr


def read_reference_point(path: str) -> np.ndarray:
    """input.rs:213-233: the FIRST record; an empty file is an error, and so is a first record that does not
    deserialise ("failed to deserialize first reference-point record") -- it is not skipped."""
    delim = _detect_delimiter(path)
    with open(path, "r", newline="") as f:
        for rec in csv.reader(f, delimiter=delim):
            if not rec:
                continue
            r = _parse_contour_row(rec)
            if r is None:
                raise RuntimeError(f"failed to deserialize first reference-point record of {path!r}")
            return np.array(r[:4], dtype=np.float64)
    raise RuntimeError(f"reference-point file {path!r} was empty — this data is required")


def read_records(path: str) -> List[Record]:
    """input.rs:235-249 + record.rs: header row, columns by name; unparsable measurements -> None."""
    delim = _detect_delimiter(path)
    out = []
    with open(path, "r", newline="") as f:
        for row in csv.DictReader(f, delimiter=delim):
            def opt(v):
                try:
                    return float(v)
                except (TypeError, ValueError):
                    return None
            out.append(Record(int(row["frame"]), str(row["phase"]), opt(row.get("measurement_1")),
                              opt(row.get("measurement_2"))))
    return out


def process_directory(path: str, diastole: bool, label: str) -> InputData:
    """``InputData::process_directory`` (input.rs:62-146) with the default name mapping of
    build.rs:22-27."""
    phase = "diastolic" if diastole else "systolic"
    contours_path = os.path.join(path, f"{phase}_contours.csv")
    if not os.path.exists(contours_path):
        raise RuntimeError(f"required contours file missing: {contours_path!r}")
    ref_path = os.path.join(path, f"{phase}_reference_points.csv")
    if not os.path.exists(ref_path):
        raise RuntimeError(f"required reference-point file missing: {ref_path!r}")

    def optional(prefix):
        p = os.path.join(path, f"{prefix}_{phase}_contours.csv")
        return read_contour_data(p) if os.path.exists(p) else None

    rec_path = os.path.join(path, RECORD_FILE_NAME)
    if not os.path.exists(rec_path):
        rec_path = os.path.join(path, RECORD_FILE_NAME_ALT)
    record = read_records(rec_path) if os.path.exists(rec_path) else None
    lumen, aortic = read_contour_data(contours_path, with_aortic=True)
    return InputData(lumen=lumen, ref_point=read_reference_point(ref_path),
                     diastole=diastole, label=label, eem=optional("eem"), calcification=optional("calcium"),
                     sidebranch=optional("branch"), record=record, lumen_aortic=aortic if aortic.any() else None)


# ---------------------------------------------------------------------------------------
# contour helpers
# ---------------------------------------------------------------------------------------
def sort_contour_points(points: np.ndarray) -> np.ndarray:
    """``Contour::sort_contour_points`` (contour.rs:368-405): ascending atan2 about the
    centroid (stable), then rotated so that the point with the highest y comes first."""
    n = points.shape[0]
    if n == 0:
        return points
    from . import _native as N
    if os.path.exists(N.LIB_PATH):                     # same function behind the C ABI (host-only, glibc atan2)
        out = np.ascontiguousarray(points, dtype=np.float64).copy()
        N.check(N.lib().mm_sort_contour_points(N._ptr(out), n), "sort_contour_points")
        return out
    return points[sort_contour_order(points)]


def sort_contour_order(points: np.ndarray) -> np.ndarray:
    """The permutation ``sort_contour_points`` applies (new position -> old index), pure Python."""
    n = points.shape[0]
    sx = sy = 0.0
    for p in points:                                   # fold((0,0), |(sx,sy),p| (sx+p.x, sy+p.y))
        sx += float(p[0]); sy += float(p[1])
    cx, cy = sx / float(n), sy / float(n)
    keys = [math.atan2(float(p[1]) - cy, float(p[0]) - cx) for p in points]
    order = sorted(range(n), key=keys.__getitem__)     # stable, like slice::sort_by
    best = 0
    for i in range(n):                                 # Iterator::max_by keeps the LAST maximum
        if points[order[i], 1] >= points[order[best], 1]:
            best = i
    return np.array(order[best:] + order[:best], dtype=np.int64)


def create_catheter_points(frame_z: Dict[int, float], image_center, radius: float, n_points: int) -> Dict[int, np.ndarray]:
    """``Frame::create_catheter_points`` (frame.rs:163-204): per original frame index one circle."""
    out = {}
    for frame in sorted(frame_z):
        z = frame_z[frame]
        pts = np.empty((n_points, 3), dtype=np.float64)
        for i in range(n_points):
            angle = 2.0 * math.pi * float(i) / float(n_points)
            si, co = sincos(angle)                  # frame.rs:192-193: cos and sin of one value
            pts[i, 0] = image_center[0] + radius * co
            pts[i, 1] = image_center[1] + radius * si
            pts[i, 2] = z
        out[frame] = pts
    return out


@dataclass
class _Frame:
    id: int
    orig: int
    lumen: np.ndarray                    # (n,3)
    centroid: List[float]
    extras: Dict[str, np.ndarray] = field(default_factory=dict)   # kind -> (n,3)
    ref: Optional[List[float]] = None
    aortic: Optional[float] = None
    pulmonary: Optional[float] = None
    lumen_aortic: Optional[np.ndarray] = None      # per-point ContourPoint.aortic read from the file (rare)


def _group_by_frame(arr: np.ndarray) -> Dict[int, np.ndarray]:
    """HashMap<u32, Vec<ContourPoint>> with points in input order (contour.rs:163-166)."""
    fi = arr[:, 0].astype(np.int64)
    order = np.argsort(fi, kind="stable")                 # rows of one frame stay in input order
    keys, starts = np.unique(fi[order], return_index=True)
    bounds = list(starts) + [fi.shape[0]]
    first_seen = np.argsort(order[starts], kind="stable") # dict order = order of first appearance
    return {int(keys[k]): arr[order[bounds[k]:bounds[k + 1]], 1:4].copy() for k in first_seen}


def build_geometry_from_inputdata(input_data: Optional[InputData] = None, path: Optional[str] = None, label: str = "",
                                  diastole: bool = True, image_center=(4.5, 4.5), radius: float = 0.5,
                                  n_points: int = 20, check_integrity: bool = True) -> FlatGeometry:
    """``build_geometry_from_inputdata`` (io/build.rs:9-205): behind the C ABI (``mm_build_geometry``,
    include/mm_build.h, csrc/mm_build.cpp).  ``tests/mm_checkers/py_builder.py`` is the same builder in Python, kept as
    a checker (tests/test_refbuild.py compares both with the independent restatement in tests/refbuild.py).

    ``check_integrity`` (not a reference parameter; default = the reference's behaviour): the builder ends with
    ``check_geometry_integrity`` (build.rs:199) and raises RuntimeError with the reference's message when frames
    differ in point count or no frame carries the reference point.  False = ``mm_build_geometry_lenient``."""
    if input_data is None:
        if path is None:
            raise RuntimeError("Either input_data or path must be provided")
        input_data = process_directory(path, diastole, label)
    if os.environ.get("MM_PY_BUILDER"):          # the test suite's Python checker of the builder (tests/mm_checkers)
        from .api import _checker
        return _checker("py_builder").build_geometry_python(input_data, label, image_center, radius, n_points, check_integrity)
    return _build_geometry_native(input_data, label, image_center, radius, n_points, check_integrity)


def _build_geometry_native(d: InputData, label: str, image_center, radius: float, n_points: int,
                           check_integrity: bool = True) -> FlatGeometry:
    import ctypes as C
    from . import _native as N
    L = N.lib()
    rows = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 4)
    lum, eem, calc, side = rows(d.lumen), rows(d.eem), rows(d.calcification), rows(d.sidebranch)
    ref = np.ascontiguousarray(d.ref_point, dtype=np.float64).reshape(4)
    flags = None
    if d.lumen_aortic is not None and np.any(d.lumen_aortic):
        flags = np.ascontiguousarray(d.lumen_aortic, dtype=np.uint8)
    recs = None
    if d.record is not None:
        recs = (N.MMRecord * max(len(d.record), 1))()
        for i, r in enumerate(d.record):
            recs[i].frame = int(r.frame)
            recs[i].phase = 0 if r.phase == "D" else (1 if r.phase == "S" else 2)
            recs[i].has_m1 = r.measurement_1 is not None
            recs[i].has_m2 = r.measurement_2 is not None
            recs[i].m1 = 0.0 if r.measurement_1 is None else float(r.measurement_1)
            recs[i].m2 = 0.0 if r.measurement_2 is None else float(r.measurement_2)
    h = C.c_void_p()
    n_of = lambda a: 0 if a is None else a.shape[0]
    build = L.mm_build_geometry if check_integrity else L.mm_build_geometry_lenient
    N.check(build(N._ptr(lum), lum.shape[0], N._ptr(flags), N._ptr(eem), n_of(eem), N._ptr(calc), n_of(calc),
                  N._ptr(side), n_of(side), N._ptr(ref), recs, 0 if d.record is None else len(d.record),
                  int(bool(d.diastole)), float(image_center[0]), float(image_center[1]), float(radius),
                  int(n_points), C.byref(h)), "build_geometry_from_inputdata")
    try:
        F, nl, nc, ne = C.c_int32(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        N.check(L.mm_built_dims(h, C.byref(F), C.byref(nl), C.byref(nc), C.byref(ne)), "mm_built_dims")
        F, nl, nc, ne = F.value, nl.value, nc.value, ne.value
        g = FlatGeometry(ids=np.zeros(F, np.uint32), lumen_ids=np.zeros(F, np.uint32), orig_frames=np.zeros(F, np.uint32),
                         centroids=np.zeros((F, 3)), lumen_off=np.zeros(F + 1, np.int64), lumen=np.zeros((nl, 3)),
                         has_ref=np.zeros(F, np.uint8), ref=np.zeros((F, 3)), label=label or d.label)
        if nc:
            g.cath_off, g.cath = np.zeros(F + 1, np.int64), np.zeros((nc, 3))
        if ne:
            g.extra_off, g.extra = np.zeros(F + 1, np.int64), np.zeros((ne, 3))
        counts = np.zeros((F, 3), dtype=np.int64)
        a_th, p_th = np.zeros(F), np.zeros(F)
        has_a, has_p = np.zeros(F, np.uint8), np.zeros(F, np.uint8)
        fl_out = np.zeros(nl, dtype=np.uint8) if flags is not None else None
        st = N.MMGeometry()
        p = N._ptr
        st.id, st.lumen_id, st.orig_frame, st.centroid = p(g.ids), p(g.lumen_ids), p(g.orig_frames), p(g.centroids)
        st.lumen_off, st.lumen, st.cath_off, st.cath = p(g.lumen_off), p(g.lumen), p(g.cath_off), p(g.cath)
        st.extra_off, st.extra, st.has_ref, st.ref = p(g.extra_off), p(g.extra), p(g.has_ref), p(g.ref)
        N.check(L.mm_built_export(h, C.byref(st), p(counts), p(a_th), p(has_a), p(p_th), p(has_p), p(fl_out)), "mm_built_export")
    finally:
        L.mm_built_destroy(h)
    # Contour.centroid of every lumen as the builder leaves it: compute_centroid at build.rs:113 (the value the frame
    # centroid is copied from), z rewritten together with the frame's (geometry.rs:119-121,361-363)
    g.has_lumen_centroid, g.lumen_centroids = np.ones(F, dtype=np.uint8), g.centroids.copy()
    g.meta["extra_counts"] = {"eem": counts[:, 0].copy(), "calcification": counts[:, 1].copy(),
                              "sidebranch": counts[:, 2].copy(), "wall": np.zeros(F, dtype=np.int64)}
    if fl_out is not None and fl_out.any():
        g.meta["lumen_aortic"] = fl_out.astype(bool)
    g.meta["aortic_thickness"] = [float(a_th[i]) if has_a[i] else None for i in range(F)]
    g.meta["pulmonary_thickness"] = [float(p_th[i]) if has_p[i] else None for i in range(F)]
    return g


def _integrity_error(flist) -> Optional[str]:
    """check_geometry_integrity (integrity_check.rs:8-33) on the checker builder's frames: the checks that a built
    geometry can fail, with the reference's messages (ids, original frames and stored centroids hold by
    construction; csrc/mm_build.cpp walks all eight)."""
    n_ref = sum(1 for fr in flist if fr.ref is not None)               # :107-118
    if n_ref != 1:
        return f"Expected exactly one reference point, found {n_ref}"
    names = {"eem": "Eem", "calcification": "Calcification", "sidebranch": "Sidebranch", "catheter": "Catheter"}
    expected = {}                                                      # :121-166
    for i, fr in enumerate(flist):
        n = fr.lumen.shape[0]
        if expected.setdefault("lumen", n) != n:
            return f"Lumen point count mismatch in frame {i} (ID {fr.id}). Expected {expected['lumen']}, found {n}"
        for k in ("eem", "calcification", "sidebranch", "catheter"):
            if k in fr.extras:
                n = fr.extras[k].shape[0]
                if expected.setdefault(k, n) != n:
                    return (f"{names[k]} contour point count mismatch in frame {i} (ID {fr.id}). "
                            f"Expected {expected[k]}, found {n}")
    return None


def _to_flat(flist: Sequence[_Frame], label: str) -> FlatGeometry:
    F = len(flist)
    has_cath = all("catheter" in fr.extras for fr in flist) and F > 0
    g = FlatGeometry.from_frames(
        [fr.lumen for fr in flist],
        catheters=[fr.extras["catheter"] for fr in flist] if has_cath else None,
        centroids=[fr.centroid for fr in flist], ids=[fr.id for fr in flist], orig_frames=[fr.orig for fr in flist],
        ref_points={i: fr.ref for i, fr in enumerate(flist) if fr.ref is not None}, label=label)
    # every other extras contour, concatenated per frame in EXTRA_KINDS order
    counts = {k: np.array([fr.extras[k].shape[0] if k in fr.extras else 0 for fr in flist], dtype=np.int64)
              for k in EXTRA_KINDS}
    if any(c.sum() for c in counts.values()):
        off = np.zeros(F + 1, dtype=np.int64)
        chunks = []
        for i, fr in enumerate(flist):
            tot = 0
            for k in EXTRA_KINDS:
                if k in fr.extras:
                    chunks.append(fr.extras[k]); tot += fr.extras[k].shape[0]
            off[i + 1] = off[i] + tot
        g.extra_off = off
        g.extra = np.ascontiguousarray(np.concatenate(chunks, axis=0))
    g.has_lumen_centroid, g.lumen_centroids = np.ones(F, dtype=np.uint8), g.centroids.copy()
    g.meta["extra_counts"] = counts
    if any(fr.lumen_aortic is not None and fr.lumen_aortic.any() for fr in flist):
        g.meta["lumen_aortic"] = np.concatenate([fr.lumen_aortic if fr.lumen_aortic is not None
                                                 else np.zeros(fr.lumen.shape[0], dtype=bool) for fr in flist])
    g.meta["aortic_thickness"] = [fr.aortic for fr in flist]
    g.meta["pulmonary_thickness"] = [fr.pulmonary for fr in flist]
    return g
