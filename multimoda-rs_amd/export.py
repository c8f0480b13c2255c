"""OBJ / MTL / texture export of the reference, restated:

* io/output.rs:10-307            write_obj_mesh, write_obj_mesh_without_uv, write_geometry_vec_to_obj
* to_object/process.rs:13-121    process_case (a pair: interpolated geometries + UV maps + textures),
                                 write_single_geometry
* to_object/interpolation.rs:11-149, texture.rs:6-95, write_mtl.rs:19-273
* binding/entry.rs:740-819, binding/functions.rs:1435-1500   single-mode writing, to_obj, MTL per kind

Text files are the reference's byte for byte (vertex / uv / normal / face lines, numbers in Rust's
``{}`` formatting: shortest round-trip digits, never an exponent, no trailing ``.0``).  PNG textures
carry the same pixels; the byte stream of the ``image`` crate's encoder is not reproduced.
"""
from __future__ import annotations

import math
import os
import struct
import zlib
from decimal import Decimal
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .frames import Contour, Frame, to_frames
from .geometry import FlatGeometry

DEFAULT_CONTOUR_TYPES = ("lumen", "catheter", "wall")        # multimodars/_processing.py:33-39
_MTL_SINGLE = {
    "lumen": "newmtl material\nKa 1.0 1.0 1.0\nKd 1.0 1.0 1.0\nKs 0.0 0.0 0.0\n",
    "eem": "newmtl material\nKa 1.0 1.0 1.0\nKd 1.0 1.0 1.0\nKs 0.0 0.0 0.0\n",
    "catheter": "newmtl material\nKa 0.0 0.0 0.0\nKd 0.0 0.0 0.0\nKs 0.0 0.0 0.0\n",
    "calcification": "newmtl material\nKa 0.0 0.0 0.0\nKd 0.0 0.0 0.0\nKs 0.0 0.0 0.0\n",
    "wall": "newmtl material\nKa 0.5 0.5 0.5\nKd 0.5 0.5 0.5\nKs 0.0 0.0 0.0\nd 0.7\n",
    "sidebranch": "newmtl material\nKa 0.5 0.5 0.5\nKd 0.5 0.5 0.5\nKs 0.0 0.0 0.0\nd 0.7\n",
}


def rust_f64(x: float) -> str:
    """``format!("{}", x)`` for an f64: shortest digits that round-trip, plain decimal notation."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    r = repr(x)
    if "e" in r or "E" in r:
        r = format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


def _kind_name(kind) -> str:
    return str(kind).lower()


def extract_contours_by_type(frames: Sequence[Frame], kind: str) -> List[Contour]:
    """process_utils.rs:7-20."""
    if kind == "lumen":
        return [f.lumen for f in frames]
    return [f.extras[kind] for f in frames if kind in f.extras]


# ---- io/output.rs ----------------------------------------------------------------------------------
def write_obj_mesh(contours: Sequence[Contour], uv_coords: Sequence[Tuple[float, float]], filename: str,
                   mtl_filename: str, watertight: bool) -> None:
    """io/output.rs:10-147."""
    parent = os.path.dirname(filename)
    if parent:
        os.makedirs(parent, exist_ok=True)
    if len(contours) < 2:
        raise RuntimeError("Need at least two contours to create a mesh.")
    ppc = len(contours[0])
    if any(len(c) != ppc for c in contours):
        raise RuntimeError("All contours must have the same number of points.")
    total = ppc * len(contours)
    out: List[str] = []
    for c in contours:
        for p in c.points:
            out.append(f"v {rust_f64(p[0])} {rust_f64(p[1])} {rust_f64(p[2])}\n")
    if len(uv_coords) != total:
        # the reference has created (and leaves behind) the partly written file at this point
        with open(filename, "w") as f:
            f.write("".join(out))
        raise RuntimeError(f"UV coordinates must match the number of vertices. Expected {total}, got {len(uv_coords)}.")
    out.append(f"mtllib {mtl_filename}\n")
    out.append("usemtl displacement_material\n")
    for u, v in uv_coords:
        out.append(f"vt {rust_f64(u)} {rust_f64(v)}\n")
    for c in contours:
        cen = c.centroid if c.centroid is not None else (0.0, 0.0, 0.0)
        for p in c.points:
            dx, dy = p[0] - cen[0], p[1] - cen[1]
            ln = math.sqrt(dx * dx + dy * dy)
            nx, ny, nz = (dx / ln, dy / ln, 0.0) if ln > 0.0 else (0.0, 0.0, 0.0)
            out.append(f"vn {rust_f64(-nx)} {rust_f64(-ny)} {rust_f64(-nz)}\n")
    offs = [1 + i * ppc for i in range(len(contours))]
    for c in range(len(contours) - 1):
        o1, o2 = offs[c], offs[c + 1]
        for j in range(ppc):
            jn = (j + 1) % ppc
            v1, v2, v3 = o1 + j, o1 + jn, o2 + j
            out.append(f"f {v1}/{v1}/{v1} {v2}/{v2}/{v2} {v3}/{v3}/{v3}\n")
            a, b, d = o2 + j, o1 + jn, o2 + jn
            out.append(f"f {a}/{a}/{a} {b}/{b}/{b} {d}/{d}/{d}\n")
    if watertight:
        cur = total + 1
        for c, vn in ((contours[0], "vn 0.0 0.0 -1.0\n"), (contours[-1], "vn 0.0 0.0 1.0\n")):   # literals in the reference
            cen = c.centroid if c.centroid is not None else (0.0, 0.0, 0.0)
            out.append(f"v {rust_f64(cen[0])} {rust_f64(cen[1])} {rust_f64(cen[2])}\n")
            out.append("vt 0.5 0.5\n")
            out.append(vn)
        for off, cidx, rev in ((offs[0], cur, False), (offs[-1], cur + 1, True)):      # close_end (:149-170)
            for i in range(ppc):
                v1, v2, v3 = off + i, off + (i + 1) % ppc, cidx
                if rev:
                    out.append(f"f {v3}/{v3}/{v3} {v2}/{v2}/{v2} {v1}/{v1}/{v1}\n")
                else:
                    out.append(f"f {v1}/{v1}/{v1} {v2}/{v2}/{v2} {v3}/{v3}/{v3}\n")
    with open(filename, "w") as f:
        f.write("".join(out))


def write_obj_mesh_without_uv(contours: Sequence[Contour], filename: str, mtl_filename: str, watertight: bool) -> None:
    """io/output.rs:172-188."""
    try:
        write_obj_mesh(contours, [(0.0, 0.0)] * sum(len(c) for c in contours), filename, mtl_filename, watertight)
    except RuntimeError as e:
        raise RuntimeError(f"Failed to write OBJ mesh without UV: {e}") from e


# ---- textures (to_object/texture.rs) -----------------------------------------------------------------
def _png(path: str, pixels: np.ndarray) -> None:
    """8-bit RGB (h, w, 3) or RGBA (h, w, 4) PNG, no interlace."""
    h, w, ch = pixels.shape
    raw = b"".join(b"\x00" + pixels[y].astype(np.uint8).tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if ch == 3 else 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _as_u8(x: float) -> int:
    """Rust ``f64 as u8``: truncation toward zero, saturating, NaN -> 0."""
    if math.isnan(x):
        return 0
    return int(max(0.0, min(255.0, math.trunc(x))))


def compute_uv_coordinates(contours: Sequence[Contour]) -> List[Tuple[float, float]]:
    """texture.rs:6-27."""
    if not contours or len(contours[0]) == 0:
        return []
    ppc, n = len(contours[0]), len(contours)
    uvs = []
    for ci, c in enumerate(contours):
        if len(c) == 0:
            continue
        v = (ci + 0.5) / n
        uvs.extend(((pi + 0.5) / ppc, v) for pi in range(len(c)))
    return uvs


def compute_displacements(frames: Sequence[Frame], base: Sequence[Frame]) -> List[float]:
    """texture.rs:33-51: lumen point displacements against the first geometry."""
    out: List[float] = []
    for f, b in zip(frames, base):
        n = min(len(f.lumen), len(b.lumen))
        d = f.lumen.points[:n] - b.lumen.points[:n]
        out.extend(np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).tolist())
    return out


def create_displacement_texture(displacements: Sequence[float], width: int, height: int, max_disp: float, filename: str):
    """texture.rs:53-75: red = displacement / max, blue = 1 - that, rows flipped."""
    img = np.zeros((height, width, 3), dtype=np.uint8)
    for i, disp in enumerate(displacements):
        x, row = i % width, i // width
        if row >= height:
            raise RuntimeError("Image index out of bounds")           # put_pixel panics in the reference
        with np.errstate(divide="ignore", invalid="ignore"):
            q = float(np.float64(disp) / np.float64(max_disp))
        nrm = 0.0 if math.isnan(q) else min(max(q, 0.0), 1.0)            # f64::clamp; NaN stays NaN -> as u8 = 0
        if math.isnan(q):
            r, b = 0, 0
        else:
            r, b = _as_u8(nrm * 255.0), _as_u8((1.0 - nrm) * 255.0)
        img[height - 1 - row, x] = (r, 0, b)
    _png(filename, img)


# ---- to_object/interpolation.rs ----------------------------------------------------------------------
def _lerp_contour(s: Contour, e: Contour, t: float) -> Contour:
    if len(s) != len(e):
        raise RuntimeError("Contour point counts do not match between start and end")
    pts = s.points * (1.0 - t) + e.points * t
    if s.centroid is not None and e.centroid is not None:
        cen = tuple(s.centroid[k] * (1.0 - t) + e.centroid[k] * t for k in range(3))
    else:
        cen = s.centroid if s.centroid is not None else e.centroid
    both = lambda a, b: a * (1.0 - t) + b * t if a is not None and b is not None else None
    return Contour(s.id, s.original_frame, pts, cen, both(s.aortic_thickness, e.aortic_thickness),
                   both(s.pulmonary_thickness, e.pulmonary_thickness), s.kind, s.aortic.copy())


def interpolate_contours(start: Sequence[Frame], end: Sequence[Frame], steps: int, contour_types: Sequence[str]
                         ) -> List[List[Frame]]:
    """interpolation.rs:11-89: [start, steps interpolated geometries (t = step / (steps - 1)), end]."""
    n = min(len(start), len(end))
    geoms = [[f.clone() for f in start]]
    for step in range(steps):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = float(np.float64(step) / np.float64(steps - 1))            # steps == 1: 0/0 = NaN, as in the reference
        frames = []
        for i in range(n):
            sf, ef = start[i], end[i]
            extras = {}
            for k in contour_types:
                if k != "lumen" and k in sf.extras and k in ef.extras:
                    extras[k] = _lerp_contour(sf.extras[k], ef.extras[k], t)
            if sf.reference_point is not None and ef.reference_point is not None:
                ref = sf.reference_point * (1.0 - t) + ef.reference_point * t
            else:
                ref = sf.reference_point if sf.reference_point is not None else ef.reference_point
                ref = None if ref is None else ref.copy()
            cen = [sf.centroid[k] * (1.0 - t) + ef.centroid[k] * t for k in range(3)]
            frames.append(Frame(sf.id, cen, _lerp_contour(sf.lumen, ef.lumen, t), extras, ref))
        geoms.append(frames)
    geoms.append([f.clone() for f in end])
    return geoms


# ---- to_object/write_mtl.rs + process.rs --------------------------------------------------------------
def _write_mtl_for_type(geoms: Sequence[Sequence[Frame]], output_dir: str, case_name: str, kind: str):
    """write_mtl.rs:37-253: per geometry a texture + MTL; returns the UV map of every geometry."""
    uv_all = []
    max_disp = 1.0
    if kind in ("lumen", "eem") and len(geoms) > 1:
        s, e = extract_contours_by_type(geoms[0], kind), extract_contours_by_type(geoms[-1], kind)
        if s and e:
            m = 0.0
            for a, b in zip(s, e):
                n = min(len(a), len(b))
                d = a.points[:n] - b.points[:n]
                if n:
                    m = max(m, float(np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).max()))
            max_disp = m
    for i, frames in enumerate(geoms):
        contours = extract_contours_by_type(frames, kind)
        if not contours:
            uv_all.append([])
            continue
        uv_all.append(compute_uv_coordinates(contours))
        h, w = len(contours), len(contours[0])
        tex = f"{kind}_{i:03d}_{case_name}.png"
        path = os.path.join(output_dir, tex)
        try:
            if kind in ("lumen", "eem"):
                create_displacement_texture(compute_displacements(frames, geoms[0]), w, h, max_disp, path)
                mat, ka = "displacement_material", "1 1 1"
            elif kind in ("catheter", "calcification"):
                _png(path, np.zeros((h, w, 3), dtype=np.uint8))
                mat, ka = "black_material", "0 0 0"
            else:
                px = np.zeros((h, w, 4), dtype=np.uint8)
                px[..., 3] = _as_u8(255.0 - (0.7 * 255.0))
                _png(path, px)
                mat, ka = "transparent_material", "0 0 0"
        except (RuntimeError, OSError, ValueError):
            continue                                                       # "Failed to create ... texture": MTL skipped
        with open(os.path.join(output_dir, f"{kind}_{i:03d}_{case_name}.mtl"), "w") as f:
            f.write(f"newmtl {mat}\nKa {ka}\nKd {ka}\nmap_Kd {tex}\n")
    return uv_all


def process_case(case_name: str, frames_a: Sequence[Frame], frames_b: Sequence[Frame], output_dir: str,
                 interpolation_steps: int, watertight: bool, contour_types: Sequence[str]) -> None:
    """to_object/process.rs:13-62: OBJ + MTL + texture per contour type and interpolated geometry."""
    os.makedirs(output_dir, exist_ok=True)
    kinds = [_kind_name(k) for k in contour_types]
    geoms = interpolate_contours(frames_a, frames_b, interpolation_steps, kinds)
    uv_map = {k: _write_mtl_for_type(geoms, output_dir, case_name, k) for k in kinds}
    for k in kinds:
        errors = []
        for i, (frames, uv) in enumerate(zip(geoms, uv_map[k])):          # write_geometry_vec_to_obj (output.rs:244-307)
            obj = f"{k}_{i:03d}_{case_name}.obj"
            try:
                write_obj_mesh(extract_contours_by_type(frames, k), uv, os.path.join(output_dir, obj),
                               f"{k}_{i:03d}_{case_name}.mtl", watertight)
            except RuntimeError as e:
                errors.append(f"Failed [{obj}]: {e}")
        if errors:
            raise RuntimeError("Some .obj writes failed:\n" + "\n".join(errors))


def _write_types(frames: Sequence[Frame], output_dir: str, watertight: bool, contour_types: Sequence[str], name) -> None:
    os.makedirs(output_dir, exist_ok=True)
    for k in (_kind_name(c) for c in contour_types):
        contours = extract_contours_by_type(frames, k)
        if not contours:
            continue                                                       # "Warning: No contours found ..., skipping"
        obj, mtl = name(k, "obj"), name(k, "mtl")
        with open(os.path.join(output_dir, mtl), "w") as f:
            f.write(_MTL_SINGLE[k])
        write_obj_mesh_without_uv(contours, os.path.join(output_dir, obj), os.path.join(output_dir, mtl), watertight)


def write_single_geometry(case_name: str, geometry: FlatGeometry, output_dir: str, watertight: bool,
                          contour_types: Sequence[str] = DEFAULT_CONTOUR_TYPES) -> None:
    """to_object/process.rs:65-121 (the align_* entry points on a single geometry): <case>_<type>.obj."""
    _write_types(to_frames(geometry), output_dir, watertight, contour_types, lambda k, ext: f"{case_name}_{k}.{ext}")


def write_single_mode(geometry: FlatGeometry, output_path: str, watertight: bool,
                      contour_types: Sequence[str] = DEFAULT_CONTOUR_TYPES) -> None:
    """binding/entry.rs:740-776 (from_*_single with write_obj): <type>_<label>.obj."""
    _write_types(to_frames(geometry), output_path, watertight, contour_types,
                 lambda k, ext: f"{k}_{geometry.label}.{ext}")


def to_obj(geometry: FlatGeometry, output_path: str, watertight: bool = True,
           contour_types: Optional[Sequence[str]] = None, filename_prefix: str = "") -> None:
    """multimodars/_processing.py:1308-1352 / binding/functions.rs:1435-1500: <prefix>_<type>.obj per type."""
    kinds = DEFAULT_CONTOUR_TYPES if contour_types is None else contour_types
    try:
        _write_types(to_frames(geometry), output_path, watertight, kinds,
                     lambda k, ext: f"{filename_prefix}_{k}.{ext}" if filename_prefix else f"{k}.{ext}")
    except OSError as e:
        raise RuntimeError(f"Could not create output directory '{output_path}': {e}") from e
