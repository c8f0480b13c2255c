"""Synthetic geometries of the reference's own unit tests, rebuilt with plain Python floats
in the reference's operation order (src/intravascular/utils/test_utils.rs:111-376).
Python float arithmetic is IEEE f64 without fused multiply-add and math.sin/cos are glibc's,
i.e. what Rust's f64::sin/cos call on linux-gnu.  Data only -- no reference code."""
from __future__ import annotations

import math

import ctypes

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]


def sincos(x):
    """one glibc sincos call (what the reference's cos()/sin() pair compiles to on linux-gnu)"""
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(float(x), ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value

BASE6 = [(1.0, 3.0), (0.0, 2.0), (0.0, 0.0), (1.0, 0.0), (2.0, 0.0), (2.0, 2.0)]  # test_utils.rs:28-77


def centroid(pts):
    sx = sy = sz = 0.0
    for p in pts:
        sx += p[0]; sy += p[1]; sz += p[2]
    n = float(len(pts))
    return [sx / n, sy / n, sz / n]


def rotate_pt(p, angle, c):
    """contour_point.rs:38-52"""
    if angle == 0.0:
        return list(p)
    x = p[0] - c[0]
    y = p[1] - c[1]
    sa, ca = sincos(angle)
    return [x * ca - y * sa + c[0], x * sa + y * ca + c[1], p[2]]


class Fr:
    """Minimal Frame: lumen points, frame centroid, lumen centroid, optional ref point."""

    def __init__(self, fid, pts, orig):
        self.id = fid
        self.pts = [list(p) for p in pts]
        self.orig = orig
        self.lumen_centroid = centroid(self.pts)
        self.centroid = list(self.lumen_centroid)
        self.ref = None

    def clone(self):
        f = Fr(self.id, self.pts, self.orig)
        f.lumen_centroid = list(self.lumen_centroid)
        f.centroid = list(self.centroid)
        f.ref = None if self.ref is None else list(self.ref)
        return f

    def translate(self, dx, dy, dz):  # frame.rs:18-38
        self.pts = [[p[0] + dx, p[1] + dy, p[2] + dz] for p in self.pts]
        self.lumen_centroid = centroid(self.pts)
        if self.ref is not None:
            self.ref = [self.ref[0] + dx, self.ref[1] + dy, self.ref[2] + dz]
        self.centroid = [self.centroid[0] + dx, self.centroid[1] + dy, self.centroid[2] + dz]

    def rotate(self, angle, c):  # frame.rs:40-63
        if angle == 0.0:
            return
        self.pts = [rotate_pt(p, angle, c) for p in self.pts]
        if self.ref is not None:
            self.ref = rotate_pt(self.ref, angle, c)
        x = self.centroid[0] - c[0]
        y = self.centroid[1] - c[1]
        sa, ca = sincos(angle)
        self.centroid = [x * ca - y * sa + c[0], x * sa + y * ca + c[1], self.centroid[2]]

    def sort_points(self):  # contour.rs:368-405
        n = float(len(self.pts))
        sx = sy = 0.0
        for p in self.pts:
            sx += p[0]; sy += p[1]
        cx, cy = sx / n, sy / n
        self.pts = sorted(self.pts, key=lambda p: math.atan2(p[1] - cy, p[0] - cx))  # stable, like sort_by
        # Iterator::max_by returns the LAST maximum
        best = 0
        for i, p in enumerate(self.pts):
            if p[1] >= self.pts[best][1]:
                best = i
        self.pts = self.pts[best:] + self.pts[:best]


def dummy_frames():
    """test_utils.rs:111-335 dummy_geometry(): 3 six-point frames, frame i shifted by (i,i)
    and rotated by 15 deg * i about its centroid."""
    rot = math.radians(15.0)   # f64::to_radians == x * (PI/180); math.radians is the same product
    assert rot == 15.0 * (math.pi / 180.0)
    frames = []
    for i in range(3):
        pts = [[x, y, float(i)] for (x, y) in BASE6]
        # contour.translate_mut(i, i, 0) (not applied for i == 0 in the reference)
        if i > 0:
            pts = [[p[0] + float(i), p[1] + float(i), p[2] + 0.0] for p in pts]
        c = centroid(pts)
        if i > 0:
            pts = [rotate_pt(p, rot * float(i), c) for p in pts]   # :294, :298 (rotation.to_radians() * 2.0)
        f = Fr(i, pts, i + 1)
        f.lumen_centroid = c          # centroid was computed before the rotation
        f.centroid = list(c)
        frames.append(f)
    frames[0].ref = [3.0, 1.0, 0.0]
    return frames


def dummy_aligned_long_frames():
    """test_utils.rs:353-386 dummy_geometry_aligned_long()."""
    g1 = dummy_frames()
    rot = -15.0 * (math.pi / 180.0)
    g1[1].translate(-1.0, -1.0, 0.0)
    g1[2].translate(-2.0, -2.0, 0.0)
    g1[1].rotate(rot, (g1[1].centroid[0], g1[1].centroid[1]))
    g1[2].rotate(rot * 2.0, (g1[2].centroid[0], g1[2].centroid[1]))
    g2 = [f.clone() for f in g1]
    for i, f in enumerate(g2):
        idx = i + 3
        f.translate(0.0, 0.0, 4.0)
        # set_value(Some(idx), None, frame.lumen.centroid, Some(idx as f64))  frame.rs:69-117
        f.id = idx
        f.lumen_centroid = list(f.lumen_centroid)
        f.centroid = list(f.lumen_centroid)
        z = float(idx)
        f.pts = [[p[0], p[1], z] for p in f.pts]
        f.lumen_centroid[2] = z
        if f.ref is not None:
            f.ref[2] = z
        f.centroid[2] = z
    frames = g1 + g2
    frames[3].ref = None
    return frames


def rotate_geometry(frames, angle):
    """geometry.rs:241-250 rotate_geometry: each frame about its own centroid, then sort."""
    if angle == 0.0:
        return
    for f in frames:
        f.rotate(angle, (f.centroid[0], f.centroid[1]))
        f.sort_points()


def to_arrays(frames):
    """kwargs for FlatGeometry.from_frames / OracleGeometry.from_frames."""
    return dict(
        lumens=[np.array(f.pts, dtype=np.float64) for f in frames],
        centroids=[f.centroid for f in frames],
        ids=[f.id for f in frames],
        orig_frames=[f.orig for f in frames],
        ref_points={i: f.ref for i, f in enumerate(frames) if f.ref is not None},
    )
