"""Synthetic pullbacks for the benchmark configs (generator is ours).

Four pullbacks (rest-dia, rest-sys, stress-dia, stress-sys), each F frames x M points:
frame k lumen = ellipse a = 2.5 + 0.3 sin(k/9), b = 1.8 + 0.2 cos(k/7) mm, plus three
low-order radial harmonics (orders 2..4) whose amplitudes are N(0, 0.05^2) per pullback
with a small per-frame jitter, sampled at M equal parameter steps, centred at
(4.5, 4.5) + N(0, 0.1^2), rotated by a random-walk torsion theta_k = theta_{k-1} + N(0, 3 deg),
z = 0.5 k mm.  Reference point on frame 0; catheter circle n_points = 20, r = 0.5 around
the image centre.  PCG64 seed = 1234 + pullback id.  Coordinates land in ~2..7 mm like
examples/data/ivus_rest of the reference.

The geometry is emitted in the state ``build_geometry_from_inputdata`` leaves it in
(io/build.rs:176-197): every contour starts at its highest-y point and runs
counter-clockwise, frame 0 is the proximal end, ids are 0..F-1.
"""
from __future__ import annotations

import math
from typing import List

import numpy as np

from .geometry import FlatGeometry, catheter_points, contour_centroid

PULLBACK_LABELS = ("rest_dia", "rest_sys", "stress_dia", "stress_sys")


def _start_at_highest_y(pts: np.ndarray) -> np.ndarray:
    return np.roll(pts, -int(np.argmax(pts[:, 1])), axis=0)


def synthetic_pullback(n_frames: int, n_points: int = 501, pullback_id: int = 0, seed: int = 1234,
                       image_center=(4.5, 4.5), radius: float = 0.5, n_catheter: int = 20,
                       torsion_sigma_deg: float = 3.0) -> FlatGeometry:
    rng = np.random.Generator(np.random.PCG64(seed + pullback_id))
    t = np.arange(n_points, dtype=np.float64) * (2.0 * math.pi / n_points)
    base_amp = rng.normal(0.0, 0.05, size=(3, 2))
    theta = 0.0
    lumens, caths, cents = [], [], []
    for k in range(n_frames):
        a = 2.5 + 0.3 * math.sin(k / 9.0)
        b = 1.8 + 0.2 * math.cos(k / 7.0)
        amp = base_amp + 0.2 * rng.normal(0.0, 0.05, size=(3, 2))
        rad = np.ones_like(t)
        for h in range(3):
            rad = rad + (amp[h, 0] * np.cos((h + 2) * t) + amp[h, 1] * np.sin((h + 2) * t)) / 2.0
        x = a * np.cos(t) * rad
        y = b * np.sin(t) * rad
        if k > 0:
            theta += rng.normal(0.0, math.radians(torsion_sigma_deg))
        c, s = math.cos(theta), math.sin(theta)
        cx = image_center[0] + rng.normal(0.0, 0.1)
        cy = image_center[1] + rng.normal(0.0, 0.1)
        z = 0.5 * k
        pts = np.stack([cx + c * x - s * y, cy + s * x + c * y, np.full_like(t, z)], axis=1)
        pts = _start_at_highest_y(pts)
        lumens.append(pts)
        cents.append(contour_centroid(pts))
        caths.append(_start_at_highest_y(catheter_points(z, image_center, radius, n_catheter)))
    ref0 = lumens[0][np.argmax(lumens[0][:, 0])].copy()  # a point on frame 0 (rightmost)
    g = FlatGeometry.from_frames(lumens, catheters=caths, centroids=cents,
                                 orig_frames=np.arange(n_frames - 1, -1, -1, dtype=np.uint32),
                                 ref_points={0: ref0}, label=PULLBACK_LABELS[pullback_id % 4])
    return g


def synthetic_case(n_frames: int, n_points: int = 501, seed: int = 1234) -> List[FlatGeometry]:
    """The four pullbacks of one full (4-phase) alignment."""
    return [synthetic_pullback(n_frames, n_points, pullback_id=i, seed=seed) for i in range(4)]


def synthetic_centerline_case(n_frames: int = 24, n_points: int = 200, n_ccta: int = 4000, seed: int = 7,
                              true_rotation_deg: float = 37.0, true_index: int = 12, noise: float = 0.03,
                              clutter_frac: float = 0.0, geometry: FlatGeometry = None):
    """A centerline-placement problem with a known answer (BASELINE config 5 shape; generator is ours):
    a curved vessel centerline, one pullback geometry, and a CCTA-like point cloud sampled from that
    geometry placed on the centerline at index ``true_index`` after an in-plane rotation of
    ``true_rotation_deg``, plus noise and (optionally) off-vessel clutter.

    Returns dict(centerline, geometry, main_ref_pt, ccw_ref_pt, cw_ref_pt, points, truth=...).
    """
    from . import centerline as CL

    rng = np.random.Generator(np.random.PCG64(seed))
    if geometry is None:
        g = synthetic_pullback(n_frames, n_points, pullback_id=0, seed=seed, torsion_sigma_deg=0.5)
    else:                                   # place a given geometry (e.g. the result of from_array_*)
        g = geometry.copy()
        n_frames, n_points = g.n_frames, int(g.lumen_off[1] - g.lumen_off[0])
    if g.lumen_centroids is None:
        CL.with_lumen_centroids(g)
    # vessel path: gentle 3-D curve, z descending, raw spacing 0.2 mm (resampled to ~0.5 mm by preprocess)
    s = np.arange(0.0, 0.5 * n_frames + 30.0, 0.2)
    path = np.stack([12.0 + 6.0 * np.sin(s / 17.0), -200.0 + 5.0 * np.cos(s / 23.0), 1750.0 - 0.93 * s], axis=1)
    cl = CL.Centerline.from_contour_points(path)
    rcl, _ = CL.preprocess_centerline(cl, g)
    placed = g.copy()
    CL.rotate_geometry(placed, math.radians(true_rotation_deg))
    ref_pt = rcl.xyz()[true_index]
    CL.apply_transformations([placed], rcl, ref_pt)
    lum = placed.lumen
    pick = rng.choice(lum.shape[0], size=min(n_ccta, lum.shape[0]), replace=False)
    cloud = lum[np.sort(pick)] + rng.normal(0.0, noise, size=(pick.shape[0], 3))
    # off-vessel points (other structures of a CCTA segmentation); the bounding-box filter keeps some of them
    clutter = ref_pt + rng.normal(0.0, 30.0, size=(int(n_ccta * clutter_frac), 3))
    points = np.concatenate([cloud, clutter], axis=0)
    # landmarks: where the reference frame's points with point_index = ref, 0 and n/2 land; the
    # three-point sweep rotates without re-sorting (align_algorithms.rs:286-311), so these come from
    # the un-sorted rotation of frame 0
    ref_frame = int(np.nonzero(g.has_ref)[0][0])
    ref_index = int(np.argmin(np.linalg.norm(g.frame_lumen(ref_frame) - g.ref[ref_frame], axis=1)))
    g.meta["ref_point_index"] = ref_index
    tmp = g.copy()
    f0 = tmp.frame_lumen(ref_frame)
    c, s_ = math.cos(math.radians(true_rotation_deg)), math.sin(math.radians(true_rotation_deg))
    dx, dy = f0[:, 0] - tmp.centroids[ref_frame, 0], f0[:, 1] - tmp.centroids[ref_frame, 1]
    f0[:, 0], f0[:, 1] = dx * c - dy * s_ + tmp.centroids[ref_frame, 0], dx * s_ + dy * c + tmp.centroids[ref_frame, 1]
    CL.apply_transformations([tmp], rcl, ref_pt)
    f0 = tmp.frame_lumen(ref_frame)
    return dict(centerline=cl, geometry=g, main_ref_pt=f0[ref_index] + rng.normal(0.0, noise, 3),
                ccw_ref_pt=f0[0] + rng.normal(0.0, noise, 3), cw_ref_pt=f0[n_points // 2] + rng.normal(0.0, noise, 3),
                points=points, truth=dict(rotation_deg=true_rotation_deg, cl_index=true_index, placed=placed))


def synthetic_tube_case(n_points: int = 6000, n_reference: int = 5000, true_scaling_mm: float = 0.7,
                        seed: int = 3, noise: float = 0.02):
    """A diameter-search problem with a known answer (generator is ours): a vessel segment sampled
    as a noisy tube of radius 1.6 mm around a curved centerline, and a reference cloud sampled from
    the same tube at radius 1.6 + ``true_scaling_mm``.  Returns dict(centerline, points, reference)."""
    from . import centerline as CL

    rng = np.random.Generator(np.random.PCG64(seed))
    s = np.arange(0.0, 40.0, 0.5)
    path = np.stack([10.0 + 5.0 * np.sin(s / 15.0), -190.0 + 4.0 * np.cos(s / 19.0), 1700.0 - 0.95 * s], axis=1)
    cl = CL.Centerline.from_contour_points(path)
    t = cl.points

    def tube(n, r):
        k = rng.integers(2, len(s) - 2, size=n)
        tan = np.stack([t["tx"][k], t["ty"][k], t["tz"][k]], axis=1)
        a = np.cross(tan, np.array([1.0, 0.0, 0.0]))
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = np.cross(tan, a)
        phi = rng.uniform(0.0, 2.0 * math.pi, size=n)
        rad = r + rng.normal(0.0, noise, size=n)
        return path[k] + a * (rad * np.cos(phi))[:, None] + b * (rad * np.sin(phi))[:, None]

    return dict(centerline=cl, points=tube(n_points, 1.6), reference=tube(n_reference, 1.6 + true_scaling_mm),
                truth=true_scaling_mm)
