"""Flat (CSR) geometry container: the host-side mirror of the reference's
``Geometry { frames: Vec<Frame> }`` (types/native/geometry.rs:9-12, frame.rs:8-15) in the
layout the C ABI consumes (``mm_geometry``, include/mm_hausdorff.h).

Points are (N, 3) f64 arrays (x, y, z); frames are CSR slices.  ``cath`` is the synthetic
catheter contour (frame.rs:163-204); ``extra`` carries every other extras contour (eem,
calcification, sidebranch, wall) concatenated per frame, with ``extra_kind`` telling which
is which, so that frame transforms move them like the reference does.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np

from ._libm import sincos

from . import _native as N


def _pts3(a) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    if a.size == 0:
        return np.zeros((0, 3), dtype=np.float64)
    if a.ndim != 2 or a.shape[1] not in (2, 3):
        raise ValueError("points must be (n, 2) or (n, 3)")
    if a.shape[1] == 2:
        a = np.concatenate([a, np.zeros((a.shape[0], 1))], axis=1)
    return np.ascontiguousarray(a)


def contour_centroid(points: np.ndarray):
    """``Contour::compute_centroid`` (contour.rs:213-224): sequential sums / n."""
    # np.add.accumulate is a strictly sequential running sum (np.sum is pairwise and rounds differently)
    tot = np.add.accumulate(np.asarray(points, dtype=np.float64)[:, :3], axis=0)[-1]
    n = float(len(points))
    return (float(tot[0]) / n, float(tot[1]) / n, float(tot[2]) / n)


def catheter_points(z: float, image_center=(4.5, 4.5), radius=0.5, n_points=20) -> np.ndarray:
    """``Frame::create_catheter_points`` for one frame (frame.rs:189-202)."""
    out = np.empty((n_points, 3), dtype=np.float64)
    for i in range(n_points):
        angle = 2.0 * math.pi * float(i) / float(n_points)
        si, co = sincos(angle)                      # frame.rs:192-193: cos and sin of one value
        out[i, 0] = image_center[0] + radius * co
        out[i, 1] = image_center[1] + radius * si
        out[i, 2] = z
    return out


class AlignLogs(Sequence):
    """The AlignLog records of one pullback (align_within.rs:14-22) as the reference returns them to
    Python: a sequence of 7-tuples ``(id, matched_to, rot_deg, tx, ty, cx, cy)``
    (binding/functions.rs:26-40).  The tuples are materialised on first access; the C buffer the
    host code wrote is kept as is until then."""

    def __init__(self, buf, n: int):
        self._buf, self._n, self._list = buf, n, None

    def tolist(self):
        if self._list is None:
            b = self._buf
            self._list = [(b[i].contour_id, b[i].matched_to, b[i].rot_deg, b[i].tx, b[i].ty, b[i].cx, b[i].cy)
                          for i in range(self._n)]
        return self._list

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self.tolist()[i]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, AlignLogs) else list(other))

    def __repr__(self):
        return repr(self.tolist())


@dataclass
class FlatGeometry:
    ids: np.ndarray                      # (F,) u32  Frame.id
    lumen_ids: np.ndarray                # (F,) u32  Frame.lumen.id
    orig_frames: np.ndarray              # (F,) u32  Frame.lumen.original_frame
    centroids: np.ndarray                # (F,3) f64 Frame.centroid
    lumen_off: np.ndarray                # (F+1,) i64
    lumen: np.ndarray                    # (N,3) f64
    cath_off: Optional[np.ndarray] = None
    cath: Optional[np.ndarray] = None
    extra_off: Optional[np.ndarray] = None
    extra: Optional[np.ndarray] = None
    has_ref: Optional[np.ndarray] = None  # (F,) u8
    ref: Optional[np.ndarray] = None      # (F,3) f64
    label: str = ""
    meta: dict = field(default_factory=dict)   # host-only bookkeeping (extras layout, wall thickness records)
    # Frame.lumen.centroid (Contour.centroid): read by the centerline placement only (centerline.py);
    # None = every contour centroid is None and align_frame falls back to the mean of the points
    has_lumen_centroid: Optional[np.ndarray] = None   # (F,) u8
    lumen_centroids: Optional[np.ndarray] = None      # (F,3) f64

    # ------------------------------------------------------------------------------
    @property
    def n_frames(self) -> int:
        return int(self.ids.shape[0])

    @staticmethod
    def from_frames(lumens: Sequence[np.ndarray], catheters: Optional[Sequence[np.ndarray]] = None,
                    centroids=None, ids=None, orig_frames=None, ref_points: Optional[Dict[int, Sequence[float]]] = None,
                    label: str = "") -> "FlatGeometry":
        F = len(lumens)
        lum = [_pts3(l) for l in lumens]
        off = np.zeros(F + 1, dtype=np.int64)
        if F:
            off[1:] = np.cumsum([l.shape[0] for l in lum])
        if centroids is None:
            cen = np.array([contour_centroid(l) for l in lum], dtype=np.float64).reshape(F, 3)
        else:
            cen = np.array(centroids, dtype=np.float64).reshape(F, 3).copy()
        idv = np.arange(F, dtype=np.uint32) if ids is None else np.asarray(ids, dtype=np.uint32).copy()
        g = FlatGeometry(
            ids=idv,
            lumen_ids=idv.copy(),
            orig_frames=(np.arange(F, dtype=np.uint32) if orig_frames is None
                         else np.asarray(orig_frames, dtype=np.uint32).copy()),
            centroids=np.ascontiguousarray(cen),
            lumen_off=off,
            lumen=np.ascontiguousarray(np.concatenate(lum, axis=0)) if F else np.zeros((0, 3)),
            label=label,
        )
        if catheters is not None:
            cat = [_pts3(c) for c in catheters]
            coff = np.zeros(F + 1, dtype=np.int64)
            coff[1:] = np.cumsum([c.shape[0] for c in cat])
            g.cath_off = coff
            g.cath = np.ascontiguousarray(np.concatenate(cat, axis=0))
        g.has_ref = np.zeros(F, dtype=np.uint8)
        g.ref = np.zeros((F, 3), dtype=np.float64)
        if ref_points:
            for i, p in ref_points.items():
                g.has_ref[i] = 1
                g.ref[i] = np.asarray(p, dtype=np.float64)
        return g

    def copy(self) -> "FlatGeometry":
        cp = lambda a: None if a is None else a.copy()
        return FlatGeometry(cp(self.ids), cp(self.lumen_ids), cp(self.orig_frames), cp(self.centroids),
                            cp(self.lumen_off), cp(self.lumen), cp(self.cath_off), cp(self.cath),
                            cp(self.extra_off), cp(self.extra), cp(self.has_ref), cp(self.ref), self.label,
                            dict(self.meta), cp(self.has_lumen_centroid), cp(self.lumen_centroids))

    def frame_lumen(self, i: int) -> np.ndarray:
        return self.lumen[self.lumen_off[i]:self.lumen_off[i + 1]]

    def frame_cath(self, i: int) -> np.ndarray:
        return self.cath[self.cath_off[i]:self.cath_off[i + 1]]

    # ------------------------------------------------------------------------------
    def _validate(self):
        F = self.n_frames
        for name, dt, shape in (("ids", np.uint32, (F,)), ("lumen_ids", np.uint32, (F,)),
                                ("orig_frames", np.uint32, (F,)), ("centroids", np.float64, (F, 3)),
                                ("lumen_off", np.int64, (F + 1,))):
            a = getattr(self, name)
            if a.dtype != dt or a.shape != shape or not a.flags.c_contiguous:
                raise ValueError(f"FlatGeometry.{name}: expected C-contiguous {dt.__name__}{shape}")
        if self.lumen.dtype != np.float64 or self.lumen.ndim != 2 or self.lumen.shape[1] != 3 \
                or not self.lumen.flags.c_contiguous or self.lumen.shape[0] != int(self.lumen_off[-1] if F else 0):
            raise ValueError("FlatGeometry.lumen: expected C-contiguous float64 (N,3) matching lumen_off")
        for off, arr, nm in ((self.cath_off, self.cath, "cath"), (self.extra_off, self.extra, "extra")):
            if off is not None:
                if off.dtype != np.int64 or off.shape != (F + 1,) or arr is None or arr.dtype != np.float64 \
                        or not arr.flags.c_contiguous or arr.shape != (int(off[-1]), 3):
                    raise ValueError(f"FlatGeometry.{nm}: inconsistent CSR arrays")
        if self.has_ref is not None:
            if self.has_ref.dtype != np.uint8 or self.has_ref.shape != (F,) or self.ref is None \
                    or self.ref.shape != (F, 3) or self.ref.dtype != np.float64:
                raise ValueError("FlatGeometry.has_ref/ref: inconsistent arrays")

    def c_struct(self) -> N.MMGeometry:
        """Borrowed view for the C ABI (arrays stay owned by this object)."""
        self._validate()
        F = self.n_frames
        g = N.MMGeometry()
        p = N._ptr
        g.n_frames = self.n_frames
        g.id = p(self.ids); g.lumen_id = p(self.lumen_ids); g.orig_frame = p(self.orig_frames)
        g.centroid = p(self.centroids)
        g.lumen_off = p(self.lumen_off); g.lumen = p(self.lumen)
        g.has_catheter = 1 if self.cath_off is not None else 0
        g.cath_off = p(self.cath_off); g.cath = p(self.cath)
        g.extra_off = p(self.extra_off); g.extra = p(self.extra)
        g.has_ref = p(self.has_ref); g.ref = p(self.ref)
        # Frame.lumen.centroid, tracked through translations when every frame carries one
        lc = self.lumen_centroids
        if lc is not None and (self.has_lumen_centroid is None or bool(np.all(self.has_lumen_centroid))):
            if lc.dtype != np.float64 or lc.shape != (F, 3) or not lc.flags.c_contiguous:
                raise ValueError("FlatGeometry.lumen_centroids: expected C-contiguous float64 (F,3)")
            g.lumen_centroid = p(lc)
        return g


# --------------------------------------------------------------------------------------
# drivers (host orchestration lives in C++: csrc/mm_host.cpp)
# --------------------------------------------------------------------------------------
def align_within(engine: N.Engine, geoms: Sequence[FlatGeometry], step_deg: float, range_deg: float,
                 bruteforce: bool, sample_size: int, precision: int = N.MM_PRECISION_F32_MATRIX, mode: int = 0):
    """``align_frames_in_geometry`` lines 24-134 (align_within.rs) for several pullbacks in
    lockstep; geometries are updated in place.  Returns (logs per geometry as 7-tuples
    ``(id, matched_to, rot_deg, tx, ty, cx, cy)`` -- binding/functions.rs:26-40, pose_evals)."""
    G = len(geoms)
    structs = [g.c_struct() for g in geoms]
    gptrs = (C.POINTER(N.MMGeometry) * G)(*[C.pointer(s) for s in structs])
    log_bufs = [(N.MMAlignLog * max(g.n_frames - 1, 1))() for g in geoms]
    lptrs = (C.c_void_p * G)(*[C.cast(b, C.c_void_p) for b in log_bufs])
    pe = C.c_int64(0)
    N.check(N.lib().mm_align_within(engine.handle, G, C.cast(gptrs, C.c_void_p), float(step_deg), float(range_deg),
                                    int(bool(bruteforce)), int(sample_size), int(precision), int(mode),
                                    C.cast(lptrs, C.c_void_p), C.byref(pe)), "mm_align_within")
    logs = [AlignLogs(b, g.n_frames - 1) for b, g in zip(log_bufs, geoms)]
    return logs, int(pe.value)


class WithinPlan:
    """Decoupled within-pullback alignment with the point sets staged in HBM up front
    (``mm_within_plan_*``): ``WithinPlan(...)`` stages, ``run()`` searches + walks the chain.
    Same results as :func:`align_within`; a plan runs once."""

    def __init__(self, engine: N.Engine, geoms: Sequence[FlatGeometry], step_deg: float, range_deg: float,
                 bruteforce: bool, sample_size: int, precision: int = N.MM_PRECISION_F32_MATRIX, shard=None):
        """shard = (rank, world): this plan owns the share [n*rank/world, n*(rank+1)/world) of every candidate
        list from the start (same as set_shard afterwards, without staging level 0 twice).
        shard = (rank, pair_blocks, cand_slices): a tile of the (frame pair x candidate) grid
        (``mm_within_plan_create_grid``; distributed.shard_grid gives the default shape)."""
        self.engine = engine
        self.geoms = list(geoms)
        G = len(self.geoms)
        self._structs = [g.c_struct() for g in self.geoms]
        self._gptrs = (C.POINTER(N.MMGeometry) * G)(*[C.pointer(s) for s in self._structs])
        self._h = C.c_void_p()
        engine._children.add(self)
        if shard is None:
            rank, pb, cs = 0, 1, 1
        elif len(shard) == 2:
            rank, pb, cs = int(shard[0]), 1, int(shard[1])
        else:
            rank, pb, cs = int(shard[0]), int(shard[1]), int(shard[2])
        self.shard = (rank, pb, cs)
        N.check(N.lib().mm_within_plan_create_grid(engine.handle, G, C.cast(self._gptrs, C.c_void_p), float(step_deg),
                                                   float(range_deg), int(bool(bruteforce)), int(sample_size),
                                                   int(precision), rank, pb, cs, C.byref(self._h)),
                "mm_within_plan_create")

    def run(self):
        """Returns (logs per geometry, pose_evals, n_unresolved)."""
        G = len(self.geoms)
        log_bufs = [(N.MMAlignLog * max(g.n_frames - 1, 1))() for g in self.geoms]
        lptrs = (C.c_void_p * G)(*[C.cast(b, C.c_void_p) for b in log_bufs])
        pe, nu = C.c_int64(0), C.c_int64(0)
        N.check(N.lib().mm_within_plan_run(self._h, C.cast(lptrs, C.c_void_p), C.byref(pe), C.byref(nu)),
                "mm_within_plan_run")
        logs = [AlignLogs(b, g.n_frames - 1) for b, g in zip(log_bufs, self.geoms)]
        return logs, int(pe.value), int(nu.value)

    def fetch_set(self, set_index: int):
        """Diagnostics: search set `set_index` as staged in HBM -> (xy f64 (n,2), xy f32 (n,2), rho)."""
        rho = C.c_double(0.0)
        n = N.lib().mm_within_plan_fetch_set(self._h, int(set_index), None, None, None, None, 0, C.byref(rho))
        if n < 0:
            raise RuntimeError(N.last_error())
        a64 = np.zeros((2, int(n)), dtype=np.float64)
        a32 = np.zeros((2, int(n)), dtype=np.float32)
        N.lib().mm_within_plan_fetch_set(self._h, int(set_index), N._ptr(a64[0]), N._ptr(a64[1]), N._ptr(a32[0]),
                                         N._ptr(a32[1]), int(n), C.byref(rho))
        return a64.T.copy(), a32.T.copy(), rho.value

    # -- candidate axis sharded over ranks (one process per GPU) ---------------------------
    def set_shard(self, rank: int, world: int):
        N.check(N.lib().mm_within_plan_set_shard(self._h, int(rank), int(world)), "mm_within_plan_set_shard")
        self.shard = (int(rank), 1, int(world))

    def set_shard_grid(self, rank: int, pair_blocks: int, cand_slices: int):
        N.check(N.lib().mm_within_plan_set_shard_grid(self._h, int(rank), int(pair_blocks), int(cand_slices)),
                "mm_within_plan_set_shard_grid")
        self.shard = (int(rank), int(pair_blocks), int(cand_slices))

    def set_timing_rehearsal(self, on: bool = True):
        """TIMING ONLY (``mm_within_plan_set_timing_rehearsal``): let a world = 1 communicator serve this plan's tile of a
        larger grid; the result is then not an alignment and ``walk`` reports ``n_unresolved == -1``."""
        N.check(N.lib().mm_within_plan_set_timing_rehearsal(self._h, int(bool(on))), "mm_within_plan_set_timing_rehearsal")

    def search_sharded_begin(self, comm: "N.Comm"):
        """Level 0 of the sharded search enqueued up to the copy of the reduced records, nothing waited for
        (``mm_within_plan_search_sharded_begin``); ``search_sharded`` collects."""
        N.check(N.lib().mm_within_plan_search_sharded_begin(self._h, comm.handle), "mm_within_plan_search_sharded_begin")

    def search_sharded(self, comm: "N.Comm"):
        """The sharded search with the exchange inside the library: per level launch -> export -> ncclAllReduce(MIN)
        x 2 on the engine's stream -> commit (``mm_within_plan_search_sharded``)."""
        N.check(N.lib().mm_within_plan_search_sharded(self._h, comm.handle), "mm_within_plan_search_sharded")
        self._begun = False

    def staged(self):
        """(raw contour points copied to the device, points in the plan's set pool) -- ``mm_within_plan_staged``: a plan
        created on a tile with several pair blocks stages its own block's frames only."""
        raw, pts = C.c_int64(0), C.c_int64(0)
        N.check(N.lib().mm_within_plan_staged(self._h, C.byref(raw), C.byref(pts)), "mm_within_plan_staged")
        return int(raw.value), int(pts.value)

    def dims(self):
        """(n_jobs, n_levels, per-job tie tolerance; 0 for the jobs of another pair block on a plan that staged only its
        own block's frames)."""
        nj, nl = C.c_int32(0), C.c_int32(0)
        N.check(N.lib().mm_within_plan_dims(self._h, C.byref(nj), C.byref(nl), None), "mm_within_plan_dims")
        tol = np.zeros(nj.value, dtype=np.float64)
        N.check(N.lib().mm_within_plan_dims(self._h, None, None, N._ptr(tol)), "mm_within_plan_dims")
        return int(nj.value), int(nl.value), tol

    def level_local(self, level: int, n_jobs: int):
        out = {"cost": np.zeros(n_jobs), "uniform": np.zeros(n_jobs, dtype=np.int32), "angle": np.zeros(n_jobs),
               "idx": np.zeros(n_jobs, dtype=np.int32), "active": np.zeros(n_jobs, dtype=np.int32)}
        N.check(N.lib().mm_within_plan_level_local(self._h, int(level), N._ptr(out["cost"]), N._ptr(out["uniform"]),
                                                   N._ptr(out["angle"]), N._ptr(out["idx"]), N._ptr(out["active"])),
                "mm_within_plan_level_local")
        return out

    def level_collect(self, level: int, n_jobs: int):
        """The fetch half of level_local (after level_launch)."""
        out = {"cost": np.zeros(n_jobs), "uniform": np.zeros(n_jobs, dtype=np.int32), "angle": np.zeros(n_jobs),
               "idx": np.zeros(n_jobs, dtype=np.int32), "active": np.zeros(n_jobs, dtype=np.int32)}
        N.check(N.lib().mm_within_plan_level_collect(self._h, int(level), N._ptr(out["cost"]), N._ptr(out["uniform"]),
                                                     N._ptr(out["angle"]), N._ptr(out["idx"]), N._ptr(out["active"])),
                "mm_within_plan_level_collect")
        return out

    def search_begin(self, after: Optional[N.Engine] = None):
        """Single rank: enqueue level 0 of the search and return at once; ``search_end`` collects it and runs the
        remaining levels.  after = an engine whose current search this one's launch should follow on the device
        (``Engine.wait_search``): the launch then starts when that one's long kernel ends, beside its short tail."""
        if after is not None and after is not self.engine:
            self.engine.wait_search(after)
        self.level_launch(0)
        self._begun = True

    def search_end(self):
        from . import distributed as D
        n_jobs, n_levels, tol = self.dims()
        for l in range(n_levels):
            local = self.level_collect(l, n_jobs) if (l == 0 and getattr(self, "_begun", False)) else self.level_local(l, n_jobs)
            ok, angle, _idx, _cost = D.merge_shards(1, local["cost"], local["uniform"], local["angle"], local["idx"], tol)
            self.level_commit(l, ok, angle)
        self._begun = False

    def level_commit(self, level: int, ok: np.ndarray, angle: np.ndarray):
        ok = np.ascontiguousarray(ok, dtype=np.uint8)
        angle = np.ascontiguousarray(angle, dtype=np.float64)
        N.check(N.lib().mm_within_plan_level_commit(self._h, int(level), N._ptr(ok), N._ptr(angle)),
                "mm_within_plan_level_commit")

    # -- the exchange on the device (mm_within_plan_level_launch / export_* / commit_dev) --
    def level_launch(self, level: int):
        N.check(N.lib().mm_within_plan_level_launch(self._h, int(level)), "mm_within_plan_level_launch")

    def level_export_cost(self, level: int, cost_dev: int):
        """cost_dev: device address of n_jobs float64."""
        N.check(N.lib().mm_within_plan_level_export_cost(self._h, int(level), C.c_void_p(cost_dev)),
                "mm_within_plan_level_export_cost")

    def level_export_keys(self, level: int, gcost_dev: int, keys_dev: int):
        """gcost_dev: the all-reduced costs; keys_dev: device address of 3 * n_jobs int64."""
        N.check(N.lib().mm_within_plan_level_export_keys(self._h, int(level), C.c_void_p(gcost_dev), C.c_void_p(keys_dev)),
                "mm_within_plan_level_export_keys")

    def level_commit_dev(self, level: int, gcost_dev: int, keys_dev: int):
        N.check(N.lib().mm_within_plan_level_commit_dev(self._h, int(level), C.c_void_p(gcost_dev), C.c_void_p(keys_dev)),
                "mm_within_plan_level_commit_dev")

    def walk(self, take=None):
        """The chain walk (logs per geometry, pose_evals, n_unresolved).  take: a bool per pullback -- walk only those
        (``mm_within_plan_walk_geoms``; the others stay untouched and their logs unwritten: distributed.walk_sharded)."""
        G = len(self.geoms)
        log_bufs = [(N.MMAlignLog * max(g.n_frames - 1, 1))() for g in self.geoms]
        lptrs = (C.c_void_p * G)(*[C.cast(b, C.c_void_p) for b in log_bufs])
        pe, nu = C.c_int64(0), C.c_int64(0)
        if take is not None:
            mask = (C.c_uint8 * G)(*[1 if t else 0 for t in take])
            N.check(N.lib().mm_within_plan_walk_geoms(self._h, C.cast(mask, C.c_void_p), C.cast(lptrs, C.c_void_p), C.byref(pe),
                                                      C.byref(nu)), "mm_within_plan_walk_geoms")
        else:
            N.check(N.lib().mm_within_plan_walk(self._h, C.cast(lptrs, C.c_void_p), C.byref(pe), C.byref(nu)),
                    "mm_within_plan_walk")
        logs = [AlignLogs(b, g.n_frames - 1) for b, g in zip(log_bufs, self.geoms)]
        return logs, int(pe.value), int(nu.value)

    def search(self, group=None):
        """The search half of run(): per level, local search (this rank's share of the candidate axis if
        set_shard was called) -> exchange over the torch.distributed group -> merge -> commit.  Without a
        process group this is the single-rank search.  walk() is the other half; a driver may overlap it
        with the search of the next, independent case (bench.py)."""
        from . import distributed as D
        if D.world_size(group) > 1:
            mode = D.exchange_mode(group)
            if mode in ("rccl", "device") and D.first_search_pending(group):
                return D.search_checked(self, group, mode)
            if mode == "rccl":
                return self.search_sharded(D.native_comm(group))
            if mode == "device":
                return D.search_device(self, group)
        n_jobs, n_levels, tol = self.dims()
        begun, self._begun = bool(getattr(self, "_begun", False)), False      # search_begin enqueued level 0 already
        for l in range(n_levels):
            local = self.level_collect(l, n_jobs) if (begun and l == 0) else self.level_local(l, n_jobs)
            ok, angle, _idx, _cost = D.merge_level(local, tol, group)
            self.level_commit(l, ok, angle)

    def run_sharded(self, group=None):
        """run() with the candidate axis sharded over the ranks of a torch.distributed group
        (set_shard must have been called); every rank then walks the chain.  Same return value as run()."""
        self.search(group)
        return self.walk()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            N.lib().mm_within_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_between(engine: N.Engine, pairs: Sequence[Sequence[FlatGeometry]], rot_deg: float, step_rot_deg: float,
                  sample_size: int, precision: int = N.MM_PRECISION_F32_MATRIX):
    """``align_between_geometries`` (align_between.rs:11-68) for independent (a, b) pairs; every
    b is moved onto its a in place.  Returns (best rotations in radians, pose_evals)."""
    P = len(pairs)
    sa = [a.c_struct() for a, _ in pairs]
    sb = [b.c_struct() for _, b in pairs]
    pa = (C.POINTER(N.MMGeometry) * P)(*[C.pointer(s) for s in sa])
    pb = (C.POINTER(N.MMGeometry) * P)(*[C.pointer(s) for s in sb])
    best = np.zeros(P, dtype=np.float64)
    pe = C.c_int64(0)
    N.check(N.lib().mm_align_between(engine.handle, P, C.cast(pa, C.c_void_p), C.cast(pb, C.c_void_p), float(rot_deg),
                                     float(step_rot_deg), int(sample_size), int(precision), N._ptr(best),
                                     C.byref(pe)), "mm_align_between")
    return best, int(pe.value)


def search_set(geom: FlatGeometry, frame: int, sample_size: int) -> np.ndarray:
    """The point set the chain builds for one frame (align_within.rs:45-59,173-191), (n,2)."""
    s = geom.c_struct()
    cap = int(geom.lumen_off[frame + 1] - geom.lumen_off[frame])
    if geom.cath_off is not None:
        cap += int(geom.cath_off[frame + 1] - geom.cath_off[frame])
    x = np.empty(cap + 1, dtype=np.float64)
    y = np.empty(cap + 1, dtype=np.float64)
    n = N.lib().mm_catheter_lumen_vec(C.byref(s), frame, int(sample_size), N._ptr(x), N._ptr(y), cap + 1)
    return np.stack([x[:n], y[:n]], axis=1)


def between_points(geom: FlatGeometry, sample_size: int) -> np.ndarray:
    """``extract_geometry_points_with_frame_info`` (align_between.rs:154-178), (n,2)."""
    s = geom.c_struct()
    n = N.lib().mm_extract_between_points(C.byref(s), int(sample_size), None, None, 0)
    x = np.empty(int(n), dtype=np.float64)
    y = np.empty(int(n), dtype=np.float64)
    N.lib().mm_extract_between_points(C.byref(s), int(sample_size), N._ptr(x), N._ptr(y), int(n))
    return np.stack([x, y], axis=1)
