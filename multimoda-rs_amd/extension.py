"""EXTENSION axis, absent from the reference's 4-phase path:
a (rotation x frame-shift) grid per frame.

The reference compares frame i only with frame i-1 (align_within.rs:72-134); BASELINE.json's
configs name a "1-frame-shift grid (~720 x 100 candidates)".  It is honoured here as extra
(reference set, target set) pairs through the SAME batched primitive: target = frame i,
reference = frame i-1+s for every shift s of a window, each frame centred on its own centroid
like the decoupled within-pullback search.  At s = 0 a pair is exactly a step of the reference's
chain (up to the rigid motion the chain applies to both frames); for s != 0 parity is against the
CPU oracle's hausdorff/search semantics only.  Results of this module are never folded into the
reference-equivalent numbers.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from . import _native as N
from . import geometry as G


class ShiftRotationSearch:
    """Stage once (point sets resident in HBM), run many times."""

    def __init__(self, engine: N.Engine, geoms: Sequence[G.FlatGeometry], shift_lo: int, shift_hi: int,
                 step_deg: float, range_deg: float, sample_size: int, precision: int = N.MM_PRECISION_F32_MATRIX,
                 want_costs: bool = False):
        self.angles, deg, _ = N.search_angles(step_deg, range_deg)
        if deg:
            raise ValueError("degenerate candidate list")
        sets, centres, frame_of_set = [], [], []
        ref_set, tgt_set, meta = [], [], []
        base = 0
        for gi, g in enumerate(geoms):
            for i in range(g.n_frames):
                s = G.search_set(g, i, sample_size) - g.centroids[i, :2]     # centred on its own centroid
                sets.append(s); centres.append((0.0, 0.0)); frame_of_set.append((gi, i))
            for i in range(g.n_frames):
                for sh in range(shift_lo, shift_hi + 1):
                    j = i - 1 + sh
                    if 0 <= j < g.n_frames and j != i:
                        ref_set.append(base + j); tgt_set.append(base + i); meta.append((gi, i, sh, j))
            base += g.n_frames
        self.meta = np.array(meta, dtype=np.int32).reshape(-1, 4)           # (pullback, frame, shift, ref frame)
        self.batch = N.IndexedBatch(sets, centres, ref_set, tgt_set, self.angles,
                                    flags=np.full(len(ref_set), N.MM_SEARCH_SKIP_ZERO, dtype=np.int32))
        self.plan = N.Plan(engine, self.batch, precision, want_costs=want_costs)
        self.want_costs = want_costs

    @property
    def pose_evals(self) -> int:
        return int(self.batch.n_pairs) * len(self.angles)

    def run(self):
        """Returns per pair (best index, angle, exact cost) and, per (pullback, frame), the winning
        (shift, angle, cost): first minimum in (shift ascending, angle ascending) order."""
        self.plan.run()
        res = self.plan.fetch(return_costs=self.want_costs)
        m, cost = self.meta, res["best_cost"]
        key = m[:, 0].astype(np.int64) * (1 << 32) + m[:, 1]
        order = np.lexsort((m[:, 2], key))                   # groups by (pullback, frame), shift ascending
        ks = key[order]
        starts = np.concatenate([[0], np.nonzero(np.diff(ks))[0] + 1])
        winners = []
        for a, b in zip(starts, np.concatenate([starts[1:], [len(order)]])):
            seg = order[a:b]
            k = seg[int(np.argmin(cost[seg]))]               # argmin keeps the first (lowest shift) minimum
            winners.append((int(m[k, 0]), int(m[k, 1]), int(m[k, 2]), float(res["best_angle"][k]), float(cost[k])))
        res["winners"] = winners
        return res

    def close(self):
        self.plan.close()
