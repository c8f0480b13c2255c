"""Shortlist sizes of the screens: candidates per search that the exact f64 kernel re-scores (their interval
[sqrt(S - e2) - delta, sqrt(S + e2) + delta] reaches below the smallest upper bound), on the frame pairs of one synthetic
config3 pullback (511 pairs x 721 rotations, N = 521) and of the OCT-shaped pullback (279 pairs x 1201, N = 223).
The matrix-pipe screen's e2 is the knob VERDICT r3 #4 asks about.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimoda_rs_amd as mm  # noqa: E402
import bench  # noqa: E402


def pairs_of(g, ss):
    sets = [mm.search_set(g, i, ss) for i in range(g.n_frames)]
    refs, tgts, cs = [], [], []
    for i in range(1, g.n_frames):
        # the decoupled search: both frames centred on their own centroids
        refs.append(sets[i - 1] - g.centroids[i - 1, :2]); tgts.append(sets[i] - g.centroids[i, :2]); cs.append((0.0, 0.0))
    return refs, tgts, cs


def main():
    out = {}
    with mm.Engine(0) as eng:
        for name, g, ss, step, rng_deg in (("config3_pullback0", mm.synthetic_pullback(512, 501), 501, 0.5, 180.0),
                                           ("oct_single", bench.oct_pullback(mm), 200, 0.01, 6.0)):
            refs, tgts, cs = pairs_of(g, ss)
            angles, _, _ = mm.search_angles(step, rng_deg)
            batch = mm.Batch(refs, tgts, [angles] * len(refs), cs)
            row = {"pairs": len(refs), "candidates_per_pair": len(angles)}
            for pname, prec in (("direct_f32", mm.MM_PRECISION_F32), ("packed_fma", mm.MM_PRECISION_F32_FAST), ("matrix", mm.MM_PRECISION_F32_MATRIX)):
                r = eng.best_rotation_batch(batch, precision=prec)
                n = np.asarray(r["n_rescored"])
                row[pname] = {"mean": float(n.mean()), "median": float(np.median(n)), "max": int(n.max()), "total": int(n.sum())}
            out[name] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
