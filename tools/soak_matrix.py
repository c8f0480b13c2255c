"""Soak of the matrix-pipe screen (MM_PRECISION_F32_MATRIX) and of the bounded search on the matrix pipe against the packed-FMA
screen and the exact f64 kernel on random batches: set sizes 33 .. 1100 per side (every variant of k_screen_mx incl. the column
blocks, pairs below 64 points, mixed shapes in one batch; a third of the batches have one shape throughout so that the bounded
search runs its rounds), coordinate scales 1e-3 .. 1e6, offsets far from the origin, several grids.  For every batch: winners,
angles and exact costs identical between the four precisions; every screened cost of the matrix screen inside its promised
interval around the f64 cost.  Usage: python tools/soak_matrix.py [seconds] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as ge

ge.build()
import multimoda_rs_amd as mm

sys.path.insert(0, "tests")
from helpers import blob  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = mm.Engine()
eng.set_bound_min_candidates(0)
t0 = time.time()
n_batches = n_cand = 0
worst = 0.0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    P = int(rng.integers(4, 40))
    scale = float(10.0 ** rng.uniform(-3, 6))
    off = rng.uniform(-1, 1, 2) * scale * float(10.0 ** rng.uniform(0, 3))
    step, rngdeg = [(1.0, 180.0), (0.5, 180.0), (0.25, 30.0), (2.0, 90.0)][int(rng.integers(0, 4))]
    angles, _, _ = mm.search_angles(step, rngdeg)
    refs, tgts = [], []
    uniform = rng.integers(0, 3) == 0
    shape = (int(rng.integers(64, 529)),) * 2 if rng.integers(0, 2) else (int(rng.integers(64, 529)), int(rng.integers(64, 529)))
    for _ in range(P):
        if uniform:
            na, nb = shape
        elif rng.integers(0, 4) == 0:
            na, nb = int(rng.integers(33, 1101)), int(rng.integers(33, 1101))
        else:
            na, nb = int(rng.integers(449, 545)), int(rng.integers(449, 545))
        r = (blob(rng, na) - 4.5) * scale + off
        t = (blob(rng, nb) - 4.5) * scale * float(rng.uniform(0.6, 1.4)) + off
        if rng.integers(0, 5) == 0:                        # coordinates on a coarse binary grid: exact f16 ties everywhere
            g = scale / 64.0
            r = np.round(r / g) * g
            t = np.round(t / g) * g
        refs.append(r); tgts.append(t)
    cs = [t.mean(axis=0) for t in tgts]
    batch = mm.Batch(refs, tgts, [angles] * P, [(float(c[0]), float(c[1])) for c in cs])
    f = eng.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_FAST)
    x = eng.best_rotation_batch(batch, precision=mm.MM_PRECISION_F64, return_costs=True)
    m = eng.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_MATRIX, return_costs=True)
    b = eng.best_rotation_batch(batch, precision=mm.MM_PRECISION_F32_BOUNDED)
    for k in ("best_idx", "best_angle", "best_cost"):
        assert np.array_equal(f[k], x[k]) and np.array_equal(m[k], x[k]) and np.array_equal(b[k], x[k]), (seed, k)
    cx, cm = x["costs"].reshape(P, -1), m["costs"].reshape(P, -1)
    for p in range(P):
        c = cs[p]
        ra, rb = np.sqrt(((refs[p] - c) ** 2).sum(1)).max(), np.sqrt(((tgts[p] - c) ** 2).sum(1)).max()
        rho = ra + rb
        e2 = 2.0 ** -24 * (47 * rho * rho + 6 * ra * ra + 27 * rb * rb)
        if min(len(refs[p]), len(tgts[p])) < 64:
            continue                                       # not screened: every candidate scored exactly
        delta = 24 * 2.0 ** -24 * rho + 2.0 ** -49 * (abs(c).sum() + rho) + 1e-300
        S = cm[p] ** 2
        lo, hi = np.sqrt(np.maximum(0.0, S - e2)) - delta, np.sqrt(S + e2) + delta
        ok = (cx[p] >= lo) & (cx[p] <= hi)
        assert ok.all(), (seed, p, int(np.argmin(ok)))
        worst = max(worst, float((np.abs(S - cx[p] ** 2) / e2).max()))
    n_batches += 1
    n_cand += P * len(angles)
    seed += 1
print(f"MATRIX_SOAK_OK: {n_batches} batches, {n_cand} candidates in {time.time() - t0:.0f} s, seeds up to {seed - 1}; "
      f"largest |S_matrix - S_f64| / e2 = {worst:.3f}")
