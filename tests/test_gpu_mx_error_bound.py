"""Directed worst-case search for the error of the matrix-pipe screen.

The contract (include/mm_hausdorff.h, MM_PRECISION_F32_MATRIX): the screened squared value S~ of a candidate satisfies
|S~ - S| <= e2 = 2^-24 * (47 (rho_a + rho_b)^2 + 6 rho_a^2 + 27 rho_b^2) up to the rotation / input rounding that `delta` carries, i.e. the exact
cost lies in [sqrt(max(0, S~ - e2)) - delta, sqrt(S~ + e2) + delta].  A wrong constant fails SILENTLY (the true winner is
not shortlisted), and random blobs use 3 % of it.  This test goes LOOKING for large errors, in the kernel's own scaled units
(a pair is scaled by 2^e so that its larger radius lands in [256, 512), where f16 has a spacing of 0.25):
  * coordinates ON f16 rounding ties (odd multiples of 0.125 at that magnitude) and one f32 ulp either side of them,
  * radii just below the 512 edge of the scale (largest |x|, largest products) and just above 256 (smallest scale),
  * one far outlier that fixes rho while every other point is tiny (lo pieces in f16's subnormal range),
  * set sizes at the tile edges (449, 544), the smallest (64), the OCT shape (223) and two column blocks (600),
  * the candidate angles whose f32 (cos, sin) is furthest from unit norm: the target's norm is taken BEFORE the rotation,
  * configurations that make one point pair decide the Hausdorff distance, so that a single distance's error shows.
It asserts that no candidate uses more than the whole budget, and records the largest fraction seen."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U = 2.0 ** -24
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _angles_far_from_unit_norm(n=96):
    a = np.linspace(-np.pi, np.pi, 200001)[:-1]
    c, s = np.cos(a).astype(np.float32).astype(np.float64), np.sin(a).astype(np.float32).astype(np.float64)
    dev = np.abs(c * c + s * s - 1.0)
    pick = np.argsort(dev)[-n:]
    return np.sort(a[pick])


def _on_ties(rng, n, rmax, jitter):
    """n points on a rough circle of radius <= rmax (scaled units): both coordinates snapped to f16 ties at their own
    magnitude (spacing of f16 at |v|: 2^(floor(log2 |v|) - 10); a tie is an odd multiple of half of it), then moved by
    `jitter` f32 ulps."""
    t = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = rmax * (1.0 - 0.3 * rng.uniform(0, 1, n) ** 4)
    p = np.stack([r * np.cos(t), r * np.sin(t)], axis=1)
    mag = np.maximum(np.abs(p), 2.0 ** -10)
    sp = 2.0 ** (np.floor(np.log2(mag)) - 10)
    q = (np.floor(p / sp) + 0.5) * sp                                   # odd multiples of sp / 2: exactly between two f16 values
    ulp32 = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(q), 2.0 ** -100))) - 23)
    q = q + jitter * ulp32
    # keep inside the radius
    nr = np.hypot(q[:, 0], q[:, 1])
    q[nr > rmax] *= (rmax / nr[nr > rmax])[:, None] * (1 - 1e-7)
    return q


def _cases(rng):
    """(name, ref, tgt) in scaled units around the centre (0, 0)."""
    out = []
    for n in (64, 223, 449, 544, 600):
        for rmax in (511.9, 300.0, 256.01):
            for jit in (-1, 0, 1):
                out.append((f"ties n={n} r={rmax} jitter={jit}", _on_ties(rng, n, rmax, jit), _on_ties(rng, n, rmax, jit)))
        # tgt = ref moved by a fraction of the f16 spacing: tiny true distances out of large coordinates (cancellation)
        a = _on_ties(rng, n, 511.9, 0)
        out.append((f"near-identical n={n}", a, a + rng.choice([-0.0625, 0.0625, 0.03125], size=a.shape)))
        # one far outlier fixes rho; everything else tiny (lo pieces below f16's normal range after scaling)
        small = rng.normal(0, 2.0 ** -9, (n, 2))
        far = small.copy(); far[0] = (500.0, -100.0)
        out.append((f"outlier-in-ref n={n}", far, small + 2.0 ** -11))
        out.append((f"outlier-in-tgt n={n}", small + 2.0 ** -11, far))
        # one pair decides: a ring of reference points, the targets ON them except one pushed out radially
        ring = _on_ties(rng, n, 511.9, 0)
        tg = ring.copy()
        tg[n // 2] *= 0.75
        out.append((f"single-deciding-pair n={n}", ring, tg))
    return out


@pytest.mark.parametrize("scale", [1.0, 2.0 ** -7, 3.0e4])
def test_directed_search_for_the_largest_split_error(engine, oracle, mm, scale):
    rng = np.random.default_rng(20240)
    angles = np.concatenate([_angles_far_from_unit_norm(), np.linspace(-np.pi, np.pi, 33)[:-1], [0.0]])
    angles = np.sort(angles)
    worst = {"frac_of_e2": 0.0, "interval_use": 0.0}
    n_checked = 0
    for name, ref, tgt in _cases(rng):
        ref, tgt = ref * scale, tgt * scale                    # a power of two (or not): the kernel rescales to [256, 512)
        before = engine.screen_stats()
        bi, ba, bc, costs = engine.best_rotation(ref, tgt, angles, (0.0, 0.0), skip_zero=True,
                                                 precision=mm.MM_PRECISION_F32_MATRIX, return_costs=True)
        after = engine.screen_stats()
        assert (after["matrix"] + after["matrix_blocks"]) - (before["matrix"] + before["matrix_blocks"]) == len(angles), name
        oc = oracle.costs_over_angles(ref, tgt, angles, 0.0, 0.0)
        assert bi == int(np.argmin(oc)) and bc == oc[bi], name           # the winner is the oracle's, always
        ra, rb = np.hypot(ref[:, 0], ref[:, 1]).max(), np.hypot(tgt[:, 0], tgt[:, 1]).max()
        rho = ra + rb
        e2 = U * (47 * rho * rho + 6 * ra * ra + 27 * rb * rb)             # mx_e2 (csrc/mm_engine.cpp)
        delta = 24 * U * rho + 2.0 ** -49 * rho + 1e-300
        costs = np.asarray(costs)
        S = costs ** 2
        screened = costs != oc                                            # re-scored candidates carry the exact cost
        lo = np.sqrt(np.maximum(0.0, S - e2)) - delta
        hi = np.sqrt(S + e2) + delta
        assert ((oc >= lo) & (oc <= hi)).all(), name
        if screened.any():
            n_checked += int(screened.sum())
            frac = np.abs(S - oc ** 2)[screened].max() / e2
            up = np.where(oc >= costs, (oc - costs) / np.maximum(hi - costs, 1e-300), (costs - oc) / np.maximum(costs - lo, 1e-300))
            use = up[screened].max()
            if frac > worst["frac_of_e2"]:
                worst.update(frac_of_e2=float(frac), frac_case=name)
            if use > worst["interval_use"]:
                worst.update(interval_use=float(use), use_case=name)
    assert n_checked > 1000
    assert worst["interval_use"] < 1.0 and worst["frac_of_e2"] < 1.0, worst
    worst.update(scale=scale, candidates_checked=n_checked)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mx_error_bound.jsonl"), "a") as f:
        f.write(json.dumps(worst) + "\n")
    # (round 3's flat 128 u R^2: the same search found at most 2.2 % of it; the re-derived bound is 2.3 x tighter for equal
    # radii and the worst directed input uses about 5 % of it)
    assert worst["frac_of_e2"] < 0.5, worst
