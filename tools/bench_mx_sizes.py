"""Kernel time of the brute-force screens across set sizes: the matrix-pipe screen (k_screen_mx<NCT, MULTI>)
against the packed-FMA screen (k_screen_fast) on batches of P pairs x 361 rotations, N points per set, the batch sized so
that every case is ~2e11 pair-distances.  hipEvents around the screen launches (mm_engine_profile).  Prints one JSON line.
Also the single-search shape of ADVICE r3 (one pair, 361 candidates)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multimoda_rs_amd as mm  # noqa: E402
from helpers import blob  # noqa: E402


def run(eng, batch, prec, reps=3):
    best = None
    for _ in range(reps):
        eng.profile(True)
        t0 = time.perf_counter()
        out = eng.best_rotation_batch(batch, precision=prec)
        wall = time.perf_counter() - t0
        pr = eng.profile_read()
        eng.profile(False)
        if best is None or pr["ms"] < best[0]:
            best = (pr["ms"], wall * 1e3, pr["candidates"], out)
    return best


def main():
    sizes = [int(x) for x in (sys.argv[1:] or "64 128 208 240 320 448 521 600 1042 2048".split())]
    rng = np.random.default_rng(3)
    angles, _, _ = mm.search_angles(1.0, 180.0)
    rows = []
    with mm.Engine(0) as eng:
        for n in sizes:
            P = max(4, min(2044, int(2e11 / (2.0 * n * n * len(angles)))))
            refs = [blob(rng, n) for _ in range(P)]
            tgts = [blob(rng, n) for _ in range(P)]
            cs = [t.mean(axis=0) for t in tgts]
            batch = mm.Batch(refs, tgts, [angles] * P, [(float(c[0]), float(c[1])) for c in cs])
            f = run(eng, batch, mm.MM_PRECISION_F32_FAST)
            m = run(eng, batch, mm.MM_PRECISION_F32_MATRIX)
            same = bool(np.array_equal(f[3]["best_idx"], m[3]["best_idx"]) and np.array_equal(f[3]["best_cost"], m[3]["best_cost"]))
            tiles = ((n + 31) // 32) ** 2
            rows.append({"points": n, "pairs": P, "candidates": P * len(angles), "fast_ms": f[0], "matrix_ms": m[0], "speedup": f[0] / m[0],
                         "matrix_ns_per_tile_per_simd": m[0] * 1e6 / (P * len(angles) * tiles / 1024.0), "identical_winners": same})
            print(rows[-1], file=sys.stderr)
        # one search alone (361 candidates, one pair)
        ref, tgt = blob(rng, 521), blob(rng, 521)
        c = tgt.mean(axis=0)
        one = mm.Batch([ref], [tgt], [angles], [(float(c[0]), float(c[1]))])
        single = {}
        for name, prec in (("fast", mm.MM_PRECISION_F32_FAST), ("matrix", mm.MM_PRECISION_F32_MATRIX)):
            r = run(eng, one, prec, reps=10)
            single[name] = {"kernel_ms": r[0], "call_ms": r[1]}
    print(json.dumps({"sizes": rows, "single_search_361_candidates_521_points": single}))


if __name__ == "__main__":
    main()
