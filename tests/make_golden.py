"""Generate tests/golden/*.json: expected outputs of the CPU oracle (oracle/) on the data
fixtures of the reference's own tests (tests/golden/ivus_rest, ivus_stress, idealized_geometry).

The oracle's behaviour is pinned by the reference's known-answer tests (test_oracle_kat.py);
the reference itself cannot be built here, so these vectors are "oracle-generated, not
reference-verified".  The oracle's INPUT geometries come from tests/refbuild.py, an independent
restatement of the reference's CSV reader and geometry builder -- not from the product's
multimoda_rs_amd.io, so a builder bug cannot cancel out (tests/test_refbuild.py compares the two).
Floats are stored as hex strings (bit-exact).  Run:
    python tests/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc
import refbuild                          # independent pure-Python restatement of the reference's reader + builder

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [
    # (name, folder, diastole, step, range, bruteforce, sample_size)
    ("ivus_rest_dia_brute_2deg", "ivus_rest", True, 2.0, 90.0, True, 500),
    ("ivus_rest_sys_hier_0p5deg", "ivus_rest", False, 0.5, 90.0, False, 500),
    ("ivus_stress_dia_hier_0p05deg", "ivus_stress", True, 0.05, 45.0, False, 200),
    ("idealized_dia_hier_0p01deg", "idealized_geometry", True, 0.01, 20.0, False, 200),
]


def hexf(x):
    return float(x).hex()


def main():
    out = {}
    for name, folder, dia, step, rng, brute, ss in CASES:
        og = g = refbuild.oracle_geometry(orc, os.path.join(GOLD, folder), dia, folder)
        logs = orc.align_within_chain(og, step, rng, brute, ss, n_threads=8)
        out[name] = {
            "folder": folder, "diastole": dia, "step_deg": step, "range_deg": rng, "bruteforce": brute,
            "sample_size": ss, "n_frames": g.n_frames,
            "logs": [[int(l[0]), int(l[1])] + [hexf(v) for v in l[2:]] for l in logs],
            "lumen_sum_hex": [hexf(np.sum(og.lumen[:, 0])), hexf(np.sum(og.lumen[:, 1]))],
            "first_points_hex": [[hexf(v) for v in og.frame_lumen(i)[0]] for i in range(g.n_frames)],
        }
        print(name, g.n_frames, "frames; first rot_deg", logs[0][2])
    # between: rest diastole vs systole after their chains
    oa = refbuild.oracle_geometry(orc, os.path.join(GOLD, "ivus_rest"), True, "rest")
    ob = gb = refbuild.oracle_geometry(orc, os.path.join(GOLD, "ivus_rest"), False, "rest")
    orc.align_within_chain(oa, 1.0, 90.0, False, 500, n_threads=8)
    orc.align_within_chain(ob, 1.0, 90.0, False, 500, n_threads=8)
    best = orc.align_between(oa, ob, 90.0, 0.5, 500, n_threads=8)
    out["ivus_rest_between_chain_only"] = {"best_rotation_hex": hexf(best),
                                           "b_first_points_hex": [[hexf(v) for v in ob.frame_lumen(i)[0]]
                                                                  for i in range(gb.n_frames)]}
    with open(os.path.join(GOLD, "oracle_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(GOLD, "oracle_vectors.json"))


if __name__ == "__main__":
    main()
