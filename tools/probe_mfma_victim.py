"""PROBE, second half of tools/probe_mfma_interference.py: the packed-FMA bound kernel (k_screen_lb through the mm_lower_bounds
test hook) ALONE in this process, 300 calls compared with the first.  Run it once by itself (0 differing) and once beside
`tools/bin/ubench_aggr 25` (an MFMA loop of another process: 16 of 300 differing on the box of profiles/r4_mfma_interference.txt)
or `tools/bin/ubench_aggr 25 2` (a plain-FMA loop: 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multimoda_rs_amd as mm
base = mm.synthetic_case(128, 501)
g = base[0]
ref = mm.search_set(g, 2, 501) - g.centroids[2, :2]
tgt = mm.search_set(g, 3, 501) - g.centroids[3, :2]
angles, _, _ = mm.search_angles(0.05, 180.0)
with mm.Engine(0) as e0:
    run = lambda: e0.lower_bounds(ref, tgt, angles, (0.0, 0.0), matrix=False)[0]
    r0 = run()
    bad = sum(not np.array_equal(run(), r0) for _ in range(300))
    print("victim k_screen_lb alone in this process:", "differing runs", bad, "of 300", flush=True)
