"""PROBE (round 4 finding): the packed-FMA kernels of this library return WRONG values, now and then, while a kernel that
executes MFMAs runs on the chip at the same time -- another engine's k_screen_mx here, or any MFMA loop of another process
(tools/probe_mfma_victim.py beside tools/bin/ubench_aggr).  Victims: k_screen_lb (lower bounds of the packed-FMA bounded
search), k_screen_fast (MM_PRECISION_F32_FAST), k_search<float> (MM_PRECISION_F32): single candidates come out too LARGE
(e.g. 0.19 for 0.15 mm^2), in 3 - 15 % of the calls.  Not disturbed: k_screen_mx, k_bound_mx, the exact f64 kernels, the
small per-pair kernels; and a plain-FMA aggressor disturbs nothing.  The cause is not known (synthetic victims built from
the same instruction kinds -- v_pk_fma_f32, DPP row minima, ds_read_b128, ds_bpermute, v_min3_f32 -- pass beside the same
aggressor: tools/ubench_interf.hip; LDS / VGPR / global patterns of a victim stay intact: tools/ubench_interf2.hip), so the
library AVOIDS the combination: under MM_PRECISION_F32_MATRIX and MM_PRECISION_F32_BOUNDED every value that decides a
result comes from an MFMA or an f64 kernel (profiles/README.md).  One victim call after the other beside a `fast` and a
`matrix` between-alignment loop on a second engine; prints the number of calls whose values differ from a quiet run."""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multimoda_rs_amd as mm
base = mm.synthetic_case(128, 501)
g = base[0]
ref = mm.search_set(g, 2, 501) - g.centroids[2, :2]
tgt = mm.search_set(g, 3, 501) - g.centroids[3, :2]
angles, _, _ = mm.search_angles(0.05, 180.0)
stop = False
def worker_b(eng, prec):
    while not stop:
        gs = [x.copy() for x in base]
        mm.align_between(eng, [(gs[0], gs[1]), (gs[2], gs[3])], 180.0, 0.25, 501, prec)
with mm.Engine(0) as e0, mm.Engine(0) as e1:
    for victim in ("lb_packed", "lb_matrix", "costs_fast", "costs_f32", "costs_matrix"):
        def run():
            if victim.startswith("lb"):
                return e0.lower_bounds(ref, tgt, angles, (0.0, 0.0), matrix=victim == "lb_matrix")[0]
            prec = {"costs_fast": mm.MM_PRECISION_F32_FAST, "costs_f32": mm.MM_PRECISION_F32, "costs_matrix": mm.MM_PRECISION_F32_MATRIX}[victim]
            return np.asarray(e0.best_rotation(ref, tgt, angles, (0.0, 0.0), precision=prec, return_costs=True)[3])
        r0 = run()
        for name, prec in (("fast", mm.MM_PRECISION_F32_FAST), ("matrix", mm.MM_PRECISION_F32_MATRIX)):
            stop = False
            t = threading.Thread(target=worker_b, args=(e1, prec)); t.start()
            bad, worst = 0, 0.0
            for _ in range(60):
                r = run()
                if not np.array_equal(r, r0):
                    bad += 1; worst = max(worst, float(np.abs(r - r0).max()))
            stop = True; t.join()
            print(f"victim {victim:12s} beside a {name:6s} between: differing runs {bad} of 60, largest difference {worst:.3e}", flush=True)
