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
