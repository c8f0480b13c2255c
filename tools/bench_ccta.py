"""Timing of the CCTA diameter search (find_aortic_scaling: 41 scalings x symmetric nearest-neighbour
distance in 3-D) beside the CPU port.  Not the headline metric.  One search = 41 * 2 * N * M
squared-distance evaluations (9 fp64 VALU operations each in k_nn3_min).
Usage: python tools/bench_ccta.py [--points N] [--reference M] [--reps R] [--skip-cpu]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--reference", type=int, default=20000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-cpu", action="store_true")
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import multimoda_rs_amd as mm
    case = mm.synth.synthetic_tube_case(n_points=a.points, n_reference=a.reference, true_scaling_mm=0.7, seed=4)
    eng = mm.Engine()
    best = mm.find_aortic_scaling(case["points"], case["reference"], case["centerline"], engine=eng)   # warm-up
    t = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        best, d = mm.find_aortic_scaling(case["points"], case["reference"], case["centerline"], engine=eng,
                                         return_distances=True)
        t.append(time.perf_counter() - t0)
    pair_evals = 41 * 2 * a.points * a.reference
    res = dict(workload=f"find_aortic_scaling N={a.points} M={a.reference}", pair_evals=pair_evals,
               gpu_path_ms=min(t) * 1e3, gpu_path_pair_evals_per_s=pair_evals / min(t), best_scaling=best)
    if not a.skip_cpu:
        from oracle import oracle_ccta as occ, oracle_cl as ocl
        from helpers import to_oracle_cl
        t0 = time.perf_counter()
        obest, od = occ.aortic_diameter_optimization(case["points"], case["reference"],
                                                     to_oracle_cl(ocl, case["centerline"]))
        res["cpu_port_ms"] = (time.perf_counter() - t0) * 1e3
        res["cpu_port_threads"] = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count()))
        res["identical"] = bool(obest == best and (od == d).all())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
