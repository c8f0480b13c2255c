"""Timing of the third call site (refine_alignment_hausdorff / align_combined) beside the CPU port.

Not the headline metric (bench.py measures that); this reports, for one BASELINE-config-5-shaped
problem, the wall time of the product path (host rebuild of every candidate + one GPU batch) and of
the oracle's serial CPU restatement of the same grid.  Usage: python tools/bench_refine.py [--frames F]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--points", type=int, default=501)
    ap.add_argument("--ccta", type=int, default=12000)
    ap.add_argument("--angle-range", type=float, default=15.0)
    ap.add_argument("--index-range", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-cpu", action="store_true")
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import multimoda_rs_amd as mm
    case = mm.synth.synthetic_centerline_case(n_frames=a.frames, n_points=a.points, n_ccta=a.ccta, seed=5,
                                              true_rotation_deg=37.0, true_index=12)
    eng = mm.Engine()
    args = (case["centerline"], case["geometry"], case["main_ref_pt"], case["ccw_ref_pt"], case["cw_ref_pt"],
            case["points"])
    kw = dict(angle_step_deg=1.0, angle_range_deg=a.angle_range, index_range=a.index_range, engine=eng)
    out, sp, rot = mm.align_combined(*args, **kw)        # warm-up (module load, buffers)
    os.environ["MM_TRACE"] = "0"
    t = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        out, sp, rot = mm.align_combined(*args, **kw)
        t.append(time.perf_counter() - t0)
    n_evals = out.meta["refine_evals"]
    res = dict(workload=f"align_combined F={a.frames} M={a.points} ccta={len(case['points'])} grid={n_evals}",
               gpu_path_ms=min(t) * 1e3, total_rotation_deg=rot, refined_idx=out.meta["refined_cl_ref_idx"])
    if not a.skip_cpu:
        from oracle import oracle as O, oracle_cl as ocl
        from helpers import to_oracle, to_oracle_cl
        og = to_oracle(O, case["geometry"])
        t0 = time.perf_counter()
        osp, orot, oidx = ocl.align_combined(to_oracle_cl(ocl, case["centerline"]), [og],
                                             case["geometry"].meta["ref_point_index"], case["main_ref_pt"],
                                             case["ccw_ref_pt"], case["cw_ref_pt"], case["points"], math.radians(1.0),
                                             math.radians(a.angle_range), a.index_range)
        res["cpu_port_ms"] = (time.perf_counter() - t0) * 1e3
        res["cpu_port_threads"] = 1
        res["identical"] = bool(orot * (180.0 / math.pi) == rot and oidx == out.meta["refined_cl_ref_idx"]
                                and (og.lumen == out.lumen).all())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
