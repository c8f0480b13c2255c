"""One decoupled within-alignment of 4 pullbacks x F frames x 501 points (0.5 deg x +-180 deg brute force) for a large F:
brute-force screen and bounded search, staging and run time, and whether the two agree on every log entry and
coordinate.  Usage: python tools/bench_scale.py [F = 4096]   (F = 4096: 16 380 frame pairs, 11.8 M candidates, 8x config3)"""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import numpy as np
import __graft_entry__ as ge; ge.build()
import multimoda_rs_amd as mm
eng = mm.Engine()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = mm.synthetic_case(F, 501)
for prec in (mm.MM_PRECISION_F32_FAST, mm.MM_PRECISION_F32_BOUNDED):
    geoms = [g.copy() for g in base]
    t0 = time.perf_counter()
    wp = mm.WithinPlan(eng, geoms, 0.5, 180.0, True, 501, precision=prec)
    t1 = time.perf_counter()
    logs, evals, unres = wp.run()
    t2 = time.perf_counter()
    wp.close()
    rots = np.array([l[2] for l in logs[0]])
    print(f"F={F} prec={prec}: stage {1e3*(t1-t0):.1f} ms, run {1e3*(t2-t1):.1f} ms, {evals/ (t2-t1)/1e6:.1f} M pose-evals/s, unresolved {unres}, "
          f"rot range [{rots.min():.2f}, {rots.max():.2f}] deg, first logs {logs[0][0][:3]}")
    if prec == mm.MM_PRECISION_F32_FAST:
        ref = (logs, [g.lumen.copy() for g in geoms])
    else:
        same = all(a == b for a, b in zip(ref[0], logs)) and all(np.array_equal(x, g.lumen) for x, g in zip(ref[1], geoms))
        print("bounded == brute force:", same)
