import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.build()
import multimoda_rs_amd as mm
eng = mm.Engine(0)
base = mm.synthetic_case(512, 501)
def mk():
    geoms = [g.copy() for g in base]
    return mm.WithinPlan(eng, geoms, 0.5, 180.0, True, 501, precision=mm.MM_PRECISION_F32_FAST)
for name in ("run", "run_sharded", "run", "run_sharded"):
    plans = [mk() for _ in range(4)]
    eng.synchronize()
    ts = []
    for p in plans:
        t0 = time.perf_counter()
        getattr(p, name)()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(name, ["%.3f" % t for t in ts])
    # per-part timing of the sharded path
if True:
    from multimoda_rs_amd import distributed as D
    p = mk(); eng.synchronize()
    t0 = time.perf_counter(); nj, nl, tol = p.dims(); t1 = time.perf_counter()
    local = p.level_local(0, nj); t2 = time.perf_counter()
    ok, angle, _i, _c = D.merge_level(local, tol, None); t3 = time.perf_counter()
    p.level_commit(0, ok, angle); t4 = time.perf_counter()
    p.walk(); t5 = time.perf_counter()
    print("dims %.3f local %.3f merge %.3f commit %.3f walk %.3f" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
