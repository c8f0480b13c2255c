import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge; ge.build()
import multimoda_rs_amd as mm
from oracle import oracle as orc
from helpers import to_oracle
F, step = 200, 1.0
base = mm.synthetic_case(F, 501)
def eq(a, b):
    return all(np.array_equal(getattr(x, n), getattr(y, n)) for x, y in zip(a, b) for n in ("lumen", "cath", "centroids"))
o1 = [to_oracle(orc, g) for g in base]; o16 = [to_oracle(orc, g) for g in base]
l1 = [orc.align_within_chain(o, step, 180.0, True, 501, n_threads=1) for o in o1[:2]]
l16 = [orc.align_within_chain(o, step, 180.0, True, 501, n_threads=16) for o in o16[:2]]
print("oracle 1 vs 16 threads: logs", l1 == l16, "geom", eq(o1[:2], o16[:2]))
if "--gpu" in sys.argv:
    eng = mm.Engine(0)
    res = []
    for mode in (0, 1, 1):
        geoms = [g.copy() for g in base[:2]]
        logs, _ = mm.align_within(eng, geoms, step, 180.0, True, 501, precision=mm.MM_PRECISION_F32_FAST, mode=mode)
        res.append((logs, geoms))
    print("product mode0 vs mode1: logs", res[0][0] == res[1][0], "geom", eq(res[0][1], res[1][1]))
    print("product mode1 twice: geom", eq(res[1][1], res[2][1]))
    print("product mode0 vs oracle1: logs", res[0][0] == l1, "geom", eq(res[0][1], o1[:2]))
    print("product mode1 vs oracle1: geom", eq(res[1][1], o1[:2]))
    eng.close()
