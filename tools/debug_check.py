import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge; ge.build()
import multimoda_rs_amd as mm
from oracle import oracle as orc
from helpers import to_oracle
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
step = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
base = mm.synthetic_case(F, 501)
eng = mm.Engine(0)
geoms = [g.copy() for g in base]
og = [to_oracle(orc, g) for g in base]
wp = mm.WithinPlan(eng, geoms, step, 180.0, True, 501, precision=mm.MM_PRECISION_F32_FAST)
logs, ev, un = wp.run()
ologs = [orc.align_within_chain(o, step, 180.0, True, 501, n_threads=16) for o in og]
print("logs equal", logs == ologs, "unresolved", un)
def cmp(tag):
    for k, (g, o) in enumerate(zip(geoms, og)):
        for nm in ("lumen", "cath", "centroids", "ref"):
            a, b = getattr(g, nm), getattr(o, nm)
            if not np.array_equal(a, b):
                d = np.abs(a - b)
                idx = np.unravel_index(np.argmax(d), d.shape)
                print(tag, "geom", k, nm, "differs: max", d.max(), "at", idx, "n_diff", int((d > 0).sum()))
cmp("after within")
for pairs in (((0, 1), (2, 3)), ((0, 2), (1, 3))):
    rot, _ = mm.align_between(eng, [(geoms[i], geoms[j]) for i, j in pairs], 180.0, step, 501, mm.MM_PRECISION_F32_FAST)
    orot = [orc.align_between(og[i], og[j], 180.0, step, 501, n_threads=16) for i, j in pairs]
    print("between", pairs, list(rot) == orot, list(rot), orot)
    cmp("after between %s" % (pairs,))
