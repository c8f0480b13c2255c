import os, sys, math, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge; ge.build()
import multimoda_rs_amd as mm
from oracle import oracle as orc
from helpers import to_oracle
F, step = 512, 0.5
base = mm.synthetic_case(F, 501)
eng = mm.Engine(0)
geoms = [g.copy() for g in base]
og = [to_oracle(orc, g) for g in base]
logs, _ = mm.align_within(eng, geoms, step, 180.0, True, 501, precision=mm.MM_PRECISION_F32_FAST, mode=1)
ologs = [orc.align_within_chain(o, step, 180.0, True, 501, n_threads=16) for o in og]
print("logs equal", logs == ologs)
libm = ctypes.CDLL("libm.so.6")
libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
for k in range(4):
    g, o = geoms[k], og[k]
    d = (g.lumen != o.lumen).any(axis=1)
    frames = sorted(set(np.searchsorted(g.lumen_off, np.nonzero(d)[0], side="right") - 1))
    print("geom", k, "frames with diffs:", frames[:10], "count", len(frames))
    # replay frame f in python from the ORIGINAL frame using the logs: rotate by cumulative, translate, rotate by best
    for f in frames[:3]:
        cum = 0.0
        for i in range(1, f):
            cum += math.radians(0)  # placeholder
        # cumulative = sum of best angles (radians) of frames 1..f-1; best = rot_deg * pi/180 is not exactly invertible,
        # so recover the candidates from the angle list
        angles = mm.search_angles(step, 180.0)[0]
        bests = []
        for (cid, mid, rot, tx, ty, cx, cy) in logs[k][:f]:
            j = int(np.argmin(np.abs(np.degrees(angles) - rot)))
            assert math.degrees(angles[j]) == rot or abs(math.degrees(angles[j]) - rot) < 1e-9
            bests.append(float(angles[j]))
        cum = 0.0
        for b in bests[:f - 1]:
            cum += b
        s, c = ctypes.c_double(), ctypes.c_double()
        libm.sincos(cum, ctypes.byref(s), ctypes.byref(c))
        print("  frame", f, "cumulative", cum, "sin/cos == sincos:", math.sin(cum) == s.value, math.cos(cum) == c.value,
              "| best", bests[f - 1])
        libm.sincos(bests[f - 1], ctypes.byref(s), ctypes.byref(c))
        print("     best: sin/cos == sincos:", math.sin(bests[f - 1]) == s.value, math.cos(bests[f - 1]) == c.value)
eng.close()
