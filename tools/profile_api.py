"""cProfile of one API-level call (from_file_full on the bundled IVUS data), to see where the host time of the
published-benchmark protocol goes.  Usage: python tools/profile_api.py [step_deg]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
import multimoda_rs_amd as mm  # noqa: E402

step = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
rest = os.path.join(ROOT, "tests", "golden", "examples_ivus_rest")
stress = os.path.join(ROOT, "tests", "golden", "examples_ivus_stress")
eng = mm.Engine()
kw = dict(step_rotation_deg=step, range_rotation_deg=90.0, write_obj=False, smooth=False, postprocessing=False,
          bruteforce=True, interpolation_steps=0, engine=eng)
mm.from_file_full(rest, stress, **kw)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    mm.from_file_full(rest, stress, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
