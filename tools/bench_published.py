"""The reference's own published benchmark, run on this engine: `from_file_full` on the bundled IVUS
rest/stress data (20+17+25+22 frames x 501 points; committed as tests/golden/examples_ivus_{rest,stress}), step
5 ... 0.05 deg, range +-90 deg, bruteforce and optimized, write_obj / smooth / postprocessing off, median
of 3 wall-times including CSV parsing and geometry building -- the protocol of the reference's
benchmarks/benchmark_bruteforce_stepsize.py:26-80.  The reference's numbers (docs/benchmark.rst:36-38,
Xeon Gold 6234, 16 threads): 64.4 s bruteforce / 6.25 s optimized at 0.05 deg.
Usage: python tools/bench_published.py [--check]   (--check: chain logs against the CPU oracle at 0.5 deg)"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STEP_SIZES = [5.0, 2.5, 1.0, 0.5, 0.25, 0.1, 0.05]
RANGE_DEG = 90.0
REPEATS = 3
PUBLISHED = {"bruteforce": {0.05: 64.4}, "optimized": {0.05: 6.25}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import multimoda_rs_amd as mm
    rest = os.path.join(ROOT, "tests", "golden", "examples_ivus_rest")
    stress = os.path.join(ROOT, "tests", "golden", "examples_ivus_stress")
    eng = mm.Engine()

    def run(step, brute):
        t0 = time.perf_counter()
        out = mm.from_file_full(rest, stress, step_rotation_deg=step, range_rotation_deg=RANGE_DEG, write_obj=False,
                                smooth=False, postprocessing=False, bruteforce=brute, interpolation_steps=0, engine=eng)
        return time.perf_counter() - t0, out

    run(1.0, True)                                          # warm-up: library load, buffers
    res = {"bruteforce": {}, "optimized": {}}
    for step in STEP_SIZES:
        for name, brute in (("bruteforce", True), ("optimized", False)):
            res[name][step] = statistics.median([run(step, brute)[0] for _ in range(REPEATS)])
    out = {"workload": "from_file_full, ivus_rest + ivus_stress (20+17+25+22 frames x 501 pts), range +-90 deg, "
                       "write_obj/smooth/postprocessing off, median of 3 wall-times incl. CSV parsing",
           "seconds": {k: {str(s): v for s, v in d.items()} for k, d in res.items()},
           "reference_published_seconds": {"bruteforce@0.05": 64.4, "optimized@0.05": 6.25,
                                           "hardware": "Xeon Gold 6234, 16 threads (docs/benchmark.rst:4-7,36-38)"},
           "speedup_vs_published": {"bruteforce@0.05": 64.4 / res["bruteforce"][0.05],
                                    "optimized@0.05": 6.25 / res["optimized"][0.05]}}
    if a.check:
        from oracle import oracle as O
        from helpers import to_oracle
        ok = True
        for brute in (True, False):
            _, got = run(0.5, brute)
            logs = got[-1]
            geoms = [mm.build_geometry_from_inputdata(None, p, os.path.basename(p), dia) for p in (rest, stress)
                     for dia in (True, False)]
            for g, lg in zip(geoms, logs):
                exp = O.align_within_chain(to_oracle(O, g), 0.5, RANGE_DEG, brute, 500, n_threads=8)
                ok = ok and list(lg) == exp
        out["chain_logs_identical_to_oracle@0.5"] = bool(ok)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
