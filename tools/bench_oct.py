"""The reference's second published benchmark on this engine: `from_array_single` on an OCT-shaped pullback
(280 frames, 0.01 deg x +-6 deg, sample_size 200, n_points 40), brute force and optimized, end to end and the search alone
-- bench.py's `oct_single` leg on its own.  Protocol: benchmarks/benchmark_cpu_scaling.py:32-80; published figures
docs/benchmark.rst:53-86 (14.15 s / 2.40 s on 16 Xeon threads, the reference's own OCT data, which its checkout does not
hold).  Usage: python tools/bench_oct.py [--repeats 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
import multimoda_rs_amd as mm  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--precision", default="matrix", choices=["matrix", "fast", "bounded", "f32", "f64"])
    a = ap.parse_args()
    prec = {"matrix": mm.MM_PRECISION_F32_MATRIX, "fast": mm.MM_PRECISION_F32_FAST, "bounded": mm.MM_PRECISION_F32_BOUNDED,
            "f32": mm.MM_PRECISION_F32, "f64": mm.MM_PRECISION_F64}[a.precision]
    with mm.Engine(0) as eng:
        print(json.dumps(bench.oct_single_leg(mm, eng, prec, repeats=a.repeats)))
