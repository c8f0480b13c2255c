"""API-level cost of one `from_array_full` call on the config3 shape (4 pullbacks x 512 frames x 501 points,
0.5 deg x +-180 deg brute force, smooth + postprocessing on, write_obj off): wall time of the whole call -- geometry
building from InputData arrays, the searches, the post-steps of align_frames_in_geometry, the between
alignments, postprocess_geom_pair -- and a cProfile of where the host time goes.
Usage: python tools/bench_api.py [--profile] [--frames 512]"""
import argparse
import cProfile
import json
import os
import pstats
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
import multimoda_rs_amd as mm  # noqa: E402


def retain_heap(mmap_threshold: int = 32 << 20, trim_threshold: int = 1 << 30, top_pad: int = 64 << 20) -> bool:
    """Opt-in allocator policy for processes that call the entry points repeatedly (glibc only; returns False elsewhere).

    One `from_array_full` on 4 x 512 frames returns ~100 MB of arrays and builds ~50 MB of intermediates.  With glibc's
    defaults every array above 128 KB is its own mmap: freeing the previous call's results unmaps them (3-5 ms on the
    config3 shape) and the next call page-faults the same memory in again.  This sets M_MMAP_THRESHOLD (blocks up to
    32 MB come from the heap), M_TRIM_THRESHOLD and M_TOP_PAD (the heap is not handed back between calls), process-wide:
    the memory of freed results stays with the process.  Nothing in the library depends on it."""
    import ctypes
    try:
        libc = ctypes.CDLL("libc.so.6")
        mallopt = libc.mallopt
    except (OSError, AttributeError):
        return False
    M_TRIM_THRESHOLD, M_TOP_PAD, M_MMAP_THRESHOLD = -1, -2, -3
    ok = mallopt(M_MMAP_THRESHOLD, int(mmap_threshold)) == 1
    ok = (mallopt(M_TRIM_THRESHOLD, int(trim_threshold)) == 1) and ok
    ok = (mallopt(M_TOP_PAD, int(top_pad)) == 1) and ok
    return bool(ok)




def input_data(g, label, diastole):
    """The (N, 4) [frame, x, y, z] array contract of numpy_to_inputdata from a synthetic pullback."""
    F = g.n_frames
    n = np.diff(g.lumen_off)
    frame = np.repeat(g.orig_frames.astype(np.float64), n)
    arr = np.concatenate([frame[:, None], g.lumen], axis=1)
    ref_i = int(np.nonzero(g.has_ref)[0][0])
    ref = np.concatenate([[float(g.orig_frames[ref_i])], g.ref[ref_i]])
    return mm.numpy_to_inputdata(arr, ref, diastole, label=label)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--stages", action="store_true", help="per-stage wall times of one call (MM_API_TRACE)")
    ap.add_argument("--retain-heap", action="store_true", help="call retain_heap() first (see its docstring)")
    a = ap.parse_args()
    if a.retain_heap:
        print("retain_heap:", retain_heap(), file=sys.stderr)
    base = mm.synthetic_case(a.frames, 501)
    data = [input_data(g, lab, dia) for g, lab, dia in zip(base, ("rest", "rest", "stress", "stress"), (True, False, True, False))]
    eng = mm.Engine()
    kw = dict(step_rotation_deg=0.5, range_rotation_deg=180.0, sample_size=501, write_obj=False, bruteforce=True,
              smooth=True, postprocessing=True, engine=eng)
    mm.from_array_full(*data, **kw)                       # warm-up
    ts = []
    for _ in range(a.repeats):
        t0 = time.perf_counter()
        out = mm.from_array_full(*data, **kw)
        ts.append(time.perf_counter() - t0)
    res = {"workload": f"from_array_full, 4 x {a.frames} frames x 501 pts, 0.5 deg x +-180 deg bruteforce, smooth + postprocessing",
           "ms_median": 1e3 * statistics.median(ts), "ms_all": [1e3 * t for t in ts],
           "frames_out": [int(p.geom_a.n_frames) for p in out[:4]]}
    print(json.dumps(res))
    if a.stages:
        from multimoda_rs_amd import api as API
        for rep in range(3):
            API._TRACE = []
            t0 = time.perf_counter()
            mm.from_array_full(*data, **kw)
            t1 = time.perf_counter()
            prev = t0
            print(f"-- call {rep}: {1e3 * (t1 - t0):.2f} ms")
            for label, t in API._TRACE:
                print(f"   {1e3 * (t - prev):8.3f} ms  {label}")
                prev = t
            print(f"   {1e3 * (t1 - prev):8.3f} ms  (return)")
        API._TRACE = None
    if a.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            mm.from_array_full(*data, **kw)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(40)


if __name__ == "__main__":
    main()
