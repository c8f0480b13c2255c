#!/usr/bin/env python3
"""bench.py -- throughput of the Hausdorff pose search on MI355X.

Metric (BASELINE.json): Hausdorff pose-evals/sec (frames x poses) for a 4-phase full
alignment, with the best pose identical to the reference algorithm's.

A "step" is one full 4-phase alignment of a synthetic case (4 pullbacks x F frames x 501
points, N = 521 points per set): 4 within-pullback chains (bruteforce) and the AB|CD,
AC|BD between-pullback alignments, all through the product's C ABI.  Workloads
(SURVEY.md section 8(d)):
  config2: F = 128, step 1 deg,  range 180 deg -> 361 candidates/search
  config3: F = 512, step 0.5 deg, range 180 deg -> 721 candidates/search (default: the
           512-frame x 501-pt case BASELINE.json's targets are quoted on, and the one its
           multi-GPU config shards)

`value` is the WHOLE step: the case starts on the host, its raw pullbacks go to HBM over PCIe, the
search sets are built on the device, then search (N > 1: + the RCCL exchange), chain walk and between
alignments.  Consecutive (independent) cases are pipelined: K stagings, K searches, K finishes inside
the timed region.  Candidates are scored by the matrix-pipe screen (MM_PRECISION_F32_MATRIX: squared distances from
the f16 matrix pipe, minima on the vector pipe) + exact f64 re-score; the packed-FMA screen, the all-f64 kernel, the
bounded search, the reference's default ladder and the extension grid are legs of the same line.

Usage: python bench.py --gpus N --steps K --warmup W   (N > 1: under torch.distributed.run, or plainly --
then the N ranks are started as a child job)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "config2": dict(frames=128, points=501, step_deg=1.0, range_deg=180.0, sample_size=501),
    "config3": dict(frames=512, points=501, step_deg=0.5, range_deg=180.0, sample_size=501),
    "tiny": dict(frames=12, points=501, step_deg=2.0, range_deg=180.0, sample_size=501),
    # EXTENSION axis (absent from the reference's 4-phase path): every frame against a
    # window of 100 neighbouring frames x 721 rotations; reported separately, never as the headline
    "config3ext": dict(frames=512, points=501, step_deg=0.5, range_deg=180.0, sample_size=501, shift=(-50, 49)),
}
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, Peak FP32 (vector)
# What bounds every screen of this path is vector ISSUE: a SIMD takes one vector wave-instruction per 4 clocks (an MFMA holds
# the issue port for 8: two slots).  256 CUs x 4 SIMDs x 2.4 GHz / 4 clk = 614.4 G issue slots per second.
VALU_ISSUE_PEAK_GSLOTS = 256 * 4 * 2.4 / 4.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md, HBM3E peak
FLOPS_PER_PAIR_EVAL = 6.0         # 2 sub, 2 mul, 1 add, 1 min
BYTES_PER_POSE_EVAL = lambda na, nb: (na + nb) * 2 * 4 + 8   # SURVEY 8(d) no-reuse model


def between_stage(mm, eng, geoms, cfg, precision=1):
    """AB | CD then AC | BD (entry.rs:206-277)."""
    a, b, c, d = geoms
    r1, e1 = mm.align_between(eng, [(a, b), (c, d)], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], precision)
    r2, e2 = mm.align_between(eng, [(a, c), (b, d)], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], precision)
    return np.concatenate([r1, r2]), e1 + e2


def full_alignment(mm, eng, geoms, cfg, plan=None, precision=1):
    """One 4-phase alignment (entry.rs:140-277 order); returns (logs, between angles, pose_evals).
    plan = a pre-staged mm.WithinPlan (decoupled mode, point sets already in HBM) or None
    (faithful per-step chain)."""
    if plan is None:
        logs, evals = mm.align_within(eng, geoms, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"],
                                      precision=precision, mode=0)
        unresolved = 0
    else:
        t0 = time.perf_counter()
        logs, evals, unresolved = plan.run()
        t1 = time.perf_counter()
    rot, e2 = between_stage(mm, eng, geoms, cfg, precision)
    if plan is not None and os.environ.get("MM_TRACE"):
        print(f"[bench trace] within {1e3 * (t1 - t0):.3f} ms, between {1e3 * (time.perf_counter() - t1):.3f} ms",
              file=sys.stderr)
    return logs, rot, evals + e2, unresolved


def run_steps(ks, search, finish, pipelined, stage=None, lookahead=2, begin=None, stager=False, pre=None, finishers=1):
    """Run steps `ks`: search(k) then finish(k), and -- if `stage` is given -- stage(k + lookahead) after
    finish(k) (the caller has staged the first `lookahead` steps of `ks` itself: priming), so a call over K steps
    does K stagings, K searches and K finishes.
    pipelined: steps are independent cases, so the search of step k+1 (the GPU-heavy half, and the only half with
    collectives) overlaps the chain walk and between alignment of step k and the staging of a later step on other
    host threads.  Step k lives on engine k % lookahead: step k + lookahead is staged only after step k is finished
    (it takes over that engine), and searched only after it is staged.  Every step's work completes inside the call.
    stager (pipelined, lookahead >= 3): the stagings run on a third host thread, so stage(k + lookahead) overlaps
    finish(k + 1) -- with three engines the step being searched, the one being finished and the one being staged
    never share an engine.  Without it the finishing thread stages too (two engines suffice).
    begin (optional, pipelined only): the search split in two -- begin(k, prev) enqueues step k's launch behind step
    prev's long kernel and returns, search(k) then only collects -- so that step k+1 is already queued on the
    device when step k's launch ends: the device does not idle while the host fetches, commits and launches.
    pre (optional, with begin): pre(k) is called before begin(k+1, k) -- the sharded search enqueues step k's whole
    exchange there (exports, all-reduces, record copy; nothing waited for), so that begin(k+1, k) can order step k+1's
    launch behind it.
    finishers: host threads that finish steps (chain walk + between alignments, ~2.8 ms per step): one keeps up with a
    20-30 ms step; at N = 8 a step is 2.5 ms and two take turns."""
    ks = list(ks)
    if not pipelined:
        out = []
        for k in ks:
            search(k)
            out.append(finish(k))
            if stage is not None:
                stage(k + lookahead)
        return out
    import queue
    import threading
    L = lookahead
    q, sq, out, err = queue.Queue(), queue.Queue(), {}, []
    # ready[j]: step j may be started -- the step it takes its engine over from (j - L) is finished and, if this
    # call stages it, j is staged
    ready = {k + L: threading.Event() for k in ks}

    def guarded(fn):
        def run():
            try:
                fn()
            except BaseException as ex:   # surfaced on the main thread
                err.append(ex)
                for e in ready.values():
                    e.set()
        return run

    def finisher():
        while True:
            k = q.get()
            if k is None:
                return
            out[k] = finish(k)
            if stage is not None and stager:
                sq.put(k + L)
            else:
                if stage is not None:
                    stage(k + L)
                ready[k + L].set()

    def stage_worker():
        while True:
            j = sq.get()
            if j is None:
                return
            stage(j)
            ready[j].set()

    threads = [threading.Thread(target=guarded(finisher), name=f"bench-finish-{i}") for i in range(max(1, finishers))]
    n_fin = len(threads)
    if stage is not None and stager:
        threads.append(threading.Thread(target=guarded(stage_worker), name="bench-stage"))
    for th in threads:
        th.start()

    def wait_ready(j):
        if j in ready and (j - L) in out_keys:
            ready[j].wait()

    out_keys = set(ks)
    try:
        if begin is not None and ks:
            begin(ks[0], None)
        for i, k in enumerate(ks):
            if begin is None:
                wait_ready(k)                     # step k takes its engine over from step k - L (finished, k staged)
                if err:
                    break
                search(k)
            else:
                if pre is not None:
                    pre(k)
                if i + 1 < len(ks):
                    wait_ready(ks[i + 1])
                    if err:
                        break
                    begin(ks[i + 1], k)           # queued on the device behind step k's long kernel
                if err:
                    break
                search(k)                         # collect step k (waits for its kernels), remaining levels, commit
            q.put(k)
    finally:
        for _ in range(n_fin):
            q.put(None)
        for th in threads[:n_fin]:
            th.join()
        sq.put(None)
        for th in threads[n_fin:]:
            th.join()
    if err:
        raise err[0]
    return [out[k] for k in ks]


# candidates of the big launch of the committed --pmc passes (config3: 4 pullbacks x 511 frame pairs x 721 rotations)
PMC_CANDIDATES = {"config3": 4 * 511 * 721}


def issue_roofline(pmc, ms, pair_evals, na, nb):
    """The roofline of the dominant kernel on the scale that binds it: vector issue slots per second.
    achieved = issue slots of one big launch -- an INSTRUCTION COUNT from the committed rocprofv3 --pmc pass
    (SQ_INSTS_VALU + SQ_INSTS_MFMA: every vector instruction one slot, an MFMA two), scaled by the candidates of the live
    launch where the workload differs from the profiled one (same kernel, same set size: same slots per candidate) --
    divided by the big launch's mean duration measured live (hipEvents on the kernel's stream);
    peak = 1024 SIMDs x 2.4 GHz / 4 clk.  None where no --pmc pass of the kernel is committed."""
    if not pmc or not pmc.get("issue_slots_per_launch") or not pmc.get("candidates_per_launch") or len(ms) == 0:
        return None
    big = pair_evals >= 0.5 * pair_evals.max()
    t_ms = float(ms[big].mean())
    cand = float(pair_evals[big].mean()) / (2.0 * na * nb)
    slots = pmc["issue_slots_per_launch"] * cand / pmc["candidates_per_launch"]
    ach = slots / (t_ms * 1e-3) * 1e-9
    return {"achieved": ach, "frac": ach / VALU_ISSUE_PEAK_GSLOTS, "issue_slots_per_launch": slots,
            "issue_slots_per_candidate": pmc["issue_slots_per_launch"] / pmc["candidates_per_launch"],
            "candidates_per_launch": cand, "launch_ms": t_ms}


def committed_pmc(workload, precision):
    """Figures of the dominant kernel's big launch from the committed rocprofv3 --pmc passes (profiles/, tools/gpu_pmc.sh;
    bench.py cannot run the profiler itself): HBM bytes per launch (FETCH_SIZE + WRITE_SIZE in KiB, raw -- on gfx950
    FETCH_SIZE can under-report wide streaming reads by up to 2x; these are dword loads, uncalibrated) and the quantity
    that actually saturates: VALU issue.  None where no matching profile is committed."""
    names = {("config3", "fast"): ["r3_config3_fast_pmc_summary.csv", "r2_config3_fast_pmc_summary.csv"],
             ("config3", "matrix"): ["r4_config3_matrix_pmc_summary.csv", "r3_config3_matrix_pmc_summary.csv"],
             ("config3", "f32"): ["r1_config3_pmc_summary.csv"]}.get((workload, precision), [])
    if not names and workload != "config3":
        # same kernel, same set size (N = 521), other frame / candidate counts: the per-candidate figures carry over
        out = committed_pmc("config3", precision)
        if out:
            out["source"] += " (config3's launch; per-candidate figures)"
            out.pop("traffic", None)
        return out
    path = next((os.path.join(ROOT, "profiles", n) for n in names if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if not path:
        return None
    import csv
    val, dur, grid = {}, {}, 0
    for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
        if r["kernel"].startswith(("k_screen_mx",) if precision == "matrix" else ("k_screen_fast", "k_search<float")):
            g = int(r["grid_threads"])
            if g >= grid:
                if g > grid:
                    val, dur, grid = {}, {}, g
                val[r["counter"]] = float(r["mean_value"])
                dur[r["counter"]] = float(r.get("mean_duration_ns") or 0.0)
    out = {"source": "profiles/" + os.path.basename(path), "grid_threads": grid}
    if "FETCH_SIZE" in val and "WRITE_SIZE" in val:
        out["traffic"] = (val["FETCH_SIZE"] + val["WRITE_SIZE"]) * 1024.0
    if "GRBM_GUI_ACTIVE" in val and "SQ_INSTS_VALU" in val:
        cycles = val["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
        out["valu_clk_per_instr"] = cycles / (val["SQ_INSTS_VALU"] / 1024.0)      # per SIMD (256 CUs x 4)
        out["valu_instr_per_launch"] = val["SQ_INSTS_VALU"]
        if dur.get("GRBM_GUI_ACTIVE"):
            out["achieved_clock_ghz"] = cycles / dur["GRBM_GUI_ACTIVE"]
        if "SQ_INSTS_MFMA" in val and val["SQ_INSTS_MFMA"] > 0:
            # matrix-pipe screen: one MFMA per 32 x 32 tile; the vector pipe folds it with 16 minima at best.  An MFMA holds
            # the SIMD's vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md), a vector instruction for 4: two issue
            # slots against one.  SQ_INSTS_VALU counts the MFMAs as well (the kernel's instruction count from its ISA --
            # tests/test_screen_mx_asm.py for the asm block -- matches the counter only that way).
            mf = val["SQ_INSTS_MFMA"]
            vec = val["SQ_INSTS_VALU"] - mf
            out["mfma_per_launch"] = mf
            out["valu_instr_per_tile"] = vec / mf
            out["vector_issue_busy"] = (4.0 * vec + 8.0 * mf) / 1024.0 / cycles
            out["matrix_pipe_busy"] = 32.0 * mf / 1024.0 / cycles
            # issue slots of the launch per SIMD (an instruction count: the same in every run); bench.py divides the launch
            # time it measures live by it.  A stream of nothing but v_min3_i32 on every SIMD of the chip runs at 2.20 ns
            # (3 waves per SIMD) / 2.38 ns (2 waves) per instruction: profiles/r3_ubench_mfma16c.txt
            out["issue_slots_per_simd"] = (vec + 2.0 * mf) / 1024.0
            out["issue_slot_ns_min3_stream"] = {"waves_per_simd_2": 38.11 / 16.0, "waves_per_simd_3": 35.21 / 16.0,
                                                "source": "profiles/r3_ubench_mfma16c.txt"}
        # issue slots of the launch (an instruction count, the same in every run): SQ_INSTS_VALU counts every vector
        # instruction including the MFMAs, an MFMA takes a second slot
        out["issue_slots_per_launch"] = val["SQ_INSTS_VALU"] + val.get("SQ_INSTS_MFMA", 0.0)
        out["candidates_per_launch"] = PMC_CANDIDATES.get(workload)
    return out


def cpu_baseline(cfg, geoms, threads, budget_s=10.0):
    """The CPU oracle (a port of the reference algorithm) timed on this box's host cores on a
    bounded sample of the same workload: the first frame pairs of pullback 0, all candidates,
    candidates evaluated in parallel (OpenMP) like the reference's rayon par_iter."""
    from oracle import oracle as orc
    cores = threads
    g = geoms[0]
    if g.n_frames < 64:   # tiny workloads: still give the baseline a few seconds of work
        import multimoda_rs_amd as _mm
        g = _mm.synthetic_pullback(512, cfg["points"])
    ss = cfg["sample_size"]
    import multimoda_rs_amd as mm
    n_angles = len(mm.search_angles(cfg["step_deg"], cfg["range_deg"])[0])
    done, t_used, pairs = 0, 0.0, 0
    i = 1
    while i < g.n_frames and t_used < budget_s:
        ref = np.concatenate([mm.search_set(g, i - 1, ss), np.zeros((0, 2))])
        tgt = mm.search_set(g, i, ss)
        t0 = time.perf_counter()
        orc.bruteforce_rotation(ref, tgt, cfg["step_deg"], cfg["range_deg"], float(g.centroids[i, 0]),
                                float(g.centroids[i, 1]), n_threads=cores)
        t_used += time.perf_counter() - t0
        done += n_angles
        pairs += 1
        i += 1
    return {"value": done / t_used, "unit": "pose-evals/s", "cores": cores, "kind": "port",
            "sample": f"oracle bruteforce search on the first {pairs} frame pairs of pullback 0 "
                      f"({n_angles} candidates each, N={ss + 20} pts/set), {t_used:.1f} s"}


# The reference's SECOND published benchmark (benchmarks/benchmark_cpu_scaling.py:32-80, docs/benchmark.rst:53-86):
# `from_array_single` on its OCT example pullback -- 280 frames, step 0.01 deg, range +-6 deg (1201 candidates brute force, a
# 135-evaluation ladder otherwise), sample_size = 200, image_center = (5, 5), n_points = 40, write_obj / smooth off.  The OCT
# contours themselves are not in the reference checkout (examples/data/oct_single holds only oct_ref.csv: frame 280,
# (6, 9), z = 56 -> 0.2 mm frame spacing), so the pullback is synthetic: 280 frames x 360 points around (5, 5), 1 deg
# torsion walk -> 200 lumen + ceil(40 * 200 / 360) = 23 catheter points = 223 points per set (7 x 7 tiles).
OCT = dict(frames=280, points=360, step_deg=0.01, range_deg=6.0, sample_size=200, image_center=(5.0, 5.0), n_points=40,
           published_s={"bruteforce": 14.15, "optimized": 2.40, "hardware": "Xeon Gold 6234, 16 threads (docs/benchmark.rst:53-86)"})


def oct_pullback(mm, frames=None):
    return mm.synthetic_pullback(frames or OCT["frames"], OCT["points"], pullback_id=0, seed=4321, image_center=OCT["image_center"],
                                 n_catheter=OCT["n_points"], torsion_sigma_deg=1.0)


def oct_input_data(mm, g):
    """The (N, 4) [frame, x, y, z] arrays of numpy_to_inputdata, as the reference's benchmark builds them from its CSVs."""
    frame = np.repeat(g.orig_frames.astype(np.float64), np.diff(g.lumen_off))
    ref_i = int(np.nonzero(g.has_ref)[0][0])
    return mm.numpy_to_inputdata(np.concatenate([frame[:, None], g.lumen], axis=1),
                                 np.concatenate([[float(g.orig_frames[ref_i])], g.ref[ref_i]]), True, label="oct")


def oct_single_leg(mm, eng, precision, repeats=5, oracle_frames=64, cpu_threads=16):
    """Both modes of that benchmark, end to end through `mm.from_array_single` (InputData arrays -> geometry builder ->
    search -> chain walk -> post-steps) and the search alone (mm.WithinPlan on the built geometry: stage, search, walk),
    with the kernel time of the screen (hipEvents) and the first 63 chain steps re-done by the CPU oracle."""
    import statistics
    g0 = oct_pullback(mm)
    data = oct_input_data(mm, g0)
    kw = dict(step_rotation_deg=OCT["step_deg"], range_rotation_deg=OCT["range_deg"], sample_size=OCT["sample_size"],
              image_center=OCT["image_center"], n_points=OCT["n_points"], write_obj=False, smooth=False, engine=eng)
    n_set = min(OCT["sample_size"], OCT["points"]) + -(-OCT["n_points"] * OCT["sample_size"] // OCT["points"])
    out = {"workload": (f"from_array_single, synthetic OCT-like pullback: {OCT['frames']} frames x {OCT['points']} pts, step "
                        f"{OCT['step_deg']} deg x +-{OCT['range_deg']} deg, sample_size {OCT['sample_size']}, n_points {OCT['n_points']} "
                        f"-> {n_set} pts/set; write_obj / smooth off (benchmarks/benchmark_cpu_scaling.py:32-80)"),
           "points_per_set": n_set,
           "reference_published_seconds": OCT["published_s"],
           "note": "the reference's OCT contours are not in its checkout (only oct_ref.csv): synthetic data of the published shape; "
                   "the published seconds are another machine's and the reference's own data -- context, not a baseline"}
    for name, brute in (("bruteforce", True), ("optimized", False)):
        mm.from_array_single(data, bruteforce=brute, **kw)                                  # warm-up
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            g_api, logs_api = mm.from_array_single(data, bruteforce=brute, **kw)
            ts.append(time.perf_counter() - t0)
        # the search alone, through the precision of the headline
        ss, evals, kms, launches = [], 0, 0.0, 0
        for rep in range(repeats + 1):
            case = [g0.copy()]
            eng.synchronize()
            if rep:
                eng.profile(True)
            t0 = time.perf_counter()
            plan = mm.WithinPlan(eng, case, OCT["step_deg"], OCT["range_deg"], brute, OCT["sample_size"], precision=precision)
            logs, evals, unresolved = plan.run()
            dt_ = time.perf_counter() - t0
            plan.close()
            if rep:
                pr = eng.profile_read()
                eng.profile(False)
                ss.append(dt_); kms += pr["ms"]; launches += pr["launches"]
        same = bool(list(logs[0]) == list(logs_api))          # (the entry point goes on to the post-steps: compare the logs)
        out[name] = {"from_array_single_ms": 1e3 * statistics.median(ts), "search_alone_ms": 1e3 * statistics.median(ss),
                     "pose_evals_per_call": int(evals), "pose_evals_per_s_search_alone": evals / statistics.median(ss),
                     "screen_kernel_ms_per_call": kms / repeats, "screen_launches_per_call": launches / repeats,
                     "chain_steps_researched_on_chain_state": int(unresolved),
                     "search_alone_logs_identical_to_from_array_single": same,
                     "published_s": OCT["published_s"][name]}
        # the oracle's chain on the first 64 frames (a chain's first steps do not depend on the rest)
        from oracle import oracle as orc
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import to_oracle  # type: ignore
        head = oct_pullback(mm, oracle_frames)
        same_inputs = bool(np.array_equal(head.lumen, g0.lumen[:head.lumen.shape[0]]))
        ol = orc.align_within_chain(to_oracle(orc, head), OCT["step_deg"], OCT["range_deg"], brute, OCT["sample_size"], n_threads=cpu_threads)
        out[name]["identical_to_oracle_first_63_chain_steps"] = bool(same_inputs and list(logs[0])[:oracle_frames - 1] == list(ol))
    sst = eng.screen_stats()
    out["kernel"] = "mm::k_screen_mx<7, false> (223 pts/set = 7 x 7 tiles of 32 x 32; the bounded search behind from_array_single " \
                    "screens its survivors and small levels with the same kernel)"
    out["screen_stats_of_this_engine"] = sst
    return out


def dominant_launch(ms, pair_evals, peak=FP32_VECTOR_PEAK_TFLOPS):
    """The launches that carry the work (>= half of the largest launch's pair-distances: the one
    within-stage launch of every step): their count, mean device time and algorithmic rate -- the
    figures to hold against the per-dispatch rows of the committed rocprofv3 kernel trace."""
    if len(ms) == 0:
        return None
    big = pair_evals >= 0.5 * pair_evals.max()
    t = float(ms[big].mean())
    pe = float(pair_evals[big].mean())
    tf = pe * FLOPS_PER_PAIR_EVAL / (t * 1e-3) * 1e-12
    return {"launches": int(big.sum()), "avg_ms": t, "pair_distance_evals": pe, "tflops": tf,
            "frac": tf / peak}


FP64_VECTOR_PEAK_TFLOPS = 78.6    # MI355X_MICROARCH.md, Peak FP64 (vector)


class Runner:
    """The three pieces of a step for one precision: stage(k) (raw pullbacks -> HBM, search sets built on the
    device, level 0 staged), search(k) (all levels: local search -> exchange -> commit) and finish(k) (chain walk,
    AB|CD and AC|BD between alignments).  Step k lives on engine k % len(engs) and works on its own copy of the case."""

    def __init__(self, mm, engs, base, cfg, prec, mode, rank, world, ext, n_cases, rehearse=0, grid=None, comm=None, lazy_until=0, bruteforce=True):
        self.bruteforce = bruteforce
        self.mm, self.engs, self.cfg, self.prec, self.mode, self.rank, self.world, self.ext = mm, engs, cfg, prec, mode, rank, world, ext
        self.rehearse = rehearse           # > 1: this process plays rank 0 of `rehearse` ranks without peers (timing only)
        self.grid = grid                   # (pair_blocks, cand_slices) of the shard grid (N > 1 / rehearsal)
        self.comm = comm                   # rehearsal: the library's world = 1 RCCL communicator (None: torch's)
        self.fin_group = None              # N > 1: the process group of the finishing thread's broadcasts (its own communicator)
        # the caller's input data: one fresh copy of the case per step (made before the timed region -- this is
        # the data a caller hands over, not work of the step)
        # (lazy_until: the cases of the untimed clock-ramp steps are copied when they are staged and dropped when they
        # are finished -- at N = 8 there are dozens of them)
        self.base, self.lazy_until = base, lazy_until
        self.cases = [base if ext is not None else (None if k < lazy_until else [g.copy() for g in base]) for k in range(n_cases)]
        self.plans = [None] * n_cases
        self.stage_s = 0.0
        self.staged = 0
        if ext is not None:
            # the EXTENSION grid keeps round 1's protocol: its (Python-built) sets are staged before the timed region
            for k in range(n_cases - 2):
                self.plans[k] = mm.ShiftRotationSearch(engs[0], self.cases[k], ext[0], ext[1], cfg["step_deg"],
                                                       cfg["range_deg"], cfg["sample_size"], precision=prec)

    def stage(self, k):
        if k >= len(self.cases) or self.mode != 1 or self.ext is not None:
            return
        mm, cfg = self.mm, self.cfg
        if self.cases[k] is None:
            self.cases[k] = [g.copy() for g in self.base]
        t0 = time.perf_counter()
        self.plans[k] = mm.WithinPlan(self.engs[k % len(self.engs)], self.cases[k], cfg["step_deg"], cfg["range_deg"],
                                      self.bruteforce, cfg["sample_size"], precision=self.prec,
                                      shard=(self.rank, *self.grid) if self.world > 1 else
                                      ((0, *self.grid) if self.rehearse > 1 else None))
        if self.rehearse > 1 and self.comm is not None:
            self.plans[k].set_timing_rehearsal(True)     # a world = 1 communicator serves rank 0 of a larger job: timing only
        self.stage_s += time.perf_counter() - t0
        self.staged += 1

    def chained(self):
        """N > 1 (or its rehearsal) over the library's communicator: step k+1's launch is ordered behind step k's whole
        exchange (Engine.wait_exchange), so no collective runs beside a launch and the device never waits for the host."""
        return self.native_comm() is not None and not os.environ.get("MM_BENCH_SHARD_LOOKAHEAD")

    def native_comm(self):
        if self.rehearse > 1:
            return self.comm
        if self.world > 1:
            from multimoda_rs_amd import distributed as D
            return D.native_comm() if D.exchange_mode() == "rccl" else None
        return None

    def pre(self, k):
        if self.plans[k] is not None and self.ext is None and self.chained():
            self.plans[k].search_sharded_begin(self.native_comm())

    def begin(self, k, prev):
        """Enqueue step k's level-0 launch behind step prev's long kernel (Engine.wait_search) -- sharded: behind step
        prev's whole exchange (Engine.wait_exchange) -- and return."""
        if self.plans[k] is not None and self.ext is None:
            if self.chained():
                if prev is not None:
                    self.engs[k % len(self.engs)].wait_exchange(self.engs[prev % len(self.engs)])
                self.plans[k].level_launch(0)
                self.plans[k]._begun = True
            else:
                if os.environ.get("MM_BENCH_NO_WAIT"):       # experiment: the look-ahead launch does not wait for the previous step's
                    prev = None
                self.plans[k].search_begin(after=None if prev is None else self.engs[prev % len(self.engs)])

    def search(self, k):
        if self.plans[k] is not None and self.ext is None:
            if getattr(self.plans[k], "_begun", False) and self.world == 1 and self.rehearse <= 1:
                self.plans[k].search_end()               # collect level 0, remaining levels, commit
            elif self.rehearse > 1:
                if self.comm is not None:
                    self.plans[k].search_sharded(self.comm)  # the N > 1 code path over the library's world = 1 communicator
                else:
                    from multimoda_rs_amd import distributed as D
                    D.search_device(self.plans[k])       # the same through torch's world = 1 RCCL group (the checker)
            else:
                self.plans[k].search()                   # levels: local search -> exchange -> commit

    def finish(self, k):
        mm, cfg = self.mm, self.cfg
        if self.ext is not None:
            r = self.plans[k].run()
            out = (r["winners"], None, self.plans[k].pose_evals, 0)
        elif self.plans[k] is None:
            out = full_alignment(mm, self.engs[0], self.cases[k], cfg, None, self.prec)
        elif self.sharded_finish():
            # N > 1: pullback g is walked on rank g mod N, pair k of a between batch aligned on rank k mod N, and what they
            # changed is broadcast from the owner (multimoda_rs_amd.distributed.walk_sharded / align_between_sharded): every
            # rank ends up with the whole alignment without repeating the whole finish.  Rehearsal: this process plays rank 0
            # without peers -- its share of the finish, nothing exchanged.
            from multimoda_rs_amd import distributed as D
            rk, wd = (self.rank, self.world) if self.world > 1 else (0, self.rehearse)
            kw = dict(group=self.fin_group, rank=rk, world=wd, exchange=self.world > 1)
            logs, ev, unres = D.walk_sharded(self.plans[k], **kw)
            eng = self.engs[k % len(self.engs)]
            a, b, c, d = self.cases[k]
            r1, e1 = D.align_between_sharded(eng, [(a, b), (c, d)], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], self.prec, **kw)
            r2, e2 = D.align_between_sharded(eng, [(a, c), (b, d)], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], self.prec, **kw)
            out = (logs, np.concatenate([r1, r2]), ev + e1 + e2, unres)
        else:
            logs, ev, unres = self.plans[k].walk()
            rot, e2 = between_stage(mm, self.engs[k % len(self.engs)], self.cases[k], cfg, self.prec)
            out = (logs, rot, ev + e2, unres)
        if self.plans[k] is not None:
            self.plans[k].close()                        # HBM of the step is released; at most three plans are alive
            self.plans[k] = None
        if k < self.lazy_until:
            self.cases[k] = None
        return out

    def sharded_finish(self):
        return (self.world > 1 or self.rehearse > 1) and os.environ.get("MM_BENCH_SHARDED_FINISH", "1") != "0"

    def close(self):
        for i, p in enumerate(self.plans):
            if p is not None:
                p.close()
                self.plans[i] = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)   # the first two big launches of a process run at ramping clocks
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="decoupled", choices=["chain", "decoupled"])
    ap.add_argument("--precision", default="matrix", choices=["f32", "fast", "bounded", "f64", "matrix"],
                    help="candidate scoring: matrix = squared distances from the f16 matrix pipe (hi + lo split, fp32 "
                         "accumulate), minima on the vector pipe + exact f64 re-score (default; the same workload through "
                         "the packed-FMA screen, the bounded search and the all-f64 kernel is reported beside it); f32 = "
                         "direct-form f32 screen + exact f64 re-score; fast = expanded-form "
                         "f32 screen + exact f64 re-score; bounded = lower bounds rule candidates out "
                         "before the screen (not pose-evals in SURVEY 8(d)'s sense: for measurements of that path, "
                         "never the headline); f64 = every candidate in exact f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the bounded-search, all-f64 and sequential legs")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline (default: the GPU box's CPU share per GPU, 16, or fewer cores)")
    ap.add_argument("--check", action="store_true", help="verify the result against the CPU oracle (slow)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as a child job (one process per GPU) and pass its
        # exit code on.  This parent never touches the GPU and does not exec.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # stdout carries ONE line, the JSON: whatever libraries write to file descriptor 1 meanwhile (RCCL prints a version
    # banner there when a communicator is created) goes to stderr; the JSON is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist

    # Three or four host threads hand the interpreter lock around; with the default switch interval (5 ms) the thread that
    # launches the next step can wait that long for a thread that is between two native calls -- longer than a whole step
    # at N = 8.  0.2 ms keeps the hand-over far below a step.
    sys.setswitchinterval(float(os.environ.get("MM_BENCH_SWITCH_INTERVAL", "0.0002")))

    import __graft_entry__ as ge
    ge.build()
    import multimoda_rs_amd as mm

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MM_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU (RCCL refuses
    # two ranks on the same device); the driver's multi-GPU runs use nccl (= RCCL over xGMI)
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # the finishing thread's collectives (broadcasts of walked pullbacks) run beside the search thread's all-reduces: a group
    # of their own, so that the two threads' collectives are never ordered against each other
    fin_group = dist.new_group(backend=backend) if world > 1 else None

    # MM_BENCH_REHEARSE_WORLD=N (single process): play rank 0 of N ranks WITHOUT peers -- this rank's share of the
    # candidate axis, the N > 1 code path (no look-ahead, device exchange, both all-reduces over a world = 1 RCCL
    # group).  The result is not an alignment (only 1/N of the candidates are seen); what it measures is the
    # per-rank step time of an N-GPU run short of the xGMI latency of two small all-reduces.  Never a bench line.
    rehearse = int(os.environ.get("MM_BENCH_REHEARSE_WORLD", "0"))
    rehearse_comm = None
    if rehearse > 1:
        if world != 1:
            raise SystemExit("MM_BENCH_REHEARSE_WORLD is a single-process rehearsal")
        if os.environ.get("MM_EXCHANGE", "rccl") == "rccl":
            rehearse_comm = mm.Comm(mm.Comm.unique_id(), 0, 1, local_rank)   # the library's own communicator
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

    cfg = WORKLOADS[args.workload]
    mode = 0 if args.mode == "chain" else 1
    PRECS = {"f32": mm.MM_PRECISION_F32, "fast": mm.MM_PRECISION_F32_FAST, "bounded": mm.MM_PRECISION_F32_BOUNDED,
             "f64": mm.MM_PRECISION_F64, "matrix": mm.MM_PRECISION_F32_MATRIX}
    PREC = PRECS[args.precision]
    base = mm.synthetic_case(cfg["frames"], cfg["points"])
    # N > 1: the (frame pair x candidate) grid is cut into pair_blocks x cand_slices tiles, one per rank
    # (multimoda_rs_amd.distributed.shard_grid: frame pairs first, the candidate axis when the pairs are few;
    # MM_SHARD_GRID=PxC overrides, 1xN = the pure candidate-axis split)
    from multimoda_rs_amd import distributed as D
    n_jobs = sum(g.n_frames - 1 for g in base)
    grid = D.shard_grid(max(world, rehearse, 1), n_jobs)
    # three engines (main stream, side stream, staging buffers each): step k lives on engine k % 3, so the search of
    # step k+1, the finish of step k and the staging of step k+3 (which takes over step k's engine once that is
    # finished) never share one, and each of the three has a host thread of its own
    # (N > 1: five, and two finishing threads -- a step is an eighth as long there: finish(k) -> stage(k + LOOK) has to fit
    # into LOOK - 1 steps, and with two steps being finished at once two more engines are taken)
    LOOK = int(os.environ.get("MM_BENCH_ENGINES", "5" if (world > 1 or rehearse > 1) else "3"))
    if LOOK < 2:
        raise SystemExit("MM_BENCH_ENGINES must be >= 2")
    # The bounded search resolves a case in ~1.3 ms of device time, less than one host thread needs to stage a case
    # (1.4 ms: 25.6 MB through pinned memory and PCIe, level 0's descriptors) or to finish one (1.3 ms: chain walk, two
    # between batches): its pipeline is six engines deep with two finishing threads (measured 3/1: 1.8 - 2.1 ms per step,
    # 5/2: 1.60 - 1.64, 6/2: 1.56, 8/2: 1.48; MM_BENCH_BOUNDED_ENGINES / MM_BENCH_FINISHERS)
    LOOK_BOUNDED = max(LOOK, int(os.environ.get("MM_BENCH_BOUNDED_ENGINES", "6")))
    n_engines = LOOK_BOUNDED if (args.precision == "bounded" or not args.no_extra_legs) and world == 1 and rehearse <= 1 else LOOK
    engs_all = [mm.Engine(local_rank) for _ in range(n_engines)]
    engs = engs_all[:LOOK]
    if os.environ.get("MM_BENCH_BOUND_MATRIX"):      # A/B: 0 = the bounded search on the packed-FMA kernels of rounds 1-3; 11/12/21/22 = variant
        for e_ in engs_all:
            e_.set_bound_matrix(int(os.environ["MM_BENCH_BOUND_MATRIX"]))
    ext = cfg.get("shift")
    pipelined = mode == 1 and ext is None and not os.environ.get("MM_BENCH_SEQUENTIAL")
    STAGER = LOOK >= 3

    # N > 1: the candidate axis is sharded -- rank r scores candidates [n*r/N, n*(r+1)/N) of every frame pair;
    # per-shard bests are all-reduced over RCCL on the device (multimoda_rs_amd.distributed) and every rank then
    # walks the chain and runs the small between stage (strong scaling).
    if world > 1 and (mode != 1 or ext is not None):
        raise SystemExit("--gpus N > 1 shards the decoupled 4-phase search; use --mode decoupled on config2/config3")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for e in engs:
            e.synchronize()

    def read_profiles(engs=None):
        ms, pe, tot, bound_ = [], [], {"launches": 0, "ms": 0.0, "pair_evals": 0.0, "candidates": 0}, None
        for e in (engs_main if engs is None else engs):
            a, b = e.profile_launches()
            ms.append(a); pe.append(b)
            bs = e.bound_stats()
            bound_ = bs if bound_ is None else {k: bound_[k] + bs[k] for k in bs}
            p = e.profile_read()
            for k in tot:
                tot[k] += p[k]
            e.profile(False)
        return np.concatenate(ms), np.concatenate(pe), tot, bound_

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item())

    LOOK_MAIN, engs_main = LOOK, engs

    def timed_leg(prec, warmup, steps, pipe, resident=False, cfg=cfg, bruteforce=True, deep=False):
        """W untimed + K timed steps of one precision.  Inside the timed region: K stagings (raw pullbacks ->
        HBM -> search sets), K searches, K finishes.  The pipeline is primed before it (the first LOOK cases are
        staged during set-up / warm-up), so the stagings in the region are those of steps W+LOOK .. W+K+LOOK-1:
        steady-state throughput of a stream of cases, the last LOOK staged cases are not searched.
        resident: every case is staged before the timed region (inputs and search sets resident in HBM when it
        starts); the region then holds K searches and K finishes only."""
        # The clocks need ~0.2 s of load to settle (the first launches after a pause run up to 20 % slower), and
        # at N = 8 a step is 4 ms: the untimed warm-up is extended to that, by a count every rank computes alike
        # (pose-evals of a step / 45 M/s / ranks -- an estimate of the step time, not a measurement).
        ramp = 0
        if mode == 1 and ext is None and pipe and not resident:
            n_ang = len(mm.search_angles(cfg["step_deg"], cfg["range_deg"])[0])
            est_ms = n_jobs * n_ang / 45e6 / max(world, rehearse, 1) * 1e3 * (3.0 if prec == mm.MM_PRECISION_F64 else 1.0)
            ramp = max(0, min(96, int(np.ceil(200.0 / max(est_ms, 0.05))) - warmup))
            if prec == mm.MM_PRECISION_F32_BOUNDED:
                ramp = 0
        warmup += ramp
        n_total = warmup + steps
        # deep: the bounded search's pipeline (see LOOK_BOUNDED above)
        deep = deep and len(engs_all) > len(engs_main) and pipe
        LOOK, engs = (len(engs_all), engs_all) if deep else (LOOK_MAIN, engs_main)
        STAGER = LOOK >= 3
        r = Runner(mm, engs, base, cfg, prec, mode, rank, world, ext, n_total + LOOK, rehearse, grid, rehearse_comm,
                   lazy_until=warmup - 1 if ramp else 0, bruteforce=bruteforce)
        for k in range(n_total if resident else LOOK):
            r.stage(k)                                   # priming (setup, untimed)
        stage_fn = None if resident else r.stage
        # N > 1 (opt-in, MM_BENCH_SHARD_LOOKAHEAD=1; never measured on a multi-GPU box): the same look-ahead -- step
        # k+1's local launch is queued behind step k's long kernel and runs beside step k's re-score, exports and
        # all-reduces.  The order of the collectives is unchanged on every rank.
        sharded = world > 1 or rehearse > 1
        begin = r.begin if (pipe and mode == 1 and ext is None and not os.environ.get("MM_BENCH_NO_LOOKAHEAD")
                            and (not sharded or os.environ.get("MM_BENCH_SHARD_LOOKAHEAD") == "1" or r.chained())) else None
        pre = r.pre if (begin is not None and sharded and r.chained()) else None
        import gc
        gc.collect()
        gc.disable()                                      # keep the interpreter's cyclic GC (tens of ms) out of the steps
        # (collected BEFORE the warm-up: a collection between warm-up and timed region idles the device for ~40 ms,
        # and the first big launch after such a pause runs 34.9 instead of 31.3 ms -- the clocks have dropped)
        # one finishing thread: with the finish sharded over the ranks it is a fraction of a step (and the finishing thread's
        # broadcasts must be issued in one order on every rank); MM_BENCH_SHARDED_FINISH=0: every rank finishes everything,
        # two threads take turns
        r.fin_group = fin_group
        nfin = int(os.environ.get("MM_BENCH_FINISHERS", "2" if ((sharded and not r.sharded_finish()) or deep) else "1"))
        run_steps(range(warmup), r.search, r.finish, pipe, stage_fn, LOOK, begin, STAGER, pre, nfin)
        barrier()
        for e in engs:
            e.profile(True)
        r.stage_s, r.staged = 0.0, 0
        t0 = time.perf_counter()
        results = run_steps(range(warmup, n_total), r.search, r.finish, pipe, stage_fn, LOOK, begin, STAGER, pre, nfin)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        prof = read_profiles(engs)
        last_case = r.cases[n_total - 1]
        stage_ms = 1e3 * r.stage_s / max(r.staged, 1)
        r.close()
        return dict(ramp=ramp, engines=LOOK, finishers=nfin, dt=reduce_max(dt), results=results, evals=sum(x[2] for x in results), unresolved=sum(x[3] for x in results),
                    prof=prof, last_case=last_case, stage_ms=stage_ms, staged=r.staged)

    def ladder_leg():
        """The reference's DEFAULT search (bruteforce=False: coarse -> fine, dependent levels, align_within.rs:193-247) on the
        same pullbacks with its default grid (0.5 deg, +-90 deg: a 1 deg pass, then +-5 deg at 0.5 deg = 202 evaluations per
        frame pair): whole steps like the headline (stage, search: two dependent launches, chain walk, between alignments)."""
        try:
            lcfg = dict(cfg, step_deg=0.5, range_deg=90.0)
            k = max(args.steps, 10)
            leg = timed_leg(PREC, 2, k, pipelined, cfg=lcfg, bruteforce=False)
            lms, lpe, lprof, _ = leg["prof"]
            dom = dominant_launch(lms, lpe)
            # the oracle's ladder chain on the first 64 frames of pullback 0 (a chain's first steps do not depend on the rest)
            from oracle import oracle as orc
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import to_oracle  # type: ignore
            head = mm.synthetic_pullback(64, cfg["points"], pullback_id=0)
            same_inputs = bool(np.array_equal(head.lumen, base[0].lumen[:head.lumen.shape[0]]))
            ol = orc.align_within_chain(to_oracle(orc, head), 0.5, 90.0, False, cfg["sample_size"], n_threads=min(16, os.cpu_count() or 1))
            got = list(leg["results"][-1][0][0])[:63]
            return {"value": leg["evals"] / leg["dt"], "unit": "pose-evals/s (the reference's evaluation count: 202 per frame pair)",
                    "ms_per_step": leg["dt"] / k * 1e3, "steps": k, "pose_evals_per_step": leg["evals"] // k,
                    "grid": "bruteforce=False, 0.5 deg x +-90 deg (levels: 1 deg x +-90, 0.5 deg x +-5 around the level-1 winner)",
                    "chain_steps_researched_on_chain_state": leg["unresolved"],
                    "dominant_launch": dom,
                    "kernel": ("mm::k_screen_mx" if args.precision == "matrix" else "mm::k_screen_fast<33, false>") + " (level 0: 181 candidates per pair)",
                    "identical_to_oracle_first_63_chain_steps_of_pullback_0": bool(same_inputs and got == list(ol)),
                    "note": "two dependent launches per step (the second level's candidate lists depend on the first level's "
                            "winners: one host round trip between them); the full-size oracle comparison runs in "
                            "tests/test_gpu_fullsize.py"}
        except Exception as ex:
            return {"error": f"{type(ex).__name__}: {ex}"}

    def extension_leg():
        """EXTENSION grid (absent from the reference's 4-phase path; BASELINE config 3's "~720 x 100 candidates"):
        every frame against the 100 frames of a +-50 window x 721 rotations, 4 x 512 frames, as extra set pairs through
        the same primitive.  One brute-force step (one launch of ~3 s) and one step through the bounded search."""
        try:
            ecfg = WORKLOADS["config3ext"]
            lo, hi = ecfg["shift"]
            eng = engs[0]
            out = {}
            res = {}
            for name, prec in (("bruteforce", PREC), ("bounded", mm.MM_PRECISION_F32_BOUNDED)):
                ts = time.perf_counter()
                srs = mm.ShiftRotationSearch(eng, base, lo, hi, ecfg["step_deg"], ecfg["range_deg"], ecfg["sample_size"], precision=prec)
                eng.synchronize()
                stage_s = time.perf_counter() - ts           # set construction (host, Python) + staging: reported, see the note
                eng.profile(True)
                t0 = time.perf_counter()
                r = srs.run()
                dt_ = time.perf_counter() - t0
                lms, lpe = eng.profile_launches()
                eng.bound_stats()
                pr = eng.profile_read()
                eng.profile(False)
                res[name] = (r, srs.meta.copy(), srs.pose_evals)
                srs.close()
                if name == "bruteforce":
                    tf = pr["pair_evals"] * FLOPS_PER_PAIR_EVAL / (pr["ms"] * 1e-3) * 1e-12 if pr["ms"] > 0 else 0.0
                    eiss = issue_roofline(committed_pmc("config3", args.precision), lms, lpe, ecfg["sample_size"] + 20, ecfg["sample_size"] + 20)
                    out.update({"value": srs.pose_evals / dt_, "unit": "pose-evals/s", "ms_per_step": dt_ * 1e3, "steps": 1,
                                "staging_ms": stage_s * 1e3, "value_including_staging": srs.pose_evals / (dt_ + stage_s),
                                "pose_evals_per_step": srs.pose_evals, "pairs": int(srs.meta.shape[0]), "candidates_per_pair": len(srs.angles),
                                "roofline": {"bound": "valu-issue", "unit": "G issue-slots/s", "peak": VALU_ISSUE_PEAK_GSLOTS,
                                             **{k_: (eiss or {}).get(k_) for k_ in ("achieved", "frac", "issue_slots_per_launch")},
                                             "algorithmic_tflops_vs_fp32_vector": {"achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS,
                                                                                   "ratio": tf / FP32_VECTOR_PEAK_TFLOPS},
                                             "kernel": "mm::k_screen_mx" if args.precision == "matrix" else "mm::k_screen_fast<33, false>",
                                             "launches": pr["launches"],
                                             "avg_launch_ms": pr["ms"] / max(pr["launches"], 1)}})
                else:
                    out["bounded"] = {"candidates_resolved_per_s": srs.pose_evals / dt_, "ms_per_step": dt_ * 1e3, "staging_ms": stage_s * 1e3}
            rb, meta, _ = res["bruteforce"]
            rq = res["bounded"][0]
            out["bounded"]["identical_to_bruteforce_result"] = bool(
                np.array_equal(rb["best_idx"], rq["best_idx"]) and np.array_equal(rb["best_cost"], rq["best_cost"]) and rb["winners"] == rq["winners"])
            # at shift 0 a pair is a step of the reference's chain: its winner must be the headline's log entry
            logs = results[-1][0]
            ok = True
            for p in np.nonzero(meta[:, 2] == 0)[0]:
                gi, i = int(meta[p, 0]), int(meta[p, 1])
                ok = ok and (rb["best_angle"][p] * (180.0 / np.pi) == logs[gi][i - 1][2])
            out["shift0_winners_identical_to_headline_logs"] = bool(ok)
            out["note"] = ("EXTENSION axis: not part of the reference's 4-phase path and never folded into `value`; the point sets are built "
                           "(host, Python) and staged before the timed call -- that time is `staging_ms`, and `value_including_staging` "
                           "counts it; parity of shifted pairs is against the "
                           "oracle's metric / search (tests/test_gpu_fullsize.py)")
            return out
        except Exception as ex:
            return {"error": f"{type(ex).__name__}: {ex}"}

    # setup, not a step: let both engines grow their transient buffers (between stage) now
    if ext is None:
        for e in engs:
            between_stage(mm, e, [g.copy() for g in base], cfg, PREC)

    # N > 1, first thing: one case through every exchange -- the library's own RCCL communicator (the default on an
    # nccl group), torch's all-reduces on the same device records, and the host gather + mm_merge_shards -- all three
    # must give the same alignment on this rank (the digest below compares the ranks).  If the library's communicator
    # cannot be set up the run goes on over torch's (said in the JSON line), it does not stop.
    exchange_note = None
    if world > 1:
        def one_case(mode_):
            os.environ["MM_EXCHANGE"] = mode_
            case = [g.copy() for g in base]
            pl = mm.WithinPlan(engs[0], case, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"], precision=PREC,
                               shard=(rank, *grid))
            pl.search()
            logs_, ev_, un_ = pl.walk()
            pl.close()
            return [list(l) for l in logs_], ev_, un_
        # the library's default is torch's all-reduces (`device`); this run opts into the library's own RCCL communicator on an
        # nccl group BECAUSE it cross-checks the three exchanges right here, before anything is timed
        chosen = os.environ.get("MM_EXCHANGE", "") or ("rccl" if backend == "nccl" else "device")
        modes = ["gather", "device"] + (["rccl"] if chosen == "rccl" else [])
        got = {}
        for m in modes:
            try:
                got[m] = one_case(m)
            except Exception as ex:
                if m != "rccl":
                    raise
                # every rank must take the same decision: any rank's failure sends all ranks to torch's exchange
                got[m] = None
                exchange_note = f"library RCCL communicator unavailable ({type(ex).__name__}: {ex}); torch.distributed all-reduces used"
        flag = torch.tensor([0 if got.get("rccl", 1) is not None else 1], dtype=torch.int32,
                            device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and chosen == "rccl":
            chosen = "device"
            exchange_note = exchange_note or "library RCCL communicator unavailable on another rank; torch.distributed all-reduces used"
        ref_ = got["gather"]
        for m, v in got.items():
            if v is not None and v != ref_:
                raise SystemExit(f"exchange '{m}' and exchange 'gather' disagree on rank {rank}")
        os.environ["MM_EXCHANGE"] = chosen

    main_leg = timed_leg(PREC, args.warmup, args.steps, pipelined, deep=args.precision == "bounded")
    if rehearse > 1:
        full = cfg["frames"] and sum(g.n_frames - 1 for g in base) * len(mm.search_angles(cfg["step_deg"], cfg["range_deg"])[0])
        ms = main_leg["dt"] / args.steps * 1e3
        kms, kpe = main_leg["prof"][0], main_leg["prof"][1]
        big = kpe >= 0.5 * kpe.max() if len(kpe) else np.zeros(0, bool)
        emit(({"rehearsal": f"rank 0 of {rehearse} without peers (MM_BENCH_REHEARSE_WORLD): timing only, NOT an alignment",
                          "shard_grid": {"pair_blocks": grid[0], "cand_slices": grid[1]},
                          "exchange": "library RCCL communicator (world = 1)" if rehearse_comm is not None else "torch.distributed (world = 1)",
                          "per_rank_ms_per_step": ms, "steps": args.steps,
                          "dominant_launch_ms": float(kms[big].mean()) if len(kms) else None,
                          "stage_ms_per_case": main_leg["stage_ms"],
                          "within_pose_evals_of_the_full_grid": int(full),
                          "projected_pose_evals_per_s": full / (ms * 1e-3),
                          "note": "projected = the full grid's within pose-evals / this rank's step time: what N such ranks "
                                  "deliver if the two all-reduces cost what they cost at world = 1"}))
        if rehearse_comm is not None:
            rehearse_comm.close()
        else:
            dist.destroy_process_group()
        for e in engs:
            e.close()
        return
    dt, results, evals, unresolved = main_leg["dt"], main_leg["results"], main_leg["evals"], main_leg["unresolved"]
    launch_ms, launch_pe, prof, bound = main_leg["prof"]
    res = results[-1]
    # every step aligns a copy of the same case: all K results (logs, between rotations, pose-eval and re-search counts)
    # must be the same -- a race in the pipeline (three threads, three engines) would show here first
    steps_identical = all(list(r[0]) == list(res[0]) and np.array_equal(r[1], res[1]) and r[2:] == res[2:] for r in results)
    if not steps_identical:
        what = []
        for i, r in enumerate(results):
            d = [n for n, same in (("within logs", list(r[0]) == list(res[0])), ("between rotations", np.array_equal(r[1], res[1])),
                                   ("pose-eval / re-search counts", r[2:] == res[2:])) if not same]
            if d:
                what.append(f"step {i}: {', '.join(d)}" + (f" {r[2:]} vs {res[2:]}" if "pose-eval / re-search counts" in d else ""))
        import hashlib
        sig = [hashlib.sha1(repr((list(r[0]), None if r[1] is None else np.asarray(r[1]).tolist(), r[2:])).encode()).hexdigest()[:8] for r in results]
        raise SystemExit("the timed steps disagree with each other: identical inputs gave different alignments (" + "; ".join(what[:6]) +
                         f"); result signatures per step: {sig}")
    if args.precision != "bounded":
        bound = None

    def same_result(leg):
        r2 = leg["results"][-1]
        return bool(list(r2[0]) == list(res[0]) and np.array_equal(r2[1], res[1]) and
                    all(np.array_equal(g.lumen, h.lumen) and np.array_equal(g.cath, h.cath) and np.array_equal(g.centroids, h.centroids)
                        for g, h in zip(leg["last_case"], main_leg["last_case"])))

    extra = {}
    # (N = 1 only: at N > 1 the line is the scaling measurement and nothing else is put between it and the driver)
    if args.precision in ("fast", "matrix") and ext is None and mode == 1 and not args.no_extra_legs and world == 1:
        if args.precision == "matrix":
            # The packed-FMA screen (MM_PRECISION_F32_FAST), the headline kernel of rounds 1-2, on the same steps
            try:
                leg = timed_leg(mm.MM_PRECISION_F32_FAST, 1, args.steps, pipelined)
                lms, lpe, lprof, _ = leg["prof"]
                extra["fast_screen"] = {"value": leg["evals"] / leg["dt"], "unit": "pose-evals/s", "ms_per_step": leg["dt"] / args.steps * 1e3,
                                        "steps": args.steps, "identical_to_headline_result": same_result(leg),
                                        "kernel": "mm::k_screen_fast<33, false>", "dominant_launch": dominant_launch(lms, lpe),
                                        # the same scale as the headline's `roofline`: issue slots per second against 614.4 G
                                        "issue": issue_roofline(committed_pmc(args.workload, "fast"), lms, lpe,
                                                                cfg["sample_size"] + 20, cfg["sample_size"] + 20),
                                        "note": "expanded-form f32 screen on the vector pipe alone: 2.5 instructions per distance, "
                                                "issue-saturated (4.02 clk per wave-instruction, profiles/r3_config3_fast_pmc_summary.csv)"}
            except Exception as ex:
                extra["fast_screen"] = {"error": f"{type(ex).__name__}: {ex}"}
        # The same workload through MM_PRECISION_F32_BOUNDED (lower bounds rule most candidates out before the
        # screen; winners identical).  Reported beside the headline, never as `value`: a candidate that is ruled
        # out is resolved, not evaluated, so these are not pose-evals in SURVEY 8(d)'s sense.
        try:
            kb = max(args.steps, 200)     # a step is ~1.5 ms: enough of them that fill, drain and the clock ramp do not dominate
            leg = timed_leg(mm.MM_PRECISION_F32_BOUNDED, 40, kb, pipelined, deep=True)
            extra["bounded_search"] = {
                "candidates_resolved_per_s": leg["evals"] / leg["dt"], "ms_per_step": leg["dt"] / kb * 1e3, "steps": kb,
                "pipeline": {"engines": leg["engines"], "finishing_threads": leg["finishers"]},
                "identical_to_bruteforce_result": same_result(leg), "counts": leg["prof"][3],
                "note": "MM_PRECISION_F32_BOUNDED on the same workload and steps: every candidate of the grid is either "
                        "ruled out by a lower bound of its Hausdorff distance or evaluated; same winners, logs and "
                        "coordinates as the brute-force run above (compared here)"}
        except Exception as ex:   # an extra leg must never take the headline line down with it
            extra["bounded_search"] = {"error": f"{type(ex).__name__}: {ex}"}
        # Every candidate in exact f64, the reference's own arithmetic (MM_PRECISION_F64): the number comparable
        # with the reference's f64 path; priced against the fp64 vector peak.
        try:
            k64 = max(2, min(args.steps, 3))
            leg = timed_leg(mm.MM_PRECISION_F64, 1, k64, pipelined)
            p64 = leg["prof"][2]
            tf = p64["pair_evals"] * FLOPS_PER_PAIR_EVAL / (p64["ms"] * 1e-3) * 1e-12 if p64["ms"] > 0 else 0.0
            extra["f64_exact"] = {"value": leg["evals"] / leg["dt"], "unit": "pose-evals/s", "steps": k64,
                                  "ms_per_step": leg["dt"] / k64 * 1e3, "dtype": "f64",
                                  "identical_to_headline_result": same_result(leg),
                                  "roofline": {"bound": "valu-fp64", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                               "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                                               "kernel": "mm::k_search<double,11,16,true,true,3>",
                                               "avg_launch_ms": p64["ms"] / max(p64["launches"], 1)}}
        except Exception as ex:
            extra["f64_exact"] = {"error": f"{type(ex).__name__}: {ex}"}
        # The headline with every case staged before the timed region: inputs and search sets resident in HBM when
        # it starts, K searches + K finishes inside (round 1's protocol).  The headline `value` starts from HOST
        # memory instead and hides the staging behind the previous step's search; this leg shows what that costs.
        try:
            leg = timed_leg(PREC, args.warmup, args.steps, pipelined, resident=True)
            extra["inputs_resident"] = {"value": leg["evals"] / leg["dt"], "unit": "pose-evals/s",
                                        "ms_per_step": leg["dt"] / args.steps * 1e3, "steps": args.steps,
                                        "identical_to_headline_result": same_result(leg),
                                        "note": "all cases staged (raw pullbacks in HBM, search sets built) before the timed "
                                                "region; K searches + K finishes timed"}
        except Exception as ex:
            extra["inputs_resident"] = {"error": f"{type(ex).__name__}: {ex}"}
        # The headline steps one after the other (stage -> search -> finish, nothing overlapped)
        try:
            leg = timed_leg(PREC, 1, 3, False)
            extra["sequential"] = {"value": leg["evals"] / leg["dt"], "ms_per_step": leg["dt"] / 3 * 1e3,
                                   "stage_ms_per_step": leg["stage_ms"]}
        except Exception as ex:
            extra["sequential"] = {"error": f"{type(ex).__name__}: {ex}"}
        if args.workload == "config3":
            extra["ladder_default"] = ladder_leg()
            extra["extension_grid"] = extension_leg()
            try:
                extra["oct_single"] = oct_single_leg(mm, engs[0], PREC, cpu_threads=args.cpu_threads or 16)
            except Exception as ex:
                extra["oct_single"] = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        na = nb = cfg["sample_size"] + 20
        kern_s = prof["ms"] * 1e-3
        achieved_tflops = prof["pair_evals"] * FLOPS_PER_PAIR_EVAL / kern_s * 1e-12 if kern_s > 0 else 0.0
        algo_gbs = prof["candidates"] * BYTES_PER_POSE_EVAL(na, nb) / kern_s * 1e-9 if kern_s > 0 else 0.0
        f64_main = args.precision == "f64"
        pmc = committed_pmc(args.workload, args.precision)
        iss = issue_roofline(pmc, launch_ms, launch_pe, na, nb)
        peak = FP64_VECTOR_PEAK_TFLOPS if f64_main else FP32_VECTOR_PEAK_TFLOPS
        # executed VALU lane-operations per squared distance the reference counts twice (2 x 6 = 12 algorithmic FLOP):
        # fast screen 4 packed-FMA lanes (8 FLOP) + 1 add + 2 min -> 7 issue slots; direct form 8; f64 kernel 7
        exec_per_12 = {"fast": 7.0, "f32": 8.0, "bounded": None, "f64": 7.0, "matrix": None}[args.precision]
        out = {
            "metric": "Hausdorff pose-evals/sec (frames x poses) for 4-phase full align; best-pose match",
            "value": evals / dt,
            "unit": "pose-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"matrix": "f16 hi+lo matrix-pipe screen (fp32 accumulate) + f64 exact re-score",
                      "f32": "f32 screen (direct form) + f64 exact re-score", "fast": "f32 screen (expanded form) + f64 exact re-score",
                      "bounded": "f32 lower bound + f32 screen (expanded form) of the survivors + f64 exact re-score",
                      "f64": "f64"}[args.precision],
            "data": "synthetic",
            "config": {"workload": (f"{args.workload}: EXTENSION (not in the reference's 4-phase path) rotation x frame-shift "
                                    f"grid, 4 pullbacks x {cfg['frames']} frames x {cfg['points']} pts, shifts {ext[0]}..{ext[1]} x "
                                    f"{cfg['step_deg']} deg x +-{cfg['range_deg']} deg") if ext is not None else
                                   f"{args.workload}: full 4-phase alignment, 4 pullbacks x {cfg['frames']} frames x "
                                   f"{cfg['points']} pts (N={na} pts/set), {cfg['step_deg']} deg x +-{cfg['range_deg']} deg "
                                   f"bruteforce grid", "mode": args.mode, "pose_evals_per_step": evals // max(args.steps, 1),
                       "chain_steps_researched_on_chain_state": unresolved,
                       "value_includes": ("host staging (raw pullbacks host -> HBM over PCIe, search sets built on the device), "
                                          "search, exchange, chain walk, between alignments: the whole step, inputs on the HOST "
                                          "when a step starts") if (mode == 1 and ext is None) else
                                         ("search only, point sets staged in HBM before the timed region" if ext is not None else
                                          "the whole step (faithful chain: sets built and uploaded per chain step)"),
                       "untimed_steps_before_the_timed_region": args.warmup + main_leg["ramp"],
                       "staged_cases_in_timed_region": main_leg["staged"], "stage_ms_per_case": main_leg["stage_ms"],
                       "all_timed_steps_identical": steps_identical,
                       "parallelism": (f"(frame pair x candidate) grid in {grid[0]} x {grid[1]} tiles, one per GPU" if world > 1
                                       else "single GPU"),
                       "shard_grid": {"pair_blocks": grid[0], "cand_slices": grid[1]} if world > 1 else None,
                       "exchange_mode": os.environ.get("MM_EXCHANGE", "device") if world > 1 else None,
                       "exchange": ({"rccl": "ncclAllReduce(MIN) x 2 per level on device records, issued by the library on its own "
                                             "RCCL communicator (mm_within_plan_search_sharded)",
                                     "device": "2 all-reduces(MIN) per level on device records through torch.distributed",
                                     "gather": "host all_gather + mm_merge_shards"}[os.environ.get("MM_EXCHANGE", "device")]
                                    + (f"; NOTE: {exchange_note}" if exchange_note else "")
                                    + "; checked at start-up: rccl / device / gather exchanges give the same alignment") if world > 1 else None,
                       "step_pipeline": ((f"3 pieces per step over consecutive (independent) cases, one host thread and one engine "
                                          f"each: search of step k+1 || chain walk + between alignment of step k || staging of step "
                                          f"k+3 (engine k % {LOOK})" if STAGER else
                                          "3 pieces per step over consecutive (independent) cases: search of step k+1 || chain "
                                          "walk + between alignment of step k, then staging of step k+2 (second host thread; "
                                          "engine k % 2)") +
                                         "; K stagings, K searches, K finishes inside the timed region, the pipeline primed before it"
                                         + ("; step k+1's launch is queued on the device behind step k's long kernel "
                                            "(mm_engine_wait_search), so the device does not idle across the hand-over" if world == 1 else "")
                                         ) if pipelined else "sequential"},
            "roofline": {
                # the scale that binds the kernel: vector issue slots (one per vector instruction, two per MFMA) per second
                # against 1024 SIMDs x 2.4 GHz / 4 clk; slots = committed instruction count, time = measured live
                "bound": "valu-issue", "achieved": (iss or {}).get("achieved"), "peak": VALU_ISSUE_PEAK_GSLOTS,
                "unit": "G issue-slots/s", "frac": (iss or {}).get("frac"),
                "issue": iss,
                # SURVEY 8(d)'s ALGORITHMIC count (6 FLOP for each of the reference's 2 Na Nb pair-distances) against the fp32
                # (fp64 for --precision f64) VECTOR peak: not a fraction of a pipe for the matrix-pipe screen -- its distance
                # arithmetic does not run there and every distance is computed once, not twice -- kept as the survey defines it
                "algorithmic_tflops_vs_fp32_vector": {"achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s",
                                                      "ratio": achieved_tflops / peak},
                "executed_op_frac": (achieved_tflops / peak) * exec_per_12 / 12.0 if exec_per_12 else None,
                "traffic": (pmc or {}).get("traffic"),
                # what saturates (committed --pmc passes of the same launch): a SIMD issues one VALU wave-instruction
                # per 4 clocks at best -> valu_clk_per_instr 4.0 = back-to-back issue; the nominal peak assumes 2.4 GHz
                "valu_clk_per_instr": (pmc or {}).get("valu_clk_per_instr") if args.precision != "matrix" else None,
                # matrix-pipe screen: what saturates is still vector ISSUE -- 16 minima per 1024-distance tile at best
                "valu_instr_per_tile": (pmc or {}).get("valu_instr_per_tile"),
                "vector_issue_busy": (pmc or {}).get("vector_issue_busy"),
                "matrix_pipe_busy": (pmc or {}).get("matrix_pipe_busy"),
                "issue_slot_ns": (dominant_launch(launch_ms, launch_pe, peak)["avg_ms"] * 1e6 / pmc["issue_slots_per_simd"]
                                  if pmc and pmc.get("issue_slots_per_simd") and len(launch_ms) else None),
                "issue_slot_ns_min3_stream": (pmc or {}).get("issue_slot_ns_min3_stream"),
                # the MFMAs this kernel executes (289 tiles of 32 x 32 x 16 per candidate, K and the edges padded), live
                "matrix_pipe": ({"achieved": prof["candidates"] * 289 * 32768.0 / kern_s * 1e-12, "peak": 2500.0, "unit": "TFLOP/s (f16 dense)",
                                 "frac": prof["candidates"] * 289 * 32768.0 / kern_s * 1e-12 / 2500.0}
                                if args.precision == "matrix" and kern_s > 0 else None),
                "achieved_clock_ghz": (pmc or {}).get("achieved_clock_ghz"),
                # frac with the peak at the clock the --pmc pass ran at instead of 2.4 GHz (that pass's own duration)
                "frac_at_achieved_clock": (iss["frac"] * 2.4 / pmc["achieved_clock_ghz"]
                                           if iss and pmc.get("achieved_clock_ghz") else None),
                "pmc_source": (pmc or {}).get("source"),
                "kernel": {"matrix": "mm::k_screen_mx", "f32": "mm::k_search<float,33,16,false,false>", "fast": "mm::k_screen_fast<33, false>",
                           "bounded": "mm::k_screen_lb<5, false>",
                           "f64": "mm::k_search<double,11,16,true,true,3>"}[args.precision], "launches": prof["launches"],
                "avg_launch_ms": prof["ms"] / max(prof["launches"], 1),
                "dominant_launch": dominant_launch(launch_ms, launch_pe, peak),
                "note": ("bound = valu-issue: a point-set min/max metric is bounded by vector ISSUE (SURVEY 8(d): 'VALU instr per pair-eval vs "
                         "16 384 lanes x clock'), not by HBM and not by the matrix pipe.  achieved = issue slots of the big launch / its "
                         "duration measured in this run (hipEvents around every launch on the kernel's stream); the slots are an "
                         "instruction count taken from the committed rocprofv3 --pmc pass named in pmc_source (SQ_INSTS_VALU + "
                         "SQ_INSTS_MFMA: a vector instruction holds a SIMD's issue port for 4 clocks, an MFMA for 8); peak = 1024 SIMDs "
                         "x 2.4 GHz / 4 clk = 614.4 G slots/s.  frac therefore folds in both the slots the kernel spends above its "
                         "floor and the clock the box actually runs (2.0 - 2.2 GHz under this load, achieved_clock_ghz).  " +
                         ("MATRIX-PIPE SCREEN: d^2 = |a|^2 + |b|^2 - 2 a.b as one v_mfma_f32_32x32x16_f16 per 32 x 32 tile (f16 hi + lo "
                          "pieces, fp32 accumulate), the vector pipe keeps the minima (16 v_min3_i32 per tile at best): "
                          "valu_instr_per_tile = vector instructions other than the MFMA per tile against that floor; issue_slot_ns = "
                          "the big launch's time per issue slot per SIMD against issue_slot_ns_min3_stream, what a stream of nothing but "
                          "v_min3_i32 on every SIMD of the chip sustains (tools/ubench_mfma16c.hip); vector_issue_busy = (4 clk x vector "
                          "instructions + 8 clk x MFMAs) / GRBM_GUI_ACTIVE cycles per SIMD (the same fraction at the clock of the --pmc "
                          "pass); matrix_pipe_busy = 32 clk x MFMAs / cycles; matrix_pipe = executed f16 MFMA FLOP against the 2.5 PF "
                          "dense peak.  " if args.precision == "matrix" else "") +
                         "algorithmic_tflops_vs_fp32_vector = SURVEY 8(d)'s count, pose-evals x 2 Na Nb pair-distances x 6 FLOP / kernel "
                         "time, against the fp32 vector peak (a ratio, not a fraction of a pipe: above 1 for the matrix-pipe screen); "
                         "executed_op_frac = the same launches priced by the lane-operations the packed-FMA kernels execute (7 per 12 "
                         "algorithmic FLOP); traffic = HBM bytes of the big launch (FETCH_SIZE + WRITE_SIZE, same --pmc passes); "
                         "valu_clk_per_instr = (GRBM_GUI_ACTIVE / 8 XCDs) / (SQ_INSTS_VALU / 1024 SIMDs): 4.0 is back-to-back issue; "
                         "achieved_clock_ghz = those cycles / the launch's duration in that pass"),
                "hbm": {"bound": "hbm", "achieved": algo_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo_gbs / HBM_PEAK_GBS,
                        "note": "algorithmic no-reuse bytes ((Na+Nb)*8+8 per pose-eval) / kernel time"},
            },
        }
        if bound is not None:
            out["config"]["bounded_screen"] = bound
            out["unit"] = "candidates resolved/s"
            out["metric"] += " -- BOUNDED SEARCH: candidates ruled out by a lower bound or evaluated (same winners), not pose-evals"
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            try:
                avail = len(os.sched_getaffinity(0))
            except Exception:
                avail = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(cfg, base, args.cpu_threads or min(avail, 16))
        if args.check:
            # full-size parity: the whole 4-phase alignment of this workload through the CPU oracle
            # (sequential chain, all candidates in f64), compared bit for bit with the last step
            from oracle import oracle as orc
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import to_oracle  # type: ignore
            threads = args.cpu_threads or 16
            t0c = time.perf_counter()
            og = [to_oracle(orc, g) for g in base]
            ologs = [orc.align_within_chain(o, cfg["step_deg"], cfg["range_deg"], True, cfg["sample_size"],
                                            n_threads=threads) for o in og]
            orot = [orc.align_between(og[i], og[j], cfg["range_deg"], cfg["step_deg"], cfg["sample_size"], n_threads=threads)
                    for i, j in ((0, 1), (2, 3), (0, 2), (1, 3))]
            last = main_leg["last_case"]
            out["check"] = {
                "within_logs_identical": bool(list(res[0]) == ologs),
                "between_rotations_identical": bool(list(res[1]) == orot),
                "all_coordinates_identical": bool(all(np.array_equal(g.lumen, o.lumen) and np.array_equal(g.cath, o.cath)
                                                      and np.array_equal(g.centroids, o.centroids)
                                                      for g, o in zip(last, og))),
                "oracle_seconds": time.perf_counter() - t0c, "oracle_threads": threads,
            }
        emit(out)
    if world > 1:
        # every rank must have produced the same alignment (reduced winners -> identical host walk)
        import hashlib
        digest = hashlib.sha256(repr((res[0], None if res[1] is None else res[1].tolist())).encode()).digest()[:8]
        t = torch.frombuffer(bytearray(digest), dtype=torch.uint8).to(torch.int64)
        t = t.cuda() if backend == "nccl" else t
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit("ranks disagree on the alignment result")
        D.close_native_comms()
        dist.destroy_process_group()
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
