"""bench.py end to end on the GPU box with a small workload: the driver's command form, the JSON contract
(roofline + cpu_baseline objects, the extra legs agreeing with the headline result), and `--gpus 2` started
plainly (bench.py launches its own ranks; `gloo` because two ranks share the one GPU here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_on_a_small_workload():
    d = run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "config2", "--cpu-threads", "4"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - d["config"]["pose_evals_per_step"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"] + 1
    assert d["config"]["pose_evals_per_step"] > 4 * 127 * 361         # within stage + the between stage's coarse->fine evaluations
    assert d["config"]["staged_cases_in_timed_region"] == 3          # K stagings inside the timed region
    r = d["roofline"]
    # the roofline is a fraction of the pipe that binds the kernel: vector issue slots per second against 1024 SIMDs x 2.4 GHz / 4
    assert r["bound"] == "valu-issue" and r["unit"] == "G issue-slots/s" and abs(r["peak"] - 614.4) < 1e-9
    assert 0 < r["frac"] <= 1.0 and r["kernel"] == "mm::k_screen_mx"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["issue"]["issue_slots_per_launch"] / (r["issue"]["launch_ms"] * 1e-3) * 1e-9) < 1e-6 * r["achieved"]
    assert r["algorithmic_tflops_vs_fp32_vector"]["ratio"] > 0       # SURVEY 8(d)'s count: a ratio (> 1 possible), not `frac`
    fs = d["fast_screen"]                                           # the packed-FMA screen on the same steps
    assert fs["identical_to_headline_result"] is True and 0 < fs["dominant_launch"]["frac"] <= 1.0
    assert 0 < fs["issue"]["frac"] <= 1.0
    assert r["dominant_launch"]["launches"] == 3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["bounded_search"]["identical_to_bruteforce_result"] is True
    assert d["f64_exact"]["identical_to_headline_result"] is True and d["f64_exact"]["dtype"] == "f64"
    assert d["sequential"]["ms_per_step"] > 0


@pytest.mark.parametrize("exchange", ["device", "gather"])
def test_bench_starts_its_own_ranks(exchange):
    d = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "tiny", "--no-extra-legs", "--no-cpu-baseline"],
                  {"MM_BENCH_BACKEND": "gloo", "MM_EXCHANGE": exchange, "OMP_NUM_THREADS": "4"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and "tiles" in d["config"]["parallelism"]
    assert d["config"]["shard_grid"] == {"pair_blocks": 1, "cand_slices": 2}      # 44 frame pairs: the candidate axis
    assert d["config"]["exchange_mode"] == exchange
    assert "checked at start-up" in d["config"]["exchange"]
