"""N > 1 path on CPU: candidate-axis sharding + exchange + merge over torch.distributed `gloo`."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_candidate_axis_sharding_gloo(world):
    import __graft_entry__ as ge
    ge.build()
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world + (os.getpid() % 200)),
           os.path.join(ROOT, "tests", "_gloo_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "GLOO_SHARD_OK" in r.stdout


def test_native_comm_setup_is_decided_by_all_ranks():
    """A rank that cannot load RCCL sends EVERY rank to the error, before anybody waits in a collective of the setup
    (bench.py then falls back to torch's all-reduces on all ranks together)."""
    import __graft_entry__ as ge
    ge.build()
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    env.pop("MM_RCCL_LIB", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29800 + (os.getpid() % 150)),
           os.path.join(ROOT, "tests", "_gloo_comm_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0: COMM_SETUP_REFUSED RCCL cannot be loaded on at least one rank" in r.stdout, r.stdout[-2000:]
    assert "rank 1: COMM_SETUP_REFUSED RCCL cannot be loaded on at least one rank" in r.stdout, r.stdout[-2000:]


def test_broadcast_state_world3_gloo():
    """The collective of the sharded finish on the CPU: distributed.broadcast_state over three `gloo` ranks."""
    import __graft_entry__ as ge
    ge.build()
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", str(29100 + (os.getpid() % 150)),
           os.path.join(ROOT, "tests", "_gloo_bcast_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for k in range(3):
        assert f"rank {k}: BCAST_OK" in r.stdout, r.stdout[-2000:]


def test_merge_shards_single_rank_is_identity(mm):
    from multimoda_rs_amd import distributed as D
    cost = np.array([[0.5, np.inf, 0.1]])
    ok, angle, idx, out_cost = D.merge_shards(1, cost, np.array([[1, 1, 0]]), np.array([[0.3, 0.0, -0.2]]),
                                             np.array([[4, -1, 9]]), np.array([1e-12] * 3))
    assert list(ok) == [1, 1, 0] and list(idx) == [4, -1, 9]
    assert list(angle) == [0.3, 0.0, -0.2] and out_cost[0] == 0.5 and np.isinf(out_cost[1])


def test_shard_bounds_partition():
    from multimoda_rs_amd import distributed as D
    for n in (0, 1, 5, 361, 721):
        for world in (1, 2, 3, 8):
            b = [D.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
