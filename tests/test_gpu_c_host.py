"""The drop-in boundary from a host that is not Python: tests/c_host/host.c, plain C against include/mm_hausdorff.h,
linked with libmm_hausdorff.so (and with the oracle library as its checker), compiled here with gcc and run as its own
process on the GPU.  No torch, no ctypes: this is what a Rust or C host of the reference links."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_runs_the_path_and_matches_the_oracle(oracle, tmp_path):
    import __graft_entry__ as ge
    ge.build()
    lib_dir = os.path.join(ROOT, "multimoda-rs_amd", "lib")
    orc_dir = os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "c_host")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", orc_dir,
                           os.path.join(ROOT, "tests", "c_host", "host.c"), "-o", exe,
                           "-L", lib_dir, "-lmm_hausdorff", "-L", orc_dir, "-lmm_oracle", "-lm",
                           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{orc_dir}"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C_HOST_OK" in r.stdout and "identical to the oracle at all four precisions" in r.stdout
    assert "logs and coordinates identical to the oracle" in r.stdout
    assert "sharded entry points: world = 1 RCCL communicator" in r.stdout
