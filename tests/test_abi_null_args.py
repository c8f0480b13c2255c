"""Every function of the C ABI survives NULL pointers: a bad argument is a negative status (mm_last_error() says
why) or a harmless value, never a fault.  The calls run in a child process (tests/_abi_fuzz_worker.py) that announces
each call before making it; if the child dies, the test names the call.  On the CPU every pointer is NULL (functions
that need an engine stop at that check); on the GPU a live engine goes into every `mm_engine*` first argument, so the
validation behind it runs too.  (Found this way: mm_hausdorff_2d / _batch, the batch searches and four host helpers
dereferenced NULL arrays of non-empty sets.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def drive(mode):
    import __graft_entry__ as ge
    ge.build()
    start, crashes = 0, []
    while True:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_abi_fuzz_worker.py"), str(start), mode],
                           capture_output=True, text=True, timeout=600)
        calls = [l.split() for l in r.stdout.splitlines() if l.startswith("CALL ")]
        if "DONE" in r.stdout.splitlines():
            assert r.returncode == 0, r.stderr[-2000:]
            break
        assert calls, f"the worker did not start: rc {r.returncode}\n{r.stderr[-2000:]}"
        k, name = int(calls[-1][1]), calls[-1][2]
        crashes.append((name, r.returncode))
        start = k + 1                                   # carry on behind the call that took the child down
        assert len(crashes) < 20
    return crashes, len(calls)


def test_null_arguments_never_fault_on_the_host():
    crashes, _ = drive("null")
    assert crashes == []


@pytest.mark.gpu
def test_null_arguments_behind_a_live_engine_never_fault():
    crashes, n = drive("engine")
    assert crashes == [] and n > 50
