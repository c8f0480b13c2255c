import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _scratch_cwd(tmp_path, monkeypatch):
    """The entry points write OBJ files to relative ``output/...`` directories by default, like the
    reference (write_obj=True); every test runs in its own scratch directory."""
    monkeypatch.chdir(tmp_path)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; compiled on demand with gcc)."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def mm():
    import multimoda_rs_amd
    return multimoda_rs_amd


@pytest.fixture(scope="session")
def engine(mm):
    """One engine for the whole GPU session (GPU tests run in one process)."""
    import __graft_entry__ as ge
    ge.build()
    eng = mm.Engine()
    yield eng
    eng.close()
