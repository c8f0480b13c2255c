"""Import shim: ``import multimoda_rs_amd`` loads the package that lives in the
``multimoda-rs_amd/`` directory (a hyphen is not importable as a module name)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimoda-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "multimoda_rs_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["multimoda_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
