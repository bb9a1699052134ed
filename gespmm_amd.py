"""Import shim: the package directory is named ``ge-spmm_amd`` (not a valid Python
identifier), so ``import gespmm_amd`` loads it from that directory under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ge-spmm_amd")
_spec = importlib.util.spec_from_file_location(
    "gespmm_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gespmm_amd"] = _mod
_spec.loader.exec_module(_mod)
