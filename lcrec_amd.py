"""Import shim: the package directory is named ``lc-rec_amd`` (not a valid Python
identifier), so ``import lcrec_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lc-rec_amd")
_spec = importlib.util.spec_from_file_location(
    "lcrec_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lcrec_amd"] = _mod
_spec.loader.exec_module(_mod)
