"""Import shim: ``import pp_amd`` loads the package in ``3d-object-detection_amd/``
(a directory name that is not a Python identifier) and registers it as ``pp_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3d-object-detection_amd")
_spec = importlib.util.spec_from_file_location(
    "pp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pp_amd"] = _mod
_spec.loader.exec_module(_mod)
