"""Import shim: `import ssd_amd` loads the package in ./single-shot-detector_amd/ (a
directory name that is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "single-shot-detector_amd")
_spec = importlib.util.spec_from_file_location(
    "ssd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ssd_amd"] = _mod
_spec.loader.exec_module(_mod)
