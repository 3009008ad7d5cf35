"""Import shim: exposes the package directory ``pytorch-wavelet-toolbox_amd/`` (not a valid Python
identifier) under the importable name ``ptwt_amd``.  ``import ptwt_amd as ptwt`` from the repo root."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pytorch-wavelet-toolbox_amd")
_spec = importlib.util.spec_from_file_location(
    "ptwt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ptwt_amd"] = _mod
_spec.loader.exec_module(_mod)
