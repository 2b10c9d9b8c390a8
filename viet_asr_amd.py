"""Import shim: ``import viet_asr_amd`` -> the package living in ``viet-asr_amd/``.

The package directory carries the reference's repo name (hyphenated, as the build
contract asks), which is not a Python identifier; this file loads it under an
importable name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "viet-asr_amd")
_spec = importlib.util.spec_from_file_location(
    "viet_asr_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["viet_asr_amd"] = _mod
_spec.loader.exec_module(_mod)
