"""Importable alias for the package directory `aes-gcm-128-192-256-bits_amd/` (its name, fixed by the
project layout, is not a valid Python identifier).  `import aesgcm_amd` registers that directory as
the package `aesgcm_amd`, so `from aesgcm_amd import gcm_model, lib` works from the repo root."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aes-gcm-128-192-256-bits_amd")
_spec = importlib.util.spec_from_file_location("aesgcm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["aesgcm_amd"] = _mod
_spec.loader.exec_module(_mod)
