"""Import shim: the package directory is literally `scythe.jl_amd/` (a dot cannot appear in a module name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scythe.jl_amd")
_spec = importlib.util.spec_from_file_location("scythe_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["scythe_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
