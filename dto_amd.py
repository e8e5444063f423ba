"""Import alias for the package directory `directtrajectoryoptimization.jl_amd/`.

The directory name is fixed by the project layout and is not a valid Python identifier
(it contains a dot), so it is loaded here under the module name `dto_amd`.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "directtrajectoryoptimization.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "dto_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dto_amd"] = _mod
_spec.loader.exec_module(_mod)
