"""Import shim: the package directory is named after the reference (`implicitbvh.jl_amd/`), which is
not a valid Python identifier, so this module loads it under the name `implicitbvh_amd`.

    import implicitbvh_amd as ibvh
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "implicitbvh.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "implicitbvh_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["implicitbvh_amd"] = _mod
_spec.loader.exec_module(_mod)
