"""Import shim: makes the package directory `rrtmgp.jl_amd/` (whose name is not a
valid Python identifier) importable as `rrtmgp_jl_amd`."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rrtmgp.jl_amd")
_spec = _u.spec_from_file_location("rrtmgp_jl_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["rrtmgp_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
