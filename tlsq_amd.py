"""Import shim: the package directory is `totalleastsquares.jl_amd/` (a dot in the name), which Python's
import system cannot address directly.  `import tlsq_amd` loads it under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "totalleastsquares.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "tlsq_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tlsq_amd"] = _mod
_spec.loader.exec_module(_mod)
