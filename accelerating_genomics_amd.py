"""Import shim: the package directory is `accelerating-genomics_amd/` (not a valid
Python identifier), so `import accelerating_genomics_amd` loads it from there."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "accelerating-genomics_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_m = _u.module_from_spec(_spec)
_sys.modules[__name__] = _m
_spec.loader.exec_module(_m)
