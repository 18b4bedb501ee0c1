"""Import alias for the package directory ``graphcast-lite_amd/``.

The directory name carries a hyphen (it is the name the build contract prescribes), which the
``import`` statement cannot spell.  This one-file module replaces itself in ``sys.modules`` with
the real package loaded from that directory, so ``import graphcast_lite_amd.models`` works from
the repo root without installing anything.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "graphcast-lite_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
