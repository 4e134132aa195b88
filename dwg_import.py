"""Makes the on-disk package directory ``dreamwaltz-g_amd/`` importable as ``dreamwaltz_g_amd``.

The directory name is fixed by the project layout and contains a hyphen, which is not a valid
Python identifier; this shim registers it under the underscore spelling.  Usage::

    import dwg_import            # noqa: F401  (repo root must be on sys.path)
    import dreamwaltz_g_amd as dwg
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "dreamwaltz-g_amd")
_NAME = "dreamwaltz_g_amd"


def _register():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


package = _register()
