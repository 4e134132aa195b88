"""CPU-side checks of the C-ABI: the shared library loads, and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import dreamwaltz_g_amd._lib as _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(dwg_[a-z0-9_]+)\s*\(", src):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    decl = _declared_symbols()
    assert decl, "no declarations found"
    for name in sorted(decl):
        assert hasattr(L, name), "libdwg_hip.so does not export %s" % name
    # and every declared symbol has a ctypes signature in the binding
    assert decl <= set(_lib.SIGNATURES), sorted(decl - set(_lib.SIGNATURES))


def test_workspace_sizes_host_only():
    L = _lib.lib()
    g, p, i = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert L.dwg_raster_workspace_sizes(1000, 512, 512, 5000, ctypes.byref(g), ctypes.byref(p), ctypes.byref(i)) == 0
    assert g.value >= 1000 * (48 + 8) + 3 * 1024 * 4
    assert p.value >= 5000 * 12
    assert i.value >= 512 * 512 * 8
    assert L.dwg_raster_workspace_sizes(-1, 512, 512, 0, None, None, None) != 0
