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


def test_no_kernel_spills_beyond_the_known_ones():
    """The compiler's per-kernel resource report of the last build (csrc/_obj/*.resources.json, written by build.py): no kernel
    uses scratch memory except the three listed with their bounds.  (Round 2 lost 7 ms per step to an epilogue change that
    silently demoted the GEMM accumulators to scratch; this is the tripwire.)"""
    from dreamwaltz_g_amd import build
    build.build(verbose=False)
    res = build.resources()
    assert sum(len(v) for v in res.values()) >= 100, "resource report missing: run python dreamwaltz-g_amd/build.py --force"
    allowed = {"k_conv3x3_patchILi128ELb0": 64, "k_conv3x3_patchILi128ELb1": 64, "k_meshbind_bwd": 128}
    bad = []
    for src, kernels in res.items():
        for name, r in kernels.items():
            scratch = r.get("ScratchSize", 0)
            if scratch == 0:
                continue
            lim = [v for k, v in allowed.items() if k in name]
            if not lim or scratch > lim[0]:
                bad.append((src, name, scratch))
    assert not bad, bad
