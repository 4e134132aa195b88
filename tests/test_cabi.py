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


def test_attention_key_split_heuristic_host_only():
    """dwg_attention_split_workspace_bytes (include/dwg_nn.h): which launches of the f32x attention split their keys over workgroups.  The SD-1.5
    sites at CFG batch 2 (8 heads): the 64x64 level's 4096 queries fill the chip (no workspace); the 32x32 level (1024 queries, 128 query
    blocks) and the 16x16 level (256 queries, 32 blocks) split; the 8x8 level has two key tiles and cross-attention three (77 keys): nothing
    to split; eight views batched (B = 16) fill the chip at 32x32 too.  Bytes = ranges x B x H x Nq x (padded head + 4) floats."""
    L = _lib.lib()
    F32X, BF16 = 3, 1
    f = L.dwg_attention_split_workspace_bytes
    assert f(F32X, 2, 8, 4096, 4096, 40) == 0
    assert f(F32X, 2, 8, 1024, 1024, 80) == 4 * 16 * 1024 * (96 + 4) * 4
    assert f(F32X, 2, 8, 256, 256, 160) == 4 * 16 * 256 * (160 + 4) * 4
    assert f(F32X, 2, 8, 64, 64, 160) == 0
    assert f(F32X, 2, 8, 1024, 77, 80) == 0
    assert f(F32X, 16, 8, 1024, 1024, 80) == 0
    assert f(BF16, 2, 8, 1024, 1024, 80) == 0          # only the split-precision unit has the merge launch
    assert f(F32X, 0, 8, 1024, 1024, 80) == 0 and f(F32X, 2, 8, 1024, 1024, 200) == 0


def test_round6_entry_points_validate_their_arguments_host_only():
    """Argument errors are reported before any device call (DWG_E_ARG), empty calls succeed: the VAE boundary converters and the split-K
    workspace header of dwg_gemm_desc::workspace_counters."""
    from dreamwaltz_g_amd import gemm
    L = _lib.lib()
    fake = ctypes.c_void_p(4096)                    # 16-byte aligned, never dereferenced on the host
    odd = ctypes.c_void_p(4100)
    assert L.dwg_vae_image_pack(0, 512, 512, fake, fake, None) == 0 and L.dwg_vae_image_pack(1, 0, 512, fake, fake, None) == 0
    assert L.dwg_vae_image_pack(1, 8, 8, None, fake, None) != 0 and L.dwg_vae_image_pack(1, 8, 8, fake, odd, None) != 0
    assert L.dwg_vae_image_pack(-1, 8, 8, fake, fake, None) != 0
    assert L.dwg_vae_grad_prescale_pack(0, 64, fake, 64.0, fake, fake, None) != 0          # one workgroup scans the whole gradient: B >= 1
    assert L.dwg_vae_grad_prescale_pack(1, 64, fake, 64.0, fake, None, None) != 0 and L.dwg_vae_grad_prescale_pack(1, 64, fake, 64.0, odd, fake, None) != 0
    assert L.dwg_vae_dx_unpack(0, 8, 8, fake, fake, fake, None) == 0
    assert L.dwg_vae_dx_unpack(1, 8, 8, fake, None, fake, None) != 0 and L.dwg_vae_dx_unpack(1, 8, 8, odd, fake, fake, None) != 0
    # the workspace query counts the 16 KB counter header on top of the slabs; a workspace that cannot hold the header is an argument error
    d = gemm.GemmDesc()
    d.A, d.B, d.C = 4096, 8192, 12288
    d.M, d.N, d.K = 128, 1280, 11520
    d.a_row_stride, d.a_k_stride, d.b_row_stride, d.b_k_stride, d.ldc = 11520, 1, 11520, 1, 1280
    d.batch1 = d.batch2 = 1
    d.dtype, d.out_dtype, d.alpha, d.splitk = gemm.F32X, gemm.F32X, 1.0, 0
    need = L.dwg_gemm_workspace_bytes(ctypes.byref(d))
    assert need > 16384 and (need - 16384) % (128 * 1280 * 4) == 0, need
    d.workspace, d.workspace_bytes, d.workspace_counters = 4096, 16384, 1
    assert L.dwg_gemm(ctypes.byref(d), None) != 0
    d.workspace, d.workspace_bytes = 4100, need
    assert L.dwg_gemm(ctypes.byref(d), None) != 0


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


def _c_struct_layout(header, struct, fields, tmp_path):
    """sizeof and offsetof of `struct` as the C compiler lays it out from include/<header> (the headers are plain C)."""
    import subprocess
    src = tmp_path / "layout.c"
    exe = tmp_path / "layout"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % header, 'int main(void) {',
             '  printf("%%zu\\n", sizeof(%s));' % struct]
    lines += ['  printf("%%zu\\n", offsetof(%s, %s));' % (struct, f) for f in fields]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return int(out[0]), [int(v) for v in out[1:]]


def test_ctypes_structures_have_the_layout_of_the_c_headers(tmp_path):
    """The structs that cross the C-ABI by pointer: the ctypes mirrors must agree with include/*.h field for field."""
    from dreamwaltz_g_amd import gemm
    for header, struct, mirror in (("dwg_raster.h", "dwg_raster_settings", _lib.RasterSettingsC), ("dwg_raster.h", "dwg_raster_frames", _lib.RasterFramesC),
                                   ("dwg_gemm.h", "dwg_gemm_desc", gemm.GemmDesc), ("dwg_gaussian.h", "dwg_segment", _lib.SegmentC),
                                   ("dwg_elementwise.h", "dwg_adam_group", _lib.AdamGroupC)):
        names = [f[0] for f in mirror._fields_]
        size, offsets = _c_struct_layout(header, struct, names, tmp_path)        # a field the header lacks fails to compile
        assert size == ctypes.sizeof(mirror), (struct, size, ctypes.sizeof(mirror))
        assert offsets == [getattr(mirror, n).offset for n in names], struct
        # and the header has no field the mirror lacks: the last mirrored field ends where the struct (padding aside) ends
        last = mirror._fields_[-1]
        assert offsets[-1] + ctypes.sizeof(last[1]) + 8 > size, struct


def test_every_header_compiles_standalone_as_c_and_cxx(tmp_path):
    """include/*.h is the binding surface for a C, C++ or FFI caller: each header must stand on its own in both languages."""
    import subprocess
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        for lang, cc, std in (("c", "gcc", "-std=c99"), ("c++", "g++", "-std=c++11")):
            src = tmp_path / ("t." + ("c" if lang == "c" else "cpp"))
            src.write_text('#include "%s"\nint main(void) { return 0; }\n' % os.path.basename(h))
            r = subprocess.run([cc, std, "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                               capture_output=True, text=True)
            assert r.returncode == 0, (h, lang, r.stderr[:2000])
