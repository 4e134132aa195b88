"""Times a few denoiser conv / linear shapes on the f32x GEMM unit with COLD weights (a ring of copies larger than the Infinity Cache):
A/B of csrc/gemm.hip environment switches (DWG_GEMM_BIG, DWG_GEMM_DEBUG, ...) per shape.   DWG_...=x python tools/gemm_probe.py [f32x|bf16]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib, xfmt

DT = sys.argv[1] if len(sys.argv) > 1 else "f32x"
torch.cuda.set_stream(torch.cuda.Stream())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L = _lib.lib()
keep = []


ZERO = os.environ.get("PROBE_ZERO") == "1"       # all-zero operands: the same instruction stream without the data-dependent power draw


LO_BITS = int(os.environ.get("PROBE_LO_BITS", "11"))   # f32x only: significand bits kept in the lo halves (11 = the format's own; fewer: low bits zeroed -> less toggling)
HI_ONLY = os.environ.get("PROBE_HI_ONLY") == "1"         # f32x only: lo halves all zero (the data of an fp16 tensor through the three-MFMA stream)


def cv(t):
    if ZERO:
        t = torch.zeros_like(t)
    if DT != "f32x":
        return t.to(torch.bfloat16)
    x = xfmt.pack(t)
    if LO_BITS < 11 or HI_ONLY:
        h = x.view(torch.int16).view(x.shape[:-1] + (x.shape[-1] // 8, 2, 8)).clone()      # [..., group, {hi, lo}, 8]
        mask = 0 if HI_ONLY else (~((1 << (11 - LO_BITS)) - 1)) & 0xFFFF
        mask = mask - 65536 if mask >= 32768 else mask
        h[..., 1, :] &= mask
        x = h.reshape(x.shape[:-1] + (x.shape[-1] * 2,)).view(torch.int32)
    return x


ODT = torch.int32 if DT == "f32x" else torch.bfloat16


def make(kind, B, H, Cin, Cout, k, M=None, copies=6):
    descs = []
    for c in range(copies):
        if kind == "conv":
            x = cv(torch.randn(B, H, H, Cin, device="cuda")); w = cv(torch.randn(Cout, k, k, Cin, device="cuda") * 0.02)
            y = torch.empty(B, H, H, Cout, device="cuda", dtype=ODT)
            Mm, N, K = B * H * H, Cout, k * k * Cin
            d = gemm.gemm_raw(x, w, y, Mm, N, K, (0, 1), (K, 1), Cout, conv=(Cin, H, H, H, H, k, k, 1, k // 2, k // 2, 1), run=False)
        else:
            Mm, N, K = M, Cout, Cin
            x = cv(torch.randn(Mm, K, device="cuda")); w = cv(torch.randn(N, K, device="cuda") * 0.02)
            y = torch.empty(Mm, N, device="cuda", dtype=ODT)
            d = gemm.gemm_raw(x, w, y, Mm, N, K, (K, 1), (K, 1), N, run=False)
        d.splitk = 0
        need = L.dwg_gemm_workspace_bytes(ctypes.byref(d))
        if need:
            ws = torch.empty(need // 4, device="cuda"); keep.append(ws)
            d.workspace, d.workspace_bytes = ws.data_ptr(), need
        else:
            d.splitk = 1
        keep.extend([x, w, y]); descs.append(d)
    return descs, Mm, N, K


CASES = [("conv r16 1280", "conv", 2, 16, 1280, 1280, 3, None), ("conv r32 640", "conv", 2, 32, 640, 640, 3, None),
         ("conv r8 1280", "conv", 2, 8, 1280, 1280, 3, None), ("conv r64 320 (patch)", "conv", 2, 64, 320, 320, 3, None),
         ("vae r512 128", "conv", 1, 512, 128, 128, 3, None), ("vae r256 256", "conv", 1, 256, 256, 256, 3, None),
         ("vae r128 512", "conv", 1, 128, 512, 512, 3, None), ("vae r64 512", "conv", 1, 64, 512, 512, 3, None),
         ("ff_out 2048x640x2560", "lin", 0, 0, 2560, 640, 0, 2048), ("qkv 512x3840x1280", "lin", 0, 0, 1280, 3840, 0, 512),
         ("attn_out 8192x320x320", "lin", 0, 0, 320, 320, 0, 8192), ("big 4096^3", "lin", 0, 0, 4096, 4096, 0, 4096)]
only = os.environ.get("PROBE_ONLY")
for name, kind, B, H, Cin, Cout, k, M in CASES:
    if only and only not in name:
        continue
    descs, Mm, N, K = make(kind, B, H, Cin, Cout, k, M, copies=2 if ("4096" in name or "vae" in name) else 6)
    for d in descs:
        gemm.run_desc(d, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    REP = 10
    e0.record()
    for _ in range(REP):
        for d in descs:
            gemm.run_desc(d, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (REP * len(descs)) * 1e3
    print("%-26s M=%-5d N=%-5d K=%-6d %8.1f us  %6.0f TF/s" % (name, Mm, N, K, us, 2.0 * Mm * N * K / us / 1e6), flush=True)
    if os.environ.get("PROBE_PROF") == "1":         # per-kernel split of the same call (dwg_prof: events around every launch)
        _lib.prof_enable(True)
        for d in descs:
            gemm.run_desc(d, st)
        torch.cuda.synchronize()
        for sym, (n, ms, _) in sorted(_lib.prof_symbols().items()):
            print("    %-44s x%-3d %8.1f us each" % (sym, n, ms / n * 1e3), flush=True)
        _lib.prof_enable(False)
