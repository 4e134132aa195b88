"""One GEMM / conv shape of the denoiser timed back to back with COLD weights (a ring of weight copies larger than the Infinity Cache), so
that environment switches of csrc/gemm.hip can be A/B-ed per shape:   DWG_...=x python tools/shape_sweep.py
Prints  name  M N K  us  TF/s  (weights GB/s)  for a fixed list of shapes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib, sd15
torch.cuda.set_stream(torch.cuda.Stream())

st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L = _lib.lib()
keep = []


def make(kind, B, H, Cin, Cout, k, M=None, copies=10):
    descs = []
    for c in range(copies):
        if kind == "conv":
            x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = (torch.randn(Cout, k, k, Cin, device="cuda") * 0.02).bfloat16()
            y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
            Mm, N, K = B * H * H, Cout, k * k * Cin
            d = gemm.gemm_raw(x, w, y, Mm, N, K, (0, 1), (K, 1), Cout, conv=(Cin, H, H, H, H, k, k, 1, k // 2, k // 2, 1), run=False)
        else:
            Mm, N, K = M, Cout, Cin
            x = torch.randn(Mm, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
            y = torch.empty(Mm, N, device="cuda", dtype=torch.bfloat16)
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


SHAPES = [("conv3x3 r8 1280", "conv", 2, 8, 1280, 1280, 3, None), ("conv3x3 r16 1280", "conv", 2, 16, 1280, 1280, 3, None),
          ("conv3x3 r32 640", "conv", 2, 32, 640, 640, 3, None), ("conv3x3 r64 320", "conv", 2, 64, 320, 320, 3, None),
          ("ff_in 512x10240x1280", "lin", 0, 0, 1280, 10240, 0, 512), ("ff_in 8192x2560x320", "lin", 0, 0, 320, 2560, 0, 8192),
          ("attn_out 8192x320x320", "lin", 0, 0, 320, 320, 0, 8192), ("attn_out 512x1280x1280", "lin", 0, 0, 1280, 1280, 0, 512),
          ("ff_out 512x1280x5120", "lin", 0, 0, 5120, 1280, 0, 512), ("conv1x1 r32 640", "conv", 2, 32, 640, 640, 1, None)]
if os.environ.get("PROBE"):
    SHAPES = [("probe M=%d N=%d K=%d" % (m, n, k), "lin", 0, 0, k, n, 0, m) for (m, n, k) in
              [(8192, 320, 64), (8192, 320, 320), (8192, 320, 1280), (1024, 320, 320), (128, 64, 64), (128, 64, 6400), (32768, 320, 320),
               (8192, 1280, 320), (8192, 64, 320)]]
only = os.environ.get("SHAPES")
for name, kind, B, H, Cin, Cout, k, M in SHAPES:
    if only and not any(t in name for t in only.split(",")):
        continue
    descs, Mm, N, K = make(kind, B, H, Cin, Cout, k, M)
    plan = sd15.Plan(torch.device("cuda"), "bf16")          # one captured graph: no host time between the launches
    R = 4
    for _ in range(R):
        for d in descs:
            plan.add_gemm(d, allow_split=False)              # make() already chose the split and the workspace
    plan.run_eager(); torch.cuda.synchronize()
    plan.capture()
    plan.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        plan.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (3 * R * len(descs)) * 1e3
    print("%-24s M=%-5d N=%-5d K=%-6d %7.1f us %6.0f TF/s  weights %5.0f GB/s" % (name, Mm, N, K, us, 2.0 * Mm * N * K / us / 1e6, N * K * 2 / us / 1e3), flush=True)
    del descs; keep.clear(); torch.cuda.empty_cache()
