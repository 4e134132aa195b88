"""Timing of the batched text k/v projection GEMM (M = 154 text rows, K = 768) at several N, and of the per-block shapes it replaces."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib

st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(M, N, K, reps=30):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16() * 0.02
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = gemm.gemm_raw(x, w, y, M, N, K, (K, 1), (K, 1), N, run=False)
    d.splitk = 0
    need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(need, 16) // 4, device="cuda")
    if need:
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    else:
        d.splitk = 1
    for _ in range(3):
        gemm.run_desc(d, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gemm.run_desc(d, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("M %4d N %6d K %4d  splitk-ws %8d  %.1f us  %.1f TF/s  (weights %.1f MB -> %.0f GB/s)" % (M, N, K, need, ms * 1e3, 2.0 * M * N * K / ms / 1e9,
                                                                                                   N * K * 2 / 1e6, N * K * 2 / ms / 1e6))


for N in (640, 1280, 2560, 11520, 24960):
    run(154, N, 768)
run(128, 24960, 768)
run(256, 24960, 768)
