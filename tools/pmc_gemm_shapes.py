"""A few SDS GEMM / conv shapes run in isolation (for rocprofv3 --pmc SQ counter passes)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib

st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
keep = []


def finish(d):
    d.splitk = 0
    need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
    if need:
        ws = torch.empty(need // 4, device="cuda"); keep.append(ws)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    else:
        d.splitk = 1


def conv_case(B, H, Cin, Cout, k=3):
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, k, k, Cin, device="cuda").bfloat16() * 0.02
    y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    M, N, K = B * H * H, Cout, k * k * Cin
    d = gemm.gemm_raw(x, w, y, M, N, K, (0, 1), (K, 1), Cout, conv=(Cin, H, H, H, H, k, k, 1, k // 2, k // 2, 1), run=False)
    finish(d); keep.extend([x, w, y])
    for _ in range(10):
        gemm.run_desc(d, st)


conv_case(2, 8, 1280, 1280)      # k_gemm_glds<64,true> split-K, M = 128
conv_case(2, 16, 1280, 1280)     # M = 512
conv_case(2, 32, 640, 640)       # M = 2048
conv_case(2, 64, 320, 320)       # patch conv <64>
conv_case(1, 128, 512, 512)      # patch conv <128>
torch.cuda.synchronize()
