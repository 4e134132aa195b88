"""Per-shape throughput of the GEMM / implicit-GEMM conv primitive on the shapes the SDS step is made of."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib, xfmt

DT = (sys.argv[1] if len(sys.argv) > 1 else "bf16")       # bf16 | f16 | f32 | f32x
ODT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x": torch.int32}[DT]


def cv(t):
    return xfmt.pack(t) if DT == "f32x" else t.to(ODT)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rows = []
keep = []


def finish(d):
    d.splitk = 0
    need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
    if need:
        ws = torch.empty(need // 4, device="cuda"); keep.append(ws)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    else:
        d.splitk = 1


def conv_case(name, B, H, Cin, Cout, k=3, stride=1):
    x = cv(torch.randn(B, H, H, Cin, device="cuda")); w = cv(torch.randn(Cout, k, k, Cin, device="cuda") * 0.02)
    Ho = H // stride
    y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=ODT)
    M, N, K = B * Ho * Ho, Cout, k * k * Cin
    d = gemm.gemm_raw(x, w, y, M, N, K, (0, 1), (K, 1), Cout, conv=(Cin, H, H, Ho, Ho, k, k, stride, k // 2, k // 2, 1), run=False)
    finish(d)
    ms = timeit(lambda: gemm.run_desc(d, st))
    rows.append((name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))


def lin_case(name, M, N, K):
    x = cv(torch.randn(M, K, device="cuda")); w = cv(torch.randn(N, K, device="cuda") * 0.02)
    y = torch.empty(M, N, device="cuda", dtype=ODT)
    d = gemm.gemm_raw(x, w, y, M, N, K, (K, 1), (K, 1), N, run=False)
    finish(d)
    ms = timeit(lambda: gemm.run_desc(d, st))
    rows.append((name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))


conv_case("vae 512^2 128->128", 1, 512, 128, 128)
conv_case("vae 256^2 256->256", 1, 256, 256, 256)
conv_case("vae 128^2 512->512", 1, 128, 512, 512)
conv_case("vae 64^2 512->512", 1, 64, 512, 512)
conv_case("vae dgrad-like 512^2 128->8", 1, 512, 128, 8)
conv_case("unet 64^2 320->320", 2, 64, 320, 320)
conv_case("unet 64^2 640->320", 2, 64, 640, 320)
conv_case("unet 32^2 640->640", 2, 32, 640, 640)
conv_case("unet 32^2 1280->640", 2, 32, 1280, 640)
conv_case("unet 16^2 1280->1280", 2, 16, 1280, 1280)
conv_case("unet 16^2 2560->1280", 2, 16, 2560, 1280)
conv_case("unet 8^2 1280->1280", 2, 8, 1280, 1280)
conv_case("unet 8^2 2560->1280", 2, 8, 2560, 1280)
conv_case("cn hint 512^2 8->16", 1, 512, 8, 16)
conv_case("cn hint 512^2 16->16", 1, 512, 16, 16)
conv_case("conv1x1 64^2 320", 2, 64, 320, 320, k=1)
lin_case("ff_in 64^2", 8192, 2560, 320)
lin_case("ff_out 64^2", 8192, 320, 1280)
lin_case("qkv 64^2", 8192, 960, 320)
lin_case("ff_in 32^2", 2048, 5120, 640)
lin_case("ff_out 32^2", 2048, 640, 2560)
lin_case("ff_in 16^2", 512, 10240, 1280)
lin_case("ff_out 16^2", 512, 1280, 5120)
lin_case("kv text", 154, 640, 768)
lin_case("big 4096^3", 4096, 4096, 4096)
print("dtype", DT)
for r in rows:
    print("%-30s M=%7d N=%5d K=%6d  %8.3f ms  %8.1f TF/s" % r)
