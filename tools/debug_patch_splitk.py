"""Debug aid: LDS-patch 3x3 convolution with split-K over channel slabs -- per-slab coverage and values vs fp64 partial sums."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib

C, H = int(sys.argv[1]) if len(sys.argv) > 1 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 32
os.environ["DWG_CONV_PATCH_MINM"] = "512"
g = torch.Generator().manual_seed(C + H)
Bn = 2
x = torch.randn(Bn, C, H, H, generator=g).bfloat16(); w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).bfloat16()
b = torch.randn(C, generator=g)
xc, wc = x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda()
y = torch.empty(Bn, H, H, C, device="cuda", dtype=torch.bfloat16)
M, K = Bn * H * H, 9 * C
d = gemm.gemm_raw(xc, wc, y, M, C, K, (0, 1), (K, 1), C, bias=b.cuda(), conv=(C, H, H, H, H, 3, 3, 1, 1, 1, 1), run=False)
d.splitk = 0
need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
nslab = need // (M * C * 4)
print("workspace slabs", nslab)
ws = torch.full((max(need, 16) // 4,), float("nan"), device="cuda")
d.workspace, d.workspace_bytes = ws.data_ptr(), need
_lib.prof_enable(True)
gemm.run_desc(d, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("kernels", list(_lib.prof_symbols().keys())); _lib.prof_enable(False)
slabs = ws[: nslab * M * C].view(nslab, Bn, H, H, C).cpu()
for s in range(nslab):
    nan = torch.isnan(slabs[s])
    print("slab", s, "untouched fraction", float(nan.float().mean()),
          "untouched rows(y) of image 0:", sorted(set(torch.nonzero(nan[0].any(-1).any(-1)).reshape(-1).tolist()))[:40],
          "cols:", sorted(set(torch.nonzero(nan[0].any(0).any(0)).reshape(-1).tolist()))[:20])
ncc = C // 64
for sk in range(2, nslab + 1):
    per = (ncc + sk - 1) // sk
    tot = 0
    print("-- hypothesis: kernel used", sk, "slices")
    for s in range(sk):
        c0, c1 = s * per * 64, min(C, (s + 1) * per * 64)
        if c0 >= c1:
            continue
        ref = torch.nn.functional.conv2d(x[:, c0:c1].double(), w[:, c0:c1].double(), padding=1).permute(0, 2, 3, 1)
        got = torch.nan_to_num(slabs[s].double())
        print("   slab", s, "channels", c0, c1, "rel", float((got - ref).norm() / ref.norm()))
ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
print("final rel", float((torch.nan_to_num(y.cpu().double()) - ref).norm() / ref.norm()), "nan in y:", float(torch.isnan(y).float().mean()))
