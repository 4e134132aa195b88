"""Times the fused f32x attention at the step's shapes (HIP events over a captured chain of launches):  DWG_ATTN_V2=0|1 python tools/bench_attn.py
Prints  B H Nq Nk d  us  TF/s (algorithmic: 4 B H Nq Nk d flops)  and the rel-L2 error against float64 of the first (small) case."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import _lib, xfmt
L = _lib.lib()
torch.cuda.set_stream(torch.cuda.Stream())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for (Bn, Hh, Nq, Nk, d) in [(2, 8, 4096, 4096, 40), (2, 8, 4096, 77, 40), (2, 8, 1024, 1024, 80), (2, 8, 1024, 77, 80), (2, 8, 256, 256, 160)]:
    g = torch.Generator().manual_seed(1)
    q = torch.randn(Bn, Nq, Hh * d, generator=g); k = torch.randn(Bn, Nk, Hh * d, generator=g); v = torch.randn(Bn, Nk, Hh * d, generator=g)
    qx, kx, vx = xfmt.pack(q).cuda(), xfmt.pack(k).cuda(), xfmt.pack(v).cuda()
    o = torch.empty(Bn, Nq, Hh * d, device="cuda", dtype=xfmt.DTYPE)
    need = int(L.dwg_attention_split_workspace_bytes(3, Bn, Hh, Nq, Nk, d))
    ws = torch.empty(max(need, 4) // 4, device="cuda")

    def run():
        rc = L.dwg_attention_forward_ws(3, Bn, Hh, Nq, Nk, d, pp(qx), Hh * d, Nq * Hh * d, pp(kx), Hh * d, Nk * Hh * d, pp(vx), Hh * d, Nk * Hh * d,
                                        pp(o), Hh * d, Nq * Hh * d, float(d) ** -0.5, pp(ws) if need else None, need, st())
        assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    REP = 20
    e0.record()
    for _ in range(REP):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REP * 1e3
    qh = q[:1, :256].double().view(1, 256, Hh, d).permute(0, 2, 1, 3).cuda(); kh = k[:1].double().view(1, Nk, Hh, d).permute(0, 2, 1, 3).cuda()
    vh = v[:1].double().view(1, Nk, Hh, d).permute(0, 2, 1, 3).cuda()
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(1, 256, Hh * d).cpu()
    got = xfmt.unpack(o.cpu())[:1, :256].double()
    err = float((got - ref).norm() / ref.norm())
    print("B=%d H=%d Nq=%-5d Nk=%-5d d=%-3d splits_ws=%-9d %8.1f us  %6.1f TF/s   rel-L2 %.2e" % (Bn, Hh, Nq, Nk, d, need, us, 4.0 * Bn * Hh * Nq * Nk * d / us / 1e6, err), flush=True)
