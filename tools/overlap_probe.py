"""Does a memory-bound GroupNorm pass hide under a power-bound convolution?  The VAE's 512^2 x 128-channel layer: N convolutions on one stream,
N GroupNorm (statistics + apply) calls on another, alone and together.   python tools/overlap_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gemm, _lib, xfmt

L = _lib.lib()
H, C, G = int(os.environ.get("PROBE_H", "512")), int(os.environ.get("PROBE_C", "128")), 32
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
x = xfmt.pack(torch.randn(1, H, H, C, device="cuda")); w = xfmt.pack(torch.randn(C, 3, 3, C, device="cuda") * 0.03)
y = torch.empty(1, H, H, C, device="cuda", dtype=torch.int32)
K = 9 * C
d = gemm.gemm_raw(x, w, y, H * H, C, K, (0, 1), (K, 1), C, conv=(C, H, H, H, H, 3, 3, 1, 1, 1, 1), run=False)
x2 = xfmt.pack(torch.randn(1, H, H, C, device="cuda")); y2 = torch.empty_like(x2)
gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
stats = torch.empty(2 * G, device="cuda"); ws = torch.empty(int(L.dwg_groupnorm_workspace_floats(1, G)), device="cuda")
N = 20


def convs():
    st = ctypes.c_void_p(s1.cuda_stream)
    for _ in range(N):
        gemm.run_desc(d, st)


def norms():
    st = ctypes.c_void_p(s2.cuda_stream)
    for _ in range(N):
        _lib.check(L.dwg_groupnorm_forward_dt(3, 1, H * H, C, G, p(x2), p(gam), p(bet), 1e-6, 1, p(y2), p(stats), p(ws), st), "gn")


def timed(fs):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    s1.wait_event(e0); s2.wait_event(e0)
    for f in fs:
        f()
    a, b = torch.cuda.Event(), torch.cuda.Event()
    a.record(s1); b.record(s2); cur.wait_event(a); cur.wait_event(b)
    e1.record(cur); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


for f in (convs, norms):
    f()
tc, tn, tb = timed([convs]), timed([norms]), timed([convs, norms])
print("H=%d C=%d  conv alone %.1f us   GroupNorm(+SiLU) alone %.1f us   both streams %.1f us per pair   (serial %.1f; hidden %.0f %% of the norm)"
      % (H, C, tc, tn, tb, tc + tn, 100.0 * (tc + tn - tb) / tn))
