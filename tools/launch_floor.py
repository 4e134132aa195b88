"""Cost of a kernel boundary inside a captured hipGraph on this GPU: a chain of N dependent near-empty launches (dwg_add_dt on 8 elements),
and the same chain with each launch streaming 5 MB (the size of a 64x64-latent activation), replayed as one graph."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15, _lib

torch.cuda.set_stream(torch.cuda.Stream())
dev = torch.device("cuda")
L = _lib.lib()
for n in (8, 2621440, 8 * 2621440):
    plan = sd15.Plan(dev, "bf16")
    a = plan.buf(n); b = plan.buf(n); c = plan.buf(n)
    pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    N = 500
    for i in range(N):
        src, dst = (a, c) if i % 2 == 0 else (c, a)
        plan.add_call(L.dwg_add_dt, plan.dt, n, pp(src), pp(b), pp(dst))
    for _ in range(2):
        plan.run_eager()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run_eager(); e1.record(); torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / N * 1e3
    plan.capture()
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        plan.run()
    e1.record(); torch.cuda.synchronize()
    graph = e0.elapsed_time(e1) / (5 * N) * 1e3
    print("chain of %d dependent launches, %9d bf16 elements each (%.1f MB moved): eager %.2f us / launch, graph %.2f us / launch" % (
        N, n, n * 6 / 1e6, eager, graph))
