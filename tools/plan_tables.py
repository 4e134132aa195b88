"""Per-kernel-label time of each static plan (eager, HIP-event timers), to see where a plan's time goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import guidance as gd, _lib

torch.cuda.set_stream(torch.cuda.Stream())
g = gd.ControlNetScoreDistillation(torch.device("cuda"), image_hw=512, seed=0)
plans = {"vae_fwd": g.vae.fwd, "vae_bwd": g.vae.bwd, "denoiser": g.denoiser.plan}
for name, p in plans.items():
    for _ in range(2):
        p.run_eager()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(5):
        p.run_eager()
    torch.cuda.synchronize()
    tab = _lib.prof_table(); _lib.prof_enable(False)
    tot = sum(v[1] for v in tab.values()) / 5
    print("== %s: %.3f ms (sum of kernels), %d launches" % (name, tot, sum(v[0] for v in tab.values()) // 5))
    for k, (c, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-26s %4d x %8.1f us = %7.3f ms" % (k, c // 5, ms / c * 1e3, ms / 5))
