import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sds_step, rasterizer
step = sds_step.SDSStep(n_gaussians=100000, res=512, device=torch.device("cuda"))
rasterizer.ASYNC[0] = False
names = ["pos", "scales", "quats", "table", "mlps", "mesh"]
for i in range(14):
    step.run()
    torch.cuda.synchronize()
    g = step.opt.grad
    parts = []
    for n, grp in zip(names, step.opt.groups):
        gg = g[grp["start"]:grp["end"]]
        parts.append("%s nan=%d max=%.3g" % (n, int((~torch.isfinite(gg)).sum()), float(gg[torch.isfinite(gg)].abs().max()) if torch.isfinite(gg).any() else -1))
    print(i, "K=", rasterizer.LAST_NUM_PAIRS[0], " | ".join(parts), flush=True)
