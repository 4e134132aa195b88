import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sds_step, rasterizer
step = sds_step.SDSStep(n_gaussians=100000, res=512, device=torch.device("cuda"))
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
rasterizer.ASYNC[0] = ("async" in mode)
if "graph" in mode:
    step.capture_graphs()
print("mode", mode)
names = ["pos", "scales", "quats", "table", "mlps", "mesh"]
for i in range(8):
    step.run()
    torch.cuda.synchronize()
    g = step.opt.grad
    parts = []
    for n, grp in zip(names, step.opt.groups):
        gg = g[grp["start"]:grp["end"]]
        parts.append("%s nan=%d max=%.3g" % (n, int((~torch.isfinite(gg)).sum()), float(gg[torch.isfinite(gg)].abs().max()) if torch.isfinite(gg).any() else -1))
    print(i, "K=", rasterizer.LAST_NUM_PAIRS[0], " | ".join(parts), flush=True)
