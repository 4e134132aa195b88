import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sds_step, rasterizer
step = sds_step.SDSStep(n_gaussians=100000, res=512, device=torch.device("cuda"))
rasterizer.ASYNC[0] = False
which = sys.argv[1] if len(sys.argv) > 1 else "all"
gd = step.guidance
if which in ("all", "den"): gd.denoiser.plan.capture()
if which in ("all", "vfwd"): gd.vae.fwd.capture()
if which in ("all", "vbwd"): gd.vae.bwd.capture()
print("graphs:", which)
fin = lambda t: int((~torch.isfinite(t.float())).sum())
for i in range(10):
    step.run()
    torch.cuda.synchronize()
    print(i, "K", rasterizer.LAST_NUM_PAIRS[0], "eps", fin(gd.denoiser.eps), "moments", fin(gd.vae.moments), "dx", fin(gd.vae.dx),
          "grad", fin(step.opt.grad), flush=True)
    if fin(step.opt.grad):
        for nm, pl in (("den", gd.denoiser.plan), ("vfwd", gd.vae.fwd), ("vbwd", gd.vae.bwd)):
            bufs = [b for b in pl.keep if torch.is_tensor(b) and b.is_floating_point()]
            bad = [(k, tuple(b.shape), str(b.dtype).replace("torch.", ""), fin(b)) for k, b in enumerate(bufs) if fin(b)]
            print(nm, "bad", len(bad), bad[:6])
        break
