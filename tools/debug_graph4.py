import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15
ucfg = sd15.UNetConfig()
usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=0)
csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=1)
dev = torch.device("cuda")
plan = sd15.DenoiserPlan(ucfg, usd, csd, dev, batch=2, latent_hw=64)
del usd, csd
P = plan.plan
P.capture()
g = torch.Generator().manual_seed(5)
fin = lambda t: int((~torch.isfinite(t.float())).sum())
rel = lambda a, b: float((a - b).norm() / b.norm())
torch.cuda.set_stream(torch.cuda.Stream())
for it in range(8):
    lat = torch.randn(1, 4, 64, 64, generator=g).repeat(2, 1, 1, 1).cuda()
    text = torch.randn(2, 77, 768, generator=g).cuda()
    cond = torch.rand(1, 3, 512, 512, generator=g).cuda()
    t = torch.randint(20, 981, (1,), generator=g).cuda()
    plan.set_inputs(lat, t, text, cond)
    P.use_graph = True
    eg = plan.run().clone()
    if it % 3 == 0:
        torch.cuda.synchronize()
    P.use_graph = False
    ee = plan.run().clone()
    torch.cuda.synchronize()
    print(it, "t", int(t), "graph nonfinite", fin(eg), "eager nonfinite", fin(ee), "rel", rel(eg, ee) if fin(eg) == 0 and fin(ee) == 0 else None,
          "absmax", float(ee.abs().max()), flush=True)
