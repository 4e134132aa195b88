import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15
ucfg = sd15.UNetConfig(block_out_channels=(64, 128, 128, 128), cross_dim=64, cond_channels=(16, 32, 32, 64))
usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=1)
csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=2)
plan = sd15.DenoiserPlan(ucfg, usd, csd, torch.device("cuda"), batch=2, latent_hw=16)
g = torch.Generator().manual_seed(5)
lat = torch.randn(1, 4, 16, 16, generator=g).repeat(2, 1, 1, 1).cuda()
text = torch.randn(2, 77, ucfg.cross_dim, generator=g).cuda()
cond = torch.rand(1, 3, 128, 128, generator=g).cuda()
t = torch.tensor([437]).cuda()
plan.set_inputs(lat, t, text, cond)
P = plan.plan
P.run_eager(); torch.cuda.synchronize()
bufs = [b for b in P.keep if torch.is_tensor(b)]
snap = [b.clone() for b in bufs]
inputs = {plan.latents.data_ptr(), plan.text.data_ptr(), plan.cond.data_ptr(), plan.temb_u.tin.data_ptr(), plan.temb_c.tin.data_ptr()}
P.capture()
g2 = torch.Generator().manual_seed(99)
lat2 = torch.randn(1, 4, 16, 16, generator=g2).repeat(2, 1, 1, 1).cuda()
text2 = torch.randn(2, 77, ucfg.cross_dim, generator=g2).cuda()
cond2 = torch.rand(1, 3, 128, 128, generator=g2).cuda()
plan.set_inputs(lat2, torch.tensor([100]).cuda(), text2, cond2)
torch.cuda.synchronize()
P.graph.replay(); torch.cuda.synchronize()
snapg = [b.clone() for b in bufs]
P.run_eager(); torch.cuda.synchronize()
bad = []
for k, (b, s_) in enumerate(zip(bufs, snapg)):
    if not torch.equal(b, s_):
        d = (b.float() - s_.float()).abs().max().item()
        bad.append((k, tuple(b.shape), str(b.dtype), d, P.tags[k][0] if k < len(P.tags) else -1))
print("n bufs", len(bufs), "graph-vs-eager mismatching:", len(bad))
for x in bad[:15]:
    print(x)
print("n ops", len(P.ops))
