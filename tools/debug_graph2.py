import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15
vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
sd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
plan = sd15.VAEEncoderPlan(vcfg, sd, torch.device("cuda"), image_hw=128)
g = torch.Generator().manual_seed(0)
img = torch.rand(1, 3, 128, 128, generator=g).cuda()
gm = torch.randn(1, 8, 16, 16, generator=g).cuda()
m0 = plan.encode(img).clone(); d0 = plan.backward(gm).clone()
plan.fwd.capture(); plan.bwd.capture()
img2 = torch.rand(1, 3, 128, 128, generator=g).cuda(); gm2 = torch.randn(1, 8, 16, 16, generator=g).cuda()
m1 = plan.encode(img2).clone(); d1 = plan.backward(gm2).clone()
plan.fwd.use_graph = False; plan.bwd.use_graph = False
m2 = plan.encode(img2).clone(); d2 = plan.backward(gm2).clone()
rel = lambda a, b: float((a - b).norm() / b.norm())
print("fwd graph vs eager rel", rel(m1, m2), " bwd graph vs eager rel", rel(d1, d2))
print("sanity: different inputs differ", rel(m0, m2), rel(d0, d2))
# mixed: graph fwd + eager bwd etc.
plan.fwd.use_graph = True
m3 = plan.encode(img2).clone(); d3 = plan.backward(gm2).clone()
print("graph fwd + eager bwd:", rel(m3, m2), rel(d3, d2))
plan.fwd.use_graph = False; plan.bwd.use_graph = True
m4 = plan.encode(img2).clone(); d4 = plan.backward(gm2).clone()
print("eager fwd + graph bwd:", rel(m4, m2), rel(d4, d2))
