import os, sys
os.environ["DWG_PLAN_ZERO"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15
vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
sd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
plan = sd15.VAEEncoderPlan(vcfg, sd, torch.device("cuda"), image_hw=128)
img = torch.rand(1, 3, 128, 128)
plan.x[..., :3].copy_((img * 2 - 1).permute(0, 2, 3, 1).cuda())
print("fwd:", plan.fwd.run_debug(), "nops", len(plan.fwd.ops))
plan.dmoments.copy_(torch.randn(1, 16, 16, 8).cuda())
print("bwd:", plan.bwd.run_debug(), "nops", len(plan.bwd.ops))
