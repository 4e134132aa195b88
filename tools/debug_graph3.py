import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import guidance, sd15
ucfg = sd15.UNetConfig(block_out_channels=(64, 128, 128, 128), cross_dim=64, cond_channels=(16, 32, 32, 64))
vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
dev = torch.device("cuda")
gd = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, image_hw=128, seed=4)
g = torch.Generator().manual_seed(11)
img = torch.rand(1, 3, 128, 128, generator=g).cuda()
text = {"neg": torch.randn(1, 77, ucfg.cross_dim, generator=g).cuda(), "text": torch.randn(1, 77, ucfg.cross_dim, generator=g).cuda()}
cond = torch.rand(1, 3, 128, 128, generator=g).cuda()
noise = torch.randn(1, 4, 16, 16, generator=g).cuda(); vn = torch.randn(1, 4, 16, 16, generator=g).cuda()
t = torch.tensor([321], device=dev)
def once():
    ic = img.clone().requires_grad_(True)
    out = gd(ic, text, cond_inputs=cond, timestep=t, noise=noise, posterior_noise=vn)
    out["diffusion_loss"].backward()
    return out["gradients"].clone(), ic.grad.clone(), out["latents"].detach().clone(), out["targets"].clone()
rel = lambda a, b: float((a - b).norm() / b.norm())
r0 = once(); r1 = once()
print("eager vs eager:", [rel(a, b) for a, b in zip(r1, r0)])
import gc
gc.collect(); torch.cuda.empty_cache()
r2 = once()
print("eager after empty_cache:", [rel(a, b) for a, b in zip(r2, r0)])
for nm, pl in (("den", gd.denoiser.plan), ("vfwd", gd.vae.fwd), ("vbwd", gd.vae.bwd)):
    bad = [(k, tuple(b.shape), str(b.dtype)) for k, b in enumerate([b for b in pl.keep if torch.is_tensor(b) and b.is_floating_point()]) if not torch.isfinite(b.float()).all()]
    print(nm, "non-finite buffers:", len(bad), bad[:4])
