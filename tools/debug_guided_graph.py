"""Where a graphed guided step leaves the eager trajectory: flat-parameter distances after the builder's eager steps and after each replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import guidance, sd15, sds_step

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
res, warm = 128, 2
ucfg = sd15.UNetConfig(block_out_channels=(64, 128, 128, 128), cross_dim=64, cond_channels=(16, 32, 32, 64))
vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=1)
csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=2)
vsd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
gd = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, usd, csd, vsd, image_hw=res, dtype=sys.argv[1] if len(sys.argv) > 1 else "f32x")


def make():
    return sds_step.SDSStep(n_gaussians=6000, res=res, device=dev, guidance=True, guidance_obj=gd, async_pair_count=True, iters=1000)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def groups(a, b):
    out = {}
    for name, o in a.optimizers.items():
        for ga, gb in zip(o.param_groups, b.optimizers[name].param_groups):
            pa, pb = ga["params"][0], gb["params"][0]
            out["%s.%s" % (name, ga.get("name", "?"))] = "%.1e" % rel(pa.detach(), pb.detach())
    return out


e0, e1, tw = make(), make(), make()
for _ in range(warm + 1):
    e0.run(); e1.run()
torch.cuda.synchronize()
runner = tw.graphed(warmup=warm)
torch.cuda.synchronize()
print("after the builder's %d eager steps: e1/e0 %.3e  twin/e0 %.3e" % (warm + 1, rel(e1.optimizers.buffers.flat, e0.optimizers.buffers.flat),
                                                                      rel(tw.optimizers.buffers.flat, e0.optimizers.buffers.flat)))
for k in range(4):
    e0.run(); e1.run(); runner.step()
    torch.cuda.synchronize()
    print("replay %d: e1/e0 %.3e  twin/e0 %.3e  t=%d" % (k, rel(e1.optimizers.buffers.flat, e0.optimizers.buffers.flat),
                                                      rel(tw.optimizers.buffers.flat, e0.optimizers.buffers.flat), int(runner.graph._rand[1][0])))
print("per group twin/e0:", groups(tw, e0))
print("per group e1/e0:  ", groups(e1, e0))
