"""How unevenly the (Gaussian, block) pairs are spread over the lanes of a wave in the binning stages (config c5's avatar)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import camera, configs, scene as sc, sds_step, synth
from dreamwaltz_g_amd.rasterizer import morton_order

dev = torch.device("cuda:0")
G, res = int(sys.argv[1]) if len(sys.argv) > 1 else 300000, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = configs.TrainConfig(); cfg.device = str(dev)
avatar, N, M = sds_step.build_synthetic_avatar(G, dev, seed=0)
scene = sc.Scene(cfg, avatar, async_pair_count=False).to(dev).eval()
data = camera.make_camera(radius=2.0, azimuth=0.0, elevation=80.0, fovy=55.0, height=res, width=res, device=dev)
pose = synth.random_smpl_inputs(seed=0, device=dev)
with torch.inference_mode():
    g = scene.avatar_forward(smpl_observed_inputs=pose)
    pos = g.positions.clone()
    out = scene.renderer.render(data=data, gaussians=g, return_2d_radii=True)
r = out["radii"].float()
est = torch.where(r > 0, (2 * r / 3 / 8 + 1) ** 2, torch.zeros_like(r))        # ~blocks of the 1-sigma..2-sigma ellipse, crude
print("pairs (exact, ref):", scene.renderer.last_rasterizer.last_num_pairs, "estimate sum", float(est.sum()))
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)
print("radius quantiles", torch.quantile(r[r > 0], q).tolist(), "visible", int((r > 0).sum()))
for name, order in (("index", torch.arange(G, device=dev)), ("morton", morton_order(pos).long())):
    e = est[order]
    pad = (-G) % 64
    w = torch.cat([e, e.new_zeros(pad)]).view(-1, 64)
    print(name, "sum of wave maxima x64 / sum =", float(w.max(dim=1).values.sum() * 64 / e.sum()))
