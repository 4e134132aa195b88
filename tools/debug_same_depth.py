"""Debug aid: where the depth-tie scene differs from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dwg_import  # noqa
from tests import raster_cases as rc
from dreamwaltz_g_amd import camera

G, H, W = 2000, 128, 128
sc = rc.make_scene(G, H, W, same_depth=True)
view = sc["viewmatrix"]
pos = sc["means3D"]
z = (pos @ view[:3, 2]) + view[3, 2] if view.shape == (4, 4) else None
print("view matrix col/row layout check; depth via column 2:", None if z is None else (float(z.min()), float(z.max()), len(torch.unique(z))))
z2 = (pos @ view[2, :3]) + view[2, 3]
print("depth via row 2:", float(z2.min()), float(z2.max()), len(torch.unique(z2)))
ref = rc.oracle_forward(sc)
out = rc.hip_render(sc)
e = np.abs(out["color"].cpu().numpy() - ref["color"]).max(0)
bad = np.argwhere(e > 1e-4)
print("bad pixels", len(bad), "of", H * W)
blocks = {}
for y, x in bad:
    blocks[(y // 8, x // 8)] = blocks.get((y // 8, x // 8), 0) + 1
print("bad 8x8 blocks", len(blocks), sorted(blocks.items())[:40])
print("depth err max", float(np.abs(out["depth"].cpu().numpy() - ref["depth"]).max()), "alpha err max", float(np.abs(out["alpha"].cpu().numpy() - ref["alpha"]).max()))
# same composite order, no ties: push Gaussian i back by i * 1e-4 along the view axis
look = camera.make_camera(height=H, width=W, azimuth=30.0, elevation=80.0)["c2w"][0, :3, 2]
for sign in (1.0, -1.0):
    sc2 = dict(sc); sc2["means3D"] = (pos + sign * look[None, :] * (torch.arange(G).float()[:, None] * 1e-4)).contiguous()
    r2 = rc.oracle_forward(sc2); o2 = rc.hip_render(sc2)
    print("untied, sign", sign, rc.image_err_stats(o2, r2)["color"], "vs tied oracle", float(np.abs(r2["color"] - ref["color"]).max()))
# oracle in float64 on the tied scene
r64 = rc.oracle_forward(sc, dtype=np.float64)
print("oracle f32 vs f64 on the tied scene: max", float(np.abs(r64["color"] - ref["color"]).max()), "hip vs f64", float(np.abs(out["color"].cpu().numpy() - r64["color"]).max()))
