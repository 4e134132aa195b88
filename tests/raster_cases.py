"""Shared scene builders / comparison helpers for the rasterizer parity tests (HIP vs CPU oracle)."""
import numpy as np
import torch

import dreamwaltz_g_amd.camera as camera
import dreamwaltz_g_amd.synth as synth
from oracle import raster as oraster


def make_scene(G, H, W, seed=0, scale_mul=1.0, opacity_range=None, cluster=None, azimuth=30.0, elevation=80.0,
               same_depth=False):
    g = synth.random_gaussians(G, seed=seed, opacity_range=opacity_range)
    if cluster is not None:  # squeeze all Gaussians into a small box -> very long tile lists
        g["positions"] = g["positions"] * cluster
    cam = camera.make_camera(height=H, width=W, azimuth=azimuth, elevation=elevation)
    view, proj, campos, tfx, tfy = camera.raster_matrices(cam)
    if same_depth:  # put every Gaussian on one plane orthogonal to the view direction -> exact depth ties
        look = cam["c2w"][0, :3, 2]
        pos = g["positions"]
        g["positions"] = (pos - (pos @ look)[:, None] * look[None, :]).contiguous()
    return dict(means3D=g["positions"], opacities=g["opacities"], colors=g["colors"], scales=g["scales"] * scale_mul,
                rotations=g["quaternions"], viewmatrix=view, projmatrix=proj, campos=campos, tanfovx=tfx,
                tanfovy=tfy, bg=torch.tensor([0.5, 0.5, 0.5]), H=H, W=W)


def oracle_forward(sc, dtype=np.float32, **over):
    d = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in sc.items()}
    d["opacities"] = d["opacities"].reshape(-1)
    d.update(over)
    return oraster.forward(dtype=dtype, **d)


def oracle_backward(sc, g_color, g_depth=None, g_alpha=None, dtype=np.float64, **over):
    d = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in sc.items()}
    d["opacities"] = d["opacities"].reshape(-1)
    d.update(over)
    return oraster.backward(dtype=dtype, g_color=g_color, g_depth=g_depth, g_alpha=g_alpha, **d)


def hip_render(sc, device="cuda", requires_grad=False, use_sh=None, sh_degree=0, use_cov=None, visit_order=None):
    from dreamwaltz_g_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    t = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sc.items()}
    rs = GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t["bg"],
        scale_modifier=1.0, viewmatrix=t["viewmatrix"], projmatrix=t["projmatrix"], sh_degree=sh_degree,
        campos=t["campos"], prefiltered=False, debug=False)
    leaves = {}
    for k in ("means3D", "opacities", "colors", "scales", "rotations"):
        leaves[k] = t[k].clone().requires_grad_(requires_grad)
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=requires_grad)
    leaves["means2D"] = means2D
    kw = dict(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"])
    if use_sh is not None:
        leaves["shs"] = use_sh.to(device).clone().requires_grad_(requires_grad)
        kw.update(shs=leaves["shs"], colors_precomp=None)
    else:
        kw.update(shs=None, colors_precomp=leaves["colors"])
    if use_cov is not None:
        leaves["cov3D"] = use_cov.to(device).clone().requires_grad_(requires_grad)
        kw.update(cov3D_precomp=leaves["cov3D"], scales=None, rotations=None)
    else:
        kw.update(scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    color, radii, depth, alpha = GaussianRasterizer(rs, visit_order=visit_order)(**kw)
    return dict(color=color, radii=radii, depth=depth, alpha=alpha, leaves=leaves)


def image_err_stats(hip, ref):
    """hip: torch tensors; ref: oracle dict (numpy). Returns dict of error statistics per output."""
    out = {}
    for k in ("color", "depth", "alpha"):
        a = hip[k].detach().float().cpu().numpy().reshape(ref[k].shape)
        e = np.abs(a - ref[k])
        out[k] = dict(max=float(e.max()) if e.size else 0.0, q999=float(np.quantile(e, 0.999)) if e.size else 0.0,
                      frac_gt_1e4=float((e > 1e-4).mean()) if e.size else 0.0)
    rh, rr = hip["radii"].cpu().numpy().astype(np.int64), np.asarray(ref["radii"]).astype(np.int64)
    out["radii_equal"] = bool(np.array_equal(rh, rr))
    # where they differ: how many, by how much, and whether visibility (radius > 0) itself differs
    out["radii_diff"] = dict(n=int((rh != rr).sum()), max=int(np.abs(rh - rr).max()) if rh.size else 0,
                             visibility=int(((rh > 0) != (rr > 0)).sum()))
    return out


def grad_err(a, r):
    """relative L2 error and robust elementwise statistic between a HIP gradient and the oracle's."""
    a = np.asarray(a, np.float64).reshape(-1); r = np.asarray(r, np.float64).reshape(-1)
    nr = np.linalg.norm(r)
    rel_l2 = float(np.linalg.norm(a - r) / max(nr, 1e-30))
    scale = np.abs(r).mean() + 1e-30
    el = np.abs(a - r) / (np.abs(r) + scale)
    return dict(rel_l2=rel_l2, q99=float(np.quantile(el, 0.99)) if el.size else 0.0, max=float(el.max()) if el.size else 0.0)
