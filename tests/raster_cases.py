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


# margins of the oracle's threshold-proximity byte (oracle/raster_oracle.c composite_forward): alpha against the 1/255 cut -- relative 2e-5
# (the HIP kernel's exp is the hardware exp2, its quadratic form is contracted differently) plus the splat's own sensitivity to WHERE its
# centre landed: two fp32 projections of one centre differ by an ulp or two of the pixel coordinate (6e-5 px at x = 512), and a sharp splat
# turns that into 1e-4 of alpha -- and the running transmittance against the 1e-4 stop (a product of up to hundreds of such factors)
NEAR = (2e-5, 1e-3, 2.0)
CAUSES = ("alpha_cut", "T_stop", "depth_tie", "power_zero")


def oracle_forward(sc, dtype=np.float32, **over):
    d = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in sc.items()}
    d["opacities"] = d["opacities"].reshape(-1)
    d.setdefault("near", NEAR)
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
    """hip: torch tensors; ref: oracle dict (numpy). Returns dict of error statistics per output.  With the oracle's threshold-proximity
    byte (`near`): the maximum over the CLEAR pixels (no hard threshold within the margins -- the north star's "per-pixel within 1e-4" is
    asserted on these, as a maximum), and the flagged ones counted: how many there are, how many actually differ by more than 1e-4 (a
    flip that happened), their largest error and their causes."""
    out = {}
    near = ref.get("near")
    flagged = ((near & 15) != 0) if near is not None else None
    for k in ("color", "depth", "alpha"):
        a = hip[k].detach().float().cpu().numpy().reshape(ref[k].shape)
        e = np.abs(a - ref[k])
        out[k] = dict(max=float(e.max()) if e.size else 0.0, q999=float(np.quantile(e, 0.999)) if e.size else 0.0,
                      frac_gt_1e4=float((e > 1e-4).mean()) if e.size else 0.0)
        if flagged is not None and e.size:
            ep = e.max(axis=0) if e.ndim == 3 else e              # per pixel
            clear = ep[~flagged]
            flips = flagged & (ep > 1e-4)
            out[k].update(max_clear=float(clear.max()) if clear.size else 0.0, n_clear_gt_1e4=int((clear > 1e-4).sum()),
                          n_flagged=int(flagged.sum()), n_flips=int(flips.sum()), flip_max=float(ep[flips].max()) if flips.any() else 0.0,
                          flip_causes={c: int((flips & (((near >> i) & 1) != 0)).sum()) for i, c in enumerate(CAUSES)})
    out["pixels"] = int(near.size) if near is not None else None
    rh, rr = hip["radii"].cpu().numpy().astype(np.int64), np.asarray(ref["radii"]).astype(np.int64)
    out["radii_equal"] = bool(np.array_equal(rh, rr))
    # where they differ: how many, by how much, and whether visibility (radius > 0) itself differs
    out["radii_diff"] = dict(n=int((rh != rr).sum()), max=int(np.abs(rh - rr).max()) if rh.size else 0,
                             visibility=int(((rh > 0) != (rr > 0)).sum()))
    return out


def check_images(st, name, note=None):
    """The image bar of every rasterizer parity test.  north_star: "per-pixel within 1e-4" -- asserted as a MAXIMUM over every pixel that
    does not sit on one of the algorithm's hard thresholds (alpha < 1/255 skip, T < 1e-4 stop, depth-order tie; the oracle marks them within
    stated relative margins, NEAR).  On a marked pixel a one-ulp difference in exp() may legitimately flip a splat in or out: those flips
    are COUNTED (<= 0.05 % of the image) and bounded (<= 2e-2, the largest contribution a splat at a threshold can make; depth x 10: it
    is not normalised).  Radii = ceil(3 sqrt(lambda_max)) may differ by one on <= 4 Gaussians in 10^5 (an fp32 rounding at an integer
    boundary): each such Gaussian may touch four more / fewer tiles, whose pixels are allowed as a 'radius' class of the same bounds;
    visibility (radius > 0, what the densifier filters on) must agree everywhere."""
    rd = st["radii_diff"]
    assert rd["visibility"] == 0 and rd["max"] <= 1 and rd["n"] <= 4, (name, rd)
    P = st["pixels"]
    for k in ("color", "depth", "alpha"):
        s = st[k]
        if "max_clear" not in s:
            raise AssertionError("image_err_stats without the oracle's threshold byte")
        big = 2e-2 * (10.0 if k == "depth" else 1.0)
        if rd["n"] == 0:
            assert s["max_clear"] <= 1e-4, (name, k, s)
        else:       # pixels of the tiles a radius +-1 adds / removes: counted with the flips
            assert s["n_clear_gt_1e4"] <= rd["n"] * 4 * 256 and s["max_clear"] <= big, (name, k, s, rd)
        assert s["n_flips"] <= max(4, int(5e-4 * P)), (name, k, s)
        assert s["flip_max"] <= big, (name, k, s)
    if note is not None:
        note_parity(note, st)


def note_parity(name, st):
    """One row of gpurun_out/parity_raster.json (copied to profiles/rNN_parity_raster.json at round end)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_raster.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[str(name)] = st
    with open(path, "w") as f:
        json.dump(d, f, indent=1)


def grad_err(a, r):
    """relative L2 error and robust elementwise statistic between a HIP gradient and the oracle's."""
    a = np.asarray(a, np.float64).reshape(-1); r = np.asarray(r, np.float64).reshape(-1)
    nr = np.linalg.norm(r)
    rel_l2 = float(np.linalg.norm(a - r) / max(nr, 1e-30))
    scale = np.abs(r).mean() + 1e-30
    el = np.abs(a - r) / (np.abs(r) + scale)
    return dict(rel_l2=rel_l2, q99=float(np.quantile(el, 0.99)) if el.size else 0.0, max=float(el.max()) if el.size else 0.0)
