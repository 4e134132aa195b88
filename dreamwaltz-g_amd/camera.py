"""Synthetic camera following the reference's matrix conventions (restated, not imported):
  angle2sphere / to_extrinsic : /root/reference/data/camera/utils.py:62-113
  to_projection               : /root/reference/data/camera/utils.py:149-201
Camera space: x right, y up, +z forward; projection flips y (K[1,1] < 0) and maps z to [-1,1]
(SURVEY.md checklist Q10).  Returns the same dict keys GaussianRenderer.build_gaussian_rasterizer
reads (gaussian_renderer.py:23-41).
"""
import math

import torch


def _normalize(v, eps=1e-20):
    return v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=eps))


def make_camera(radius=2.0, azimuth=30.0, elevation=80.0, fovy=55.0, height=256, width=256,
                z_near=0.01, z_far=1000.0, device="cpu", dtype=torch.float32, at=(0.0, 0.0, 0.0)):
    az, el = math.radians(azimuth), math.radians(elevation)
    sph = torch.tensor([radius * math.sin(el) * math.sin(az), radius * math.cos(el),
                        radius * math.sin(el) * math.cos(az)], dtype=torch.float64)
    at_v = torch.tensor(at, dtype=torch.float64)
    cam_pos = at_v + sph
    look = _normalize(-sph)
    up0 = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    right = _normalize(torch.linalg.cross(look, up0))
    up = _normalize(torch.linalg.cross(right, look))
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, :3] = torch.stack((right, up, look), dim=-1)
    c2w[:3, 3] = cam_pos
    extrinsic = torch.inverse(c2w)
    tanfov = math.tan(math.radians(fovy) * 0.5)
    max_y = tanfov * z_near
    max_x = max_y * (width / height)
    K = torch.zeros(4, 4, dtype=torch.float64)
    K[0, 0] = 2.0 * z_near / (2 * max_x)
    K[1, 1] = -2.0 * z_near / (2 * max_y)
    K[2, 2] = (z_far + z_near) / (z_far - z_near)
    K[2, 3] = -(2 * z_far * z_near) / (z_far - z_near)
    K[3, 2] = 1.0
    out = {
        "extrinsic": extrinsic[None].to(dtype).to(device),
        "c2w": c2w[None].to(dtype).to(device),
        "projection": K[None].to(dtype).to(device),
        "tanfov": torch.tensor([tanfov], dtype=dtype, device=device),
        "image_height": height,
        "image_width": width,
    }
    if width != height:
        out["tanfov_x"] = torch.tensor([tanfov * width / height], dtype=dtype, device=device)
    return out


def raster_matrices(cam):
    """viewmatrix / projmatrix / campos exactly as gaussian_renderer.py:38-41 builds them."""
    viewmatrix = cam["extrinsic"][0].transpose(0, 1).contiguous()
    projmatrix = (viewmatrix @ cam["projection"][0].transpose(0, 1)).contiguous()
    campos = cam["c2w"][0, :3, 3].contiguous()
    tanfovy = float(cam["tanfov"][0])
    tanfovx = float(cam["tanfov_x"][0]) if "tanfov_x" in cam else tanfovy
    return viewmatrix, projmatrix, campos, tanfovx, tanfovy
