"""Camera matrices in the reference's conventions (mirrors of /root/reference/data/camera/utils.py; pinned against the imported
reference by tests/test_host_golden_r2.py):
  get_tan_half_fov :18-21   angle2sphere :62-76   to_extrinsic :79-113   to_projection :149-201
Camera space: x right, y up, +z forward; the projection flips y (K[1,1] < 0) and maps z to [-1,1] (SURVEY.md checklist Q10).
`make_camera` assembles the dict keys GaussianRenderer.build_gaussian_rasterizer reads (gaussian_renderer.py:23-41).
Host-side setup code: runs on whatever device its inputs live on (a handful of 4x4 values per step).
"""
from typing import Optional

import torch
from torch import Tensor


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


def get_tan_half_fov(fov: Tensor, degrees: bool = True):
    if degrees:
        fov = fov * torch.pi / 180.0
    return torch.tan(fov / 2)


def angle2sphere(radius: Tensor, elevation: Tensor, azimuth: Tensor, degrees: bool = True) -> Tensor:
    if degrees:
        azimuth = azimuth * torch.pi / 180.0
        elevation = elevation * torch.pi / 180.0
    return torch.stack([radius * torch.sin(elevation) * torch.sin(azimuth), radius * torch.cos(elevation),
                        radius * torch.sin(elevation) * torch.cos(azimuth)], dim=-1)


def to_extrinsic(radius: Tensor, azimuth: Tensor, elevation: Tensor, at_vector=((0, 0, 0),), up_vector=((0, 1, 0),)):
    """-> (extrinsic [B,4,4] world->camera, c2w [B,4,4])."""
    batch_size, device = radius.shape[0], radius.device
    if not isinstance(up_vector, Tensor):
        up_vector = torch.tensor(up_vector, dtype=torch.float, device=device).repeat(batch_size, 1)
    if not isinstance(at_vector, Tensor):
        at_vector = torch.tensor(at_vector, dtype=torch.float, device=device).repeat(batch_size, 1)
    spherical_camera_position = angle2sphere(radius=radius, azimuth=azimuth, elevation=elevation)
    camera_position = at_vector + spherical_camera_position
    lookat_vector = safe_normalize(-spherical_camera_position)
    right_vector = safe_normalize(torch.cross(lookat_vector, up_vector, dim=-1))
    up_vector = safe_normalize(torch.cross(right_vector, lookat_vector, dim=-1))
    c2w = torch.eye(4, dtype=torch.float, device=device).unsqueeze(0).repeat(batch_size, 1, 1)
    c2w[:, :3, :3] = torch.stack((right_vector, up_vector, lookat_vector), dim=-1)
    c2w[:, :3, 3] = camera_position
    return torch.inverse(c2w), c2w


def to_projection(tanfov: Tensor, z_near: float, z_far: float, aspect_wh: float = 1.0, z_range=(-1, 1),
                  tanfov_x: Optional[Tensor] = None) -> Tensor:
    N, device = tanfov.shape[0], tanfov.device
    max_y = tanfov * z_near
    min_y = -max_y
    max_x = max_y * aspect_wh if tanfov_x is None else tanfov_x * z_near
    min_x = -max_x
    K = torch.zeros((N, 4, 4), dtype=torch.float32, device=device)
    K[:, 0, 0] = 2.0 * z_near / (max_x - min_x)
    K[:, 0, 2] = (max_x + min_x) / (max_x - min_x)
    K[:, 1, 1] = -2.0 * z_near / (max_y - min_y)                 # y flipped
    K[:, 1, 2] = (max_y + min_y) / (max_y - min_y)
    if z_range == (0, 1):
        K[:, 2, 2] = z_far / (z_far - z_near)
        K[:, 2, 3] = -(z_far * z_near) / (z_far - z_near)
    else:
        K[:, 2, 2] = (z_far + z_near) / (z_far - z_near)
        K[:, 2, 3] = -(2 * z_far * z_near) / (z_far - z_near)
    K[:, 3, 2] = 1.0
    return K


def make_camera(radius=2.0, azimuth=30.0, elevation=80.0, fovy=55.0, height=256, width=256, z_near=0.01, z_far=1000.0,
                device="cpu", dtype=torch.float32, at=(0.0, 0.0, 0.0)):
    """One camera as the reference's dataloader would hand it over.  The per-step SCALARS (tanfov, radius, azimuth, elevation) stay on
    the host -- the reference reads them with .item() every step (gaussian_renderer.py:28, trainer.py:713) -- the matrices go to
    `device`."""
    t = lambda v: torch.tensor([float(v)])  # noqa: E731
    extrinsic, c2w = to_extrinsic(t(radius), t(azimuth), t(elevation), at_vector=(tuple(at),))
    tanfov = get_tan_half_fov(t(fovy))
    projection = to_projection(tanfov, z_near, z_far, aspect_wh=width / height)
    out = {
        "extrinsic": extrinsic.to(dtype).to(device), "c2w": c2w.to(dtype).to(device), "projection": projection.to(dtype).to(device),
        "tanfov": tanfov.to(dtype), "radius": t(radius), "azimuth": t(azimuth), "elevation": t(elevation),
        "image_height": height, "image_width": width,
    }
    if width != height:
        out["tanfov_x"] = (tanfov * width / height).to(dtype)
    return out


def raster_matrices(cam):
    """viewmatrix / projmatrix / campos exactly as gaussian_renderer.py:38-41 builds them."""
    viewmatrix = cam["extrinsic"][0].transpose(0, 1).contiguous()
    projmatrix = (viewmatrix @ cam["projection"][0].transpose(0, 1)).contiguous()
    campos = cam["c2w"][0, :3, 3].contiguous()
    tanfovy = float(cam["tanfov"][0])
    tanfovx = float(cam["tanfov_x"][0]) if "tanfov_x" in cam else tanfovy
    return viewmatrix, projmatrix, campos, tanfovx, tanfovy
