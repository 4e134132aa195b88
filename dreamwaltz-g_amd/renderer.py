"""Mirror of GaussianRenderer (/root/reference/core/gaussian/gaussian_renderer.py:9-224): same constructor, same
`build_gaussian_rasterizer(data)` / `render(data, gaussians, return_2d_radii, rasterizer)` contract and output dict,
bound to the HIP rasterizer (rasterizer.py) instead of the CUDA package."""
from typing import Optional

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


class GaussianRenderer:
    def __init__(self, sh_levels=4, bg_color=(0.0, 0.0, 0.0), compute_color_in_rasterizer=True,
                 compute_covariance_in_rasterizer=True) -> None:
        self.sh_levels = sh_levels
        self.bg_color = torch.tensor(bg_color, dtype=torch.float32)
        self.compute_color_in_rasterizer = compute_color_in_rasterizer
        self.compute_covariance_in_rasterizer = compute_covariance_in_rasterizer
        self._bg_dev = {}

    def build_gaussian_rasterizer(self, data: dict, **kwargs) -> GaussianRasterizer:
        """gaussian_renderer.py:23-70.  The reference reads tanfov with .item() (a host sync); a float already on the host
        (data['tanfov_host']) is used when the caller provides one."""
        world_view_matrix = data['extrinsic'][0]
        projection_matrix = data['projection'][0]
        device = world_view_matrix.device
        if 'tanfov_host' in data:
            tanfovy = float(data['tanfov_host'])
            tanfovx = float(data.get('tanfov_x_host', tanfovy))
        else:
            tanfovy = data['tanfov'][0].item()
            tanfovx = data['tanfov_x'][0].item() if 'tanfov_x' in data else tanfovy
        viewmatrix = world_view_matrix.transpose(0, 1)
        projmatrix = viewmatrix @ projection_matrix.transpose(0, 1)
        if device not in self._bg_dev:
            self._bg_dev[device] = self.bg_color.to(device)
        settings = {
            "image_height": data['image_height'], "image_width": data['image_width'], "tanfovx": tanfovx, "tanfovy": tanfovy,
            "bg": self._bg_dev[device], "viewmatrix": viewmatrix, "projmatrix": projmatrix, "sh_degree": self.sh_levels - 1,
            "campos": data['c2w'][0, :3, 3],
        }
        settings.update(kwargs)
        return GaussianRasterizer(raster_settings=GaussianRasterizationSettings(**settings, scale_modifier=1.,
                                                                                prefiltered=False, debug=False))

    def render(self, data: dict, gaussians, return_2d_radii: bool = False,
               rasterizer: Optional[GaussianRasterizer] = None) -> dict:
        if rasterizer is None:
            rasterizer = self.build_gaussian_rasterizer(data=data)
        if gaussians.colors is not None:
            gaussians.sh_features = None     # checklist Q9: the argument is mutated, as in the reference
        means3D = gaussians.positions
        screenspace_points = torch.zeros(means3D.shape[0], 3, dtype=means3D.dtype, requires_grad=True, device=means3D.device)
        if return_2d_radii:
            screenspace_points.retain_grad()
        image, radii, depth, alpha = rasterizer(
            means3D=means3D, means2D=screenspace_points, shs=gaussians.sh_features, colors_precomp=gaussians.colors,
            opacities=gaussians.opacities, scales=gaussians.scales, rotations=gaussians.quaternions,
            cov3D_precomp=gaussians.cov3D)
        outputs = {"image": image.permute(1, 2, 0).unsqueeze(0), "depth": depth.permute(1, 2, 0).unsqueeze(0),
                   "alpha": alpha.permute(1, 2, 0).unsqueeze(0)}
        if return_2d_radii:
            outputs["radii"] = radii
            outputs["viewspace_points"] = screenspace_points
        return outputs
