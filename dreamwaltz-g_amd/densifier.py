"""Adaptive density control of the 3D Gaussians (second half): mirror of
/root/reference/core/gaussian/gaussian_densifier.py:11-411 (DensificationParams :11-78, GaussianDensifier :81-387, build_densifier :390-411)
for parameters that live in the ONE flat optimizer buffer of this path.

  densifier(viewspace_points, radii, spatial_extent, train_step)     gaussian_densifier.py:330-387 -- called by Scene.densify
      update_densification_stats   :201-209   max screen radius, accumulated |d loss / d mean2D| (the rasterizer's `means2D` gradient), count
      densify_and_clone            :231-259   small Gaussians with a large mean screen gradient are duplicated
      densify_and_split            :261-307   large ones are replaced by N = 2 samples of themselves, scales / (0.8 N)
      prune                        :309-328   low opacity / too large on screen / too large in the world
      reset_opacity                :330-339

What differs from the reference is only WHERE the tensors live: the reference replaces nn.Parameters inside a torch.optim.Adam and
re-keys its state dict (:120-180); here `optim.resize_flat_params` lays the flat parameter / gradient / moment buffers out afresh, the
resized Parameters become NEW nn.Parameter objects re-registered under the same names (a Parameter whose `.data` changed shape keeps a stale
AccumulateGrad node; `_rebind` hands the new objects to the owning modules), the Adam moments are carried over row by row (zeros for new rows), and -- as in the reference, whose new
Parameters have `grad is None` at the following `optimizer.step()` -- the resized groups sit out that one step.  Runs on the device the
parameters are on with torch indexing ops (every `densification_interval` steps: not part of the per-step hot path); the three per-step
statistics updates are element-wise torch ops on [N] tensors.  `use_densifier` is off in every shipped recipe (configs/__init__.py:159)."""
from typing import Optional

import torch
import torch.nn as nn

from . import optim
from .rigid import quaternion_to_matrix


class DensificationParams:
    """gaussian_densifier.py:11-78 (same defaults, same derived iteration counts)."""

    def __init__(self, max_iteration: int, densify_from_iter: Optional[int] = None, densify_until_iter: Optional[int] = None,
                 densification_interval: Optional[int] = None, opacity_reset_interval: Optional[int] = None,
                 densify_grad_threshold: float = 0.0002, prune_opacity_threshold: float = 0.005,
                 densify_screen_size_threshold: float = 20.0, densification_percent_distinction: float = 0.01,
                 disable_densify_clone: bool = False, disable_densify_split: bool = False, disable_prune: bool = False,
                 disable_reset: bool = False, enable_grad_prune: bool = False):
        if densify_from_iter is None:
            densify_from_iter = int(max_iteration * 500 / 15000)
        if densify_until_iter is None:
            densify_until_iter = int(max_iteration * 7000 / 15000)
        if densification_interval is None:
            densification_interval = int(max_iteration * 100 / 15000)
        if opacity_reset_interval is None:
            opacity_reset_interval = int(max_iteration * 3000 / 15000)
        self.max_iteration = max_iteration
        self.densify_from_iter, self.densify_until_iter = densify_from_iter, densify_until_iter
        self.densification_interval, self.opacity_reset_interval = densification_interval, opacity_reset_interval
        self.densify_grad_threshold, self.prune_opacity_threshold = densify_grad_threshold, prune_opacity_threshold
        self.densify_screen_size_threshold = densify_screen_size_threshold
        self.densification_percent_distinction = densification_percent_distinction
        self.disable_densify_clone, self.disable_densify_split = disable_densify_clone, disable_densify_split
        self.disable_prune, self.disable_reset = disable_prune, disable_reset
        self.enable_grad_prune = enable_grad_prune


class GaussianDensifier:
    def __init__(self, model, params: DensificationParams, optimizers: "optim.FlatOptimizerDict", optimizer_name: str = "avatar"):
        self.model, self.params = model, params
        self.device = model._positions.device
        self.optimizers = optimizers                                   # the dict avatar.get_optimizer(cfg) returned (shared flat buffers)
        self.optimizer = optimizers[optimizer_name]
        n = self.model._n_points
        self.points_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        self.max_radii2D = torch.zeros((n,), device=self.device)
        self.spatial_extent = None
        self.densify_from_iter, self.densify_until_iter = params.densify_from_iter, params.densify_until_iter
        self.densification_interval, self.opacity_reset_interval = params.densification_interval, params.opacity_reset_interval
        self.max_grad = params.densify_grad_threshold
        self.max_screen_size = params.densify_screen_size_threshold
        self.percent_dense = params.densification_percent_distinction
        self.min_opacity = params.prune_opacity_threshold
        self.disable_densify_clone, self.disable_densify_split = params.disable_densify_clone, params.disable_densify_split
        self.disable_prune, self.disable_reset = params.disable_prune, params.disable_reset
        self.enable_grad_prune = params.enable_grad_prune
        # GaussianOptimizer.param_names (gaussian_optimizer.py:60-90): the per-Gaussian parameters that carry an Adam group
        self.params_to_densify = [pg["name"] for pg in self.optimizer.param_groups if pg.get("name") in
                                  ("positions", "sh_features_dc", "sh_features_rest", "opacities", "scales", "quaternions")]
        self.last_report = None

    # -- optimizer surgery (gaussian_densifier.py:120-180) on the flat buffers ----------------------------------------------------------
    def _param(self, name) -> nn.Parameter:
        return getattr(self.model, "_" + name)

    def _rebind(self, renamed):
        """update_model (:182-187): the model's attributes point at the new Parameter objects."""
        for name in self.params_to_densify:
            old = self._param(name)
            if old in renamed:
                setattr(self.model, "_" + name, renamed[old])

    def _moments(self, p):
        buf = self.optimizers.buffers
        i = [id(q) for q in self.optimizers.params].index(id(p))
        off, n = buf.slices[i]
        return buf.m[off:off + n].view_as(p.data), buf.v[off:off + n].view_as(p.data)

    def _prune_optimizer(self, mask):
        new = {}
        for name in self.params_to_densify:
            p = self._param(name)
            m, v = self._moments(p)
            new[p] = (p.data[mask], m[mask], v[mask])
        self._rebind(optim.resize_flat_params(self.optimizers, new))

    def cat_tensors_to_optimizer(self, tensors_dict):
        new = {}
        for name in self.params_to_densify:
            p = self._param(name)
            ext = tensors_dict[name]
            m, v = self._moments(p)
            new[p] = (torch.cat((p.data, ext), dim=0), torch.cat((m, torch.zeros_like(ext)), dim=0), torch.cat((v, torch.zeros_like(ext)), dim=0))
        self._rebind(optim.resize_flat_params(self.optimizers, new))

    def replace_tensor_to_optimizer(self, tensor, name):
        p = self._param(name)
        self._rebind(optim.resize_flat_params(self.optimizers, {p: (tensor, None, None)}))

    def update_model(self):
        self.model._n_points = len(self.model._positions)
        if hasattr(self.model, "invalidate_caches"):
            self.model.invalidate_caches()

    def densification_postfix(self, tensors_dict):
        self.cat_tensors_to_optimizer(tensors_dict)
        self.update_model()
        n = self.model._n_points
        self.points_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        self.max_radii2D = torch.zeros((n,), device=self.device)

    # -- statistics (:201-209) -----------------------------------------------------------------------------------------------------------
    def update_densification_stats(self, viewspace_point_tensor, radii, visibility_filter):
        grad = viewspace_point_tensor.grad
        vis = visibility_filter
        self.max_radii2D = torch.where(vis, torch.maximum(self.max_radii2D, radii.to(self.max_radii2D.dtype)), self.max_radii2D)
        self.points_gradient_accum = self.points_gradient_accum + torch.where(vis, torch.norm(grad[:, :2], dim=-1), 0.0).unsqueeze(-1)
        self.denom = self.denom + vis.to(self.denom.dtype).unsqueeze(-1)

    # -- decisions ----------------------------------------------------------------------------------------------------------------------
    def get_prune_mask(self, extent: float, grads: Optional[torch.Tensor] = None):
        """:211-229.  NB the reference reads `model.get_opacities()`: an avatar without per-Gaussian opacity PARAMETERS (DreamWaltzG: its
        opacities come out of the MLP) fails there with an AttributeError on None -- pruning needs `--render.densify_disable_prune True`
        for it, here as there."""
        with torch.no_grad():
            scales = self.model.get_scales()
            opacities = self.model.get_opacities()
        prune_mask = (opacities < self.min_opacity).squeeze()
        big_points_vs = self.max_radii2D > self.max_screen_size
        big_points_ws = scales.max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
        if grads is not None:
            prune_mask = torch.logical_or(prune_mask, torch.norm(grads, dim=-1) >= self.max_grad)
        return prune_mask

    def _carry_rows(self, selected, repeat=1):
        """Frozen per-Gaussian rows that follow their Gaussians (:251-259,297-305): _lbs_weights, vertex_indices."""
        if hasattr(self.model, 'vertex_indices'):
            vi = self.model.vertex_indices
            self.model.vertex_indices = torch.cat((vi, vi[selected.detach().cpu()].repeat(repeat)), dim=0)
        if hasattr(self.model, '_lbs_weights') and self.model._lbs_weights is not None:
            assert self.model._lbs_weights.requires_grad is False
            old = self.model._lbs_weights.data
            self.model._lbs_weights.data = torch.cat((old, old[selected].repeat(repeat, 1)), dim=0)

    def densify_and_clone(self, grads: torch.Tensor, extent: float):
        with torch.no_grad():
            scales = self.model.get_scales()
        selected = torch.norm(grads, dim=-1) >= self.max_grad
        selected = torch.logical_and(selected, torch.max(scales, dim=1).values <= self.percent_dense * extent)
        tensors_dict = {name: self._param(name).data[selected] for name in self.params_to_densify}
        self.densification_postfix(tensors_dict)
        self._carry_rows(selected, 1)

    def densify_and_split(self, grads: torch.Tensor, extent: float, N: int = 2, samples: Optional[torch.Tensor] = None):
        """`samples` (test hook): the N x selected draws of torch.normal(0, scales) the reference makes on the global generator."""
        with torch.no_grad():
            quaternions = self.model.get_quaternions()
            scales = self.model.get_scales()
        padded_grad = torch.zeros(self.model._n_points, device=self.device)
        padded_grad[:grads.shape[0]] = grads.squeeze()
        selected = padded_grad >= self.max_grad
        selected = torch.logical_and(selected, torch.max(scales, dim=1).values > self.percent_dense * extent)
        stds = scales[selected].repeat(N, 1)
        if samples is None:
            samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=self.device), std=stds)
        rots = quaternion_to_matrix(quaternions[selected]).repeat(N, 1, 1)
        tensors_dict = {}
        for name in self.params_to_densify:
            p = self._param(name).data[selected]
            tensors_dict[name] = p.repeat(N, *([1] * (p.ndim - 1)))
        tensors_dict['positions'] = tensors_dict['positions'] + torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1)
        tensors_dict['scales'] = self.model.scale_inverse_activation(self.model.scale_activation(tensors_dict['scales']) / (0.8 * N))
        self.densification_postfix(tensors_dict)
        self._carry_rows(selected, N)
        prune_mask = torch.cat((selected, torch.zeros(N * int(selected.sum()), device=self.device, dtype=torch.bool)))
        self.prune(prune_mask)

    def prune(self, prune_mask: torch.Tensor):
        valid = ~prune_mask
        self._prune_optimizer(valid)
        self.update_model()
        self.points_gradient_accum = self.points_gradient_accum[valid]
        self.denom = self.denom[valid]
        self.max_radii2D = self.max_radii2D[valid]
        if hasattr(self.model, 'vertex_indices'):
            self.model.vertex_indices = self.model.vertex_indices[valid.detach().cpu()]
        if hasattr(self.model, '_lbs_weights') and self.model._lbs_weights is not None:
            assert self.model._lbs_weights.requires_grad is False
            self.model._lbs_weights.data = self.model._lbs_weights.data[valid]

    def reset_opacity(self, value: float = 0.01):
        with torch.no_grad():
            opacities = self.model.get_opacities()
            opacities = self.model.opacity_inverse_activation(torch.min(opacities, torch.ones_like(opacities) * value))
        assert self.model._opacities is not None
        self.replace_tensor_to_optimizer(opacities, "opacities")

    @torch.no_grad()
    def __call__(self, viewspace_points: torch.Tensor, radii: torch.Tensor, spatial_extent: float, train_step: int, split_samples=None):
        """:341-387."""
        if train_step >= self.densify_until_iter:
            return
        self.update_densification_stats(viewspace_points, radii, visibility_filter=radii > 0)
        if train_step > self.densify_from_iter and train_step % self.densification_interval == 0:
            if spatial_extent is None:
                spatial_extent = self.spatial_extent
            grads = self.points_gradient_accum / self.denom
            grads[grads.isnan()] = 0.0
            n_points = self.model._n_points
            if not self.enable_grad_prune:
                if not self.disable_densify_clone:
                    self.densify_and_clone(grads=grads, extent=spatial_extent)
                if not self.disable_densify_split:
                    self.densify_and_split(grads=grads, extent=spatial_extent, samples=split_samples)
            new_points = self.model._n_points - n_points
            if not self.disable_prune:
                if self.enable_grad_prune:
                    prune_mask = self.get_prune_mask(extent=spatial_extent, grads=grads)
                    grad_prune_iters = (self.densify_until_iter - self.densify_from_iter) / 3
                    if train_step > (self.densify_from_iter + grad_prune_iters):
                        self.enable_grad_prune = False
                else:
                    prune_mask = self.get_prune_mask(extent=spatial_extent)
                self.prune(prune_mask)
            pruned_points = n_points + new_points - self.model._n_points
            self.last_report = (n_points, new_points, pruned_points, self.model._n_points)
        if not self.disable_reset and train_step % self.opacity_reset_interval == 0:
            self.reset_opacity()


def build_densifier(model, optimizers, cfg, optimizer_name: str = "avatar") -> GaussianDensifier:
    """gaussian_densifier.py:390-411.  `optimizers`: the dict `model.get_optimizer(cfg)` returned."""
    r = cfg.render
    params = DensificationParams(max_iteration=cfg.optim.iters, densify_from_iter=r.densify_from_iter, densify_until_iter=r.densify_until_iter,
                                 densify_grad_threshold=r.densify_grad_threshold, disable_densify_clone=r.densify_disable_clone,
                                 disable_densify_split=r.densify_disable_split, disable_prune=r.densify_disable_prune,
                                 disable_reset=r.densify_disable_reset, enable_grad_prune=r.enable_grad_prune)
    return GaussianDensifier(model=model, params=params, optimizers=optimizers, optimizer_name=optimizer_name)
