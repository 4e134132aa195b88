"""Adaptive density control of the free 3D Gaussians (second half) on the ONE flat optimizer buffer of this path.

The behaviour is the reference's (/root/reference/core/gaussian/gaussian_densifier.py: schedule :11-78, statistics :201-209, clone :231-259,
split :261-307, prune :211-229 / :309-328, opacity reset :330-339, driver :341-387; pinned by tests/test_densifier.py against a golden
captured from the reference's own class).  The mechanism is this design's:

  * A density update is decided ONCE, on the current rows, and expressed as a `RowPlan`: for every row of the NEXT Gaussian set the row of the
    current set it descends from, whether it continues that Gaussian (its Adam moments travel with it) or is a new one (moments start at
    zero), and whether it is a split sample (position moved by a draw from the parent's covariance, scale divided by 0.8 N).  Cloning,
    splitting, dropping the split parents and pruning are all selections on that one index vector; the reference's order of rows
    (survivors, clones, first samples, second samples) falls out of how the segments are concatenated.
  * The plan is then committed by ONE gather per tensor over the flat parameter / moment buffers and ONE re-layout of them
    (`optim.resize_flat_params`), instead of a parameter-by-parameter cat / index / re-key sequence per stage (four re-layouts per update).
    Resized Parameters become new nn.Parameter objects (a leaf whose storage changed shape keeps a stale AccumulateGrad node) which
    `_adopt` hands to the owning module; the resized groups sit out the following optimizer step, as the reference's do (grad is None).
  * Quantities a decision needs about rows that do not exist yet (the pruning test on split samples) are computed from the parent row with
    the same activation round trip the reference's accessors would apply to the stored value, so that thresholds fall the same way.

Per step only three element-wise updates of [N] statistics run (torch ops; the update itself happens every `densification_interval`
steps and is not part of the per-step hot path).  `use_densifier` is off in the shipped recipes (configs/__init__.py:159)."""
from typing import Dict, NamedTuple, Optional

import torch
import torch.nn as nn

from . import optim
from .rigid import quaternion_to_matrix

_PER_GAUSSIAN = ("positions", "sh_features_dc", "sh_features_rest", "opacities", "scales", "quaternions")


class DensificationParams:
    """Schedule and thresholds; the iteration counts default to fixed fractions of the run length (gaussian_densifier.py:11-78)."""

    _FRACTIONS = dict(densify_from_iter=500, densify_until_iter=7000, densification_interval=100, opacity_reset_interval=3000)

    def __init__(self, max_iteration: int, densify_from_iter: Optional[int] = None, densify_until_iter: Optional[int] = None,
                 densification_interval: Optional[int] = None, opacity_reset_interval: Optional[int] = None,
                 densify_grad_threshold: float = 0.0002, prune_opacity_threshold: float = 0.005,
                 densify_screen_size_threshold: float = 20.0, densification_percent_distinction: float = 0.01,
                 disable_densify_clone: bool = False, disable_densify_split: bool = False, disable_prune: bool = False,
                 disable_reset: bool = False, enable_grad_prune: bool = False):
        given = dict(densify_from_iter=densify_from_iter, densify_until_iter=densify_until_iter,
                     densification_interval=densification_interval, opacity_reset_interval=opacity_reset_interval)
        self.max_iteration = max_iteration
        for key, per_15k in self._FRACTIONS.items():
            setattr(self, key, int(max_iteration * per_15k / 15000) if given[key] is None else given[key])
        self.densify_grad_threshold, self.prune_opacity_threshold = densify_grad_threshold, prune_opacity_threshold
        self.densify_screen_size_threshold = densify_screen_size_threshold
        self.densification_percent_distinction = densification_percent_distinction
        self.disable_densify_clone, self.disable_densify_split = disable_densify_clone, disable_densify_split
        self.disable_prune, self.disable_reset = disable_prune, disable_reset
        self.enable_grad_prune = enable_grad_prune


class RowPlan(NamedTuple):
    """The next Gaussian set in terms of the current one (all tensors have one entry per NEXT row)."""
    src: torch.Tensor        # int64: the current row this one descends from
    carried: torch.Tensor    # bool: continues that Gaussian (moments kept) -- else a new Gaussian (moments zero)
    sample: torch.Tensor     # bool: a split sample (moved / shrunk)
    offset: torch.Tensor     # float [rows, 3]: world-space displacement (zero except for split samples)

    def take(self, alive: torch.Tensor) -> "RowPlan":
        return RowPlan(self.src[alive], self.carried[alive], self.sample[alive], self.offset[alive])


class GaussianDensifier:
    def __init__(self, model, params: DensificationParams, optimizers: "optim.FlatOptimizerDict", optimizer_name: str = "avatar"):
        self.model, self.params = model, params
        self.device = model._positions.device
        self.optimizers = optimizers                                   # the dict avatar.get_optimizer(cfg) returned (shared flat buffers)
        self.optimizer = optimizers[optimizer_name]
        self.densify_from_iter, self.densify_until_iter = params.densify_from_iter, params.densify_until_iter
        self.densification_interval, self.opacity_reset_interval = params.densification_interval, params.opacity_reset_interval
        self.enable_grad_prune = params.enable_grad_prune             # switches itself off after a third of the densification window
        self.split_factor = 2                                          # samples per split Gaussian
        self.spatial_extent = None
        # the per-Gaussian tensors that carry an Adam group of this optimizer, in group order (gaussian_optimizer.py:60-90)
        self.params_to_densify = [pg["name"] for pg in self.optimizer.param_groups if pg.get("name") in _PER_GAUSSIAN]
        self._zero_stats(self.model._n_points)
        self.last_report = None

    # -- per-step statistics ------------------------------------------------------------------------------------------------------------
    def _zero_stats(self, n):
        self.points_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        self.max_radii2D = torch.zeros((n,), device=self.device)

    def update_densification_stats(self, viewspace_point_tensor, radii, visibility_filter):
        """Largest screen radius seen, summed length of the screen-space mean gradient and the number of frames a Gaussian was visible in."""
        seen = visibility_filter
        step_len = viewspace_point_tensor.grad[:, :2].norm(dim=-1)
        self.max_radii2D = torch.where(seen, torch.maximum(self.max_radii2D, radii.to(self.max_radii2D.dtype)), self.max_radii2D)
        self.points_gradient_accum = self.points_gradient_accum + torch.where(seen, step_len, 0.0).unsqueeze(-1)
        self.denom = self.denom + seen.to(self.denom.dtype).unsqueeze(-1)

    # -- the flat buffers ---------------------------------------------------------------------------------------------------------------
    def _param(self, name) -> nn.Parameter:
        return getattr(self.model, "_" + name)

    def _moments(self, p):
        buf = self.optimizers.buffers
        i = [id(q) for q in self.optimizers.params].index(id(p))
        off, n = buf.slices[i]
        return buf.m[off:off + n].view_as(p.data), buf.v[off:off + n].view_as(p.data)

    def _adopt(self, renamed: Dict[nn.Parameter, nn.Parameter]):
        for name in self.params_to_densify:
            old = self._param(name)
            if old in renamed:
                setattr(self.model, "_" + name, renamed[old])
        self.model._n_points = len(self.model._positions)
        if hasattr(self.model, "invalidate_caches"):
            self.model.invalidate_caches()

    def _commit(self, plan: RowPlan):
        """One gather per tensor, one re-layout of the flat buffers; the frozen per-Gaussian rows (skinning weights, bound vertices) follow."""
        src, fresh = plan.src, ~plan.carried
        staged = {}
        for name in self.params_to_densify:
            p = self._param(name)
            m, v = self._moments(p)
            rows, m_rows, v_rows = p.data[src], m[src], v[src]
            m_rows[fresh] = 0.0
            v_rows[fresh] = 0.0
            if name == "positions":
                rows = rows + plan.offset
            elif name == "scales" and bool(plan.sample.any()):
                rows[plan.sample] = self._shrunk(rows[plan.sample])
            staged[p] = (rows, m_rows, v_rows)
        self._adopt(optim.resize_flat_params(self.optimizers, staged))
        if hasattr(self.model, 'vertex_indices'):
            self.model.vertex_indices = self.model.vertex_indices[src.cpu()]
        w = getattr(self.model, '_lbs_weights', None)
        if w is not None:
            assert w.requires_grad is False
            w.data = w.data[src]

    def _shrunk(self, stored_scales):
        """Stored (pre-activation) scale of a split sample: the parent's world-space scale divided by 0.8 N."""
        return self.model.scale_inverse_activation(self.model.scale_activation(stored_scales) / (0.8 * self.split_factor))

    # -- decisions, all on the CURRENT rows ---------------------------------------------------------------------------------------------
    def _plan_growth(self, grads: torch.Tensor, extent: float, samples: Optional[torch.Tensor]) -> RowPlan:
        p, n, N = self.params, self.model._n_points, self.split_factor
        scales, quats = self.model.get_scales(), self.model.get_quaternions()
        hot = grads.norm(dim=-1) >= p.densify_grad_threshold
        large = scales.max(dim=1).values > p.densification_percent_distinction * extent
        none = torch.zeros(n, dtype=torch.bool, device=self.device)
        cloned = none if p.disable_densify_clone else hot & ~large
        parted = none if p.disable_densify_split else hot & large
        rows = torch.arange(n, device=self.device)
        parents = rows[parted]
        # split samples: N draws per parent from its own Gaussian, all first draws, then all second draws (one generator call, as the
        # reference makes it, so that seeded runs agree)
        spread = scales[parted].repeat(N, 1)
        if samples is None:
            samples = torch.normal(mean=torch.zeros((spread.size(0), 3), device=self.device), std=spread)
        frames = quaternion_to_matrix(quats[parted]).repeat(N, 1, 1)
        moved = torch.bmm(frames, samples.unsqueeze(-1)).squeeze(-1)
        survivors, clones = rows[~parted], rows[cloned]
        src = torch.cat((survivors, clones, parents.repeat(N)))
        segment = torch.cat((torch.zeros_like(survivors), torch.ones_like(clones), torch.full_like(parents.repeat(N), 2)))
        offset = torch.zeros((src.numel(), 3), device=self.device)
        offset[segment == 2] = moved
        return RowPlan(src, segment == 0, segment == 2, offset)

    def _doomed(self, plan: RowPlan, extent: float, radii2d: torch.Tensor, grads: Optional[torch.Tensor]) -> torch.Tensor:
        """Rows of the planned set that pruning removes: nearly transparent, too large on screen, too large in the world (and, in the
        gradient-pruning phase, the ones with a large mean screen gradient).  NB an avatar without per-Gaussian opacity PARAMETERS
        (DreamWaltzG: its opacities come out of the MLP) fails in `get_opacities` exactly as the reference's accessor does on None --
        pruning needs --render.densify_disable_prune True for it, there as here."""
        p = self.params
        opacity = self.model.get_opacities().reshape(-1)[plan.src]
        stored = self._param("scales").data[plan.src]
        if bool(plan.sample.any()):
            stored = stored.clone()
            stored[plan.sample] = self._shrunk(stored[plan.sample])
        world = self.model.scale_activation(stored).max(dim=1).values
        out = (opacity < p.prune_opacity_threshold) | (radii2d > p.densify_screen_size_threshold) | (world > 0.1 * extent)
        if grads is not None:
            out = out | (grads.norm(dim=-1)[plan.src] >= p.densify_grad_threshold)
        return out

    def _density_update(self, extent: float, train_step: int, split_samples):
        n0 = self.model._n_points
        grads = torch.where(self.denom > 0, self.points_gradient_accum / self.denom, torch.zeros_like(self.denom))   # mean screen gradient; never seen: 0
        grow = not self.enable_grad_prune and not (self.params.disable_densify_clone and self.params.disable_densify_split)
        if grow:
            plan = self._plan_growth(grads, extent, split_samples)
            accum, denom = torch.zeros((plan.src.numel(), 1), device=self.device), torch.zeros((plan.src.numel(), 1), device=self.device)
            radii2d = torch.zeros((plan.src.numel(),), device=self.device)      # statistics restart whenever the set grew
        else:
            rows = torch.arange(n0, device=self.device)
            plan = RowPlan(rows, torch.ones_like(rows, dtype=torch.bool), torch.zeros_like(rows, dtype=torch.bool), torch.zeros((n0, 3), device=self.device))
            accum, denom, radii2d = self.points_gradient_accum, self.denom, self.max_radii2D
        grown = plan.src.numel()
        if not self.params.disable_prune:
            by_gradient = grads if self.enable_grad_prune else None
            if self.enable_grad_prune and train_step > self.densify_from_iter + (self.densify_until_iter - self.densify_from_iter) / 3:
                self.enable_grad_prune = False
            alive = ~self._doomed(plan, extent, radii2d, by_gradient)
            plan, accum, denom, radii2d = plan.take(alive), accum[alive], denom[alive], radii2d[alive]
        if grow or not self.params.disable_prune:
            self._commit(plan)
        self.points_gradient_accum, self.denom, self.max_radii2D = accum, denom, radii2d
        self.last_report = (n0, grown - n0, grown - self.model._n_points, self.model._n_points)

    def reset_opacity(self, value: float = 0.01):
        """Every opacity above `value` is pulled down to it; the tensor re-enters the optimizer with fresh moments."""
        p = self._param("opacities")
        assert p is not None
        capped = self.model.get_opacities().clamp(max=value)
        self._adopt(optim.resize_flat_params(self.optimizers, {p: (self.model.opacity_inverse_activation(capped), None, None)}))

    @torch.no_grad()
    def __call__(self, viewspace_points: torch.Tensor, radii: torch.Tensor, spatial_extent: float, train_step: int, split_samples=None):
        """One training step's share: statistics always, a density update at the interval, the opacity reset at its own.
        `split_samples` (test hook): the draws of the split, [N x parents, 3], in place of the generator's."""
        if train_step >= self.densify_until_iter:
            return
        self.update_densification_stats(viewspace_points, radii, radii > 0)
        due = train_step > self.densify_from_iter and train_step % self.densification_interval == 0
        if due:
            self._density_update(self.spatial_extent if spatial_extent is None else spatial_extent, train_step, split_samples)
        if not self.params.disable_reset and train_step % self.opacity_reset_interval == 0:
            self.reset_opacity()


def build_densifier(model, optimizers, cfg, optimizer_name: str = "avatar") -> GaussianDensifier:
    """From the render / optim sections of the training configuration; `optimizers`: the dict `model.get_optimizer(cfg)` returned."""
    r = cfg.render
    params = DensificationParams(max_iteration=cfg.optim.iters, densify_from_iter=r.densify_from_iter, densify_until_iter=r.densify_until_iter,
                                 densify_grad_threshold=r.densify_grad_threshold, disable_densify_clone=r.densify_disable_clone,
                                 disable_densify_split=r.densify_disable_split, disable_prune=r.densify_disable_prune,
                                 disable_reset=r.densify_disable_reset, enable_grad_prune=r.enable_grad_prune)
    return GaussianDensifier(model=model, params=params, optimizers=optimizers, optimizer_name=optimizer_name)
