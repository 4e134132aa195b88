"""Mirror of Scene (/root/reference/core/system/scene.py:15-168; SURVEY.md section 8a row R7, boundary B5): avatar_forward
(optional per-avatar scale / translation), multi-avatar merge, the debug overrides, GaussianRenderer.render, and the background
compositing `image + bg (1 - alpha)` for bg_mode in {black, white, gray}.  Learned / video / Gaussian backgrounds are outside the
hot path (scene.py:123-132,158-165) and are rejected."""
from typing import Iterable, Optional

import contextlib

import torch
import torch.nn as nn

from .avatar import DreamWaltzG, GaussianOutput, merge_gaussians
from .renderer import GaussianRenderer


class PureColorBackground:
    """core/system/background.py:14-57."""
    COLOR_VALUE_MAPPING = {'black': 0.0, 'white': 1.0, 'gray': 0.5}

    def __contains__(self, bg_mode):
        return bg_mode in ('black', 'white', 'gray')

    @staticmethod
    def get_background_like(color: str, image: torch.Tensor) -> torch.Tensor:
        return torch.full_like(image, PureColorBackground.COLOR_VALUE_MAPPING[color])


def downsample_gaussians(gaussians: GaussianOutput, n: int) -> GaussianOutput:
    """gaussian_utils.py downsample_gaussians: a random subset of n Gaussians (debug override use_fixed_n_gaussians)."""
    total = gaussians.positions.shape[0]
    if n >= total:
        return gaussians
    idx = torch.randperm(total, device=gaussians.positions.device)[:n]
    return GaussianOutput(**{k: (gaussians[k][idx] if torch.is_tensor(gaussians[k]) else None) for k in gaussians.keys()})


class Scene(nn.Module):
    def __init__(self, cfg, avatar, background=None, async_pair_count=False) -> None:
        super().__init__()
        if background is not None:
            raise NotImplementedError("learned / video / Gaussian backgrounds are outside the SDS hot path (scene.py:123-132,158-165)")
        self.device = torch.device(cfg.device)
        if isinstance(avatar, DreamWaltzG):
            self.avatar, self.avatars = avatar, None
        else:
            self.avatars = nn.ModuleList(avatar)
            self.avatar = self.avatars[0]
        self.background = None
        self.pure_colors = PureColorBackground()
        self.renderer = GaussianRenderer(sh_levels=cfg.render.sh_levels, bg_color=cfg.render.bg_color, async_pair_count=async_pair_count)
        r = cfg.render
        self.use_zero_scales = r.use_zero_scales
        self.use_constant_colors = r.use_constant_colors is not None
        self.constant_colors = None if r.use_constant_colors is None else torch.tensor([r.use_constant_colors], device=self.device)
        self.use_constant_opacities = r.use_constant_opacities is not None
        self.constant_opacities = None if r.use_constant_opacities is None else torch.tensor([r.use_constant_opacities], device=self.device)
        self.use_fixed_n_gaussians = r.use_fixed_n_gaussians is not None
        self.fixed_n_gaussians = None if r.use_fixed_n_gaussians is None else int(r.use_fixed_n_gaussians)
        self.avatar_transl = torch.tensor(eval(r.avatar_transl), device=self.device) if r.avatar_transl is not None else None
        self.avatar_scale = torch.tensor(eval(r.avatar_scale), device=self.device) if r.avatar_scale is not None else None

    def avatar_forward(self, smpl_observed_inputs: Optional[dict] = None, avatar=None, avatar_index: Optional[int] = None) -> GaussianOutput:
        if avatar is None:
            avatar = self.avatar
        gaussians = avatar.forward() if smpl_observed_inputs is None else avatar.animate(smpl_observed_inputs=smpl_observed_inputs)
        if self.avatar_scale is not None:
            s = self.avatar_scale
            if s.ndim == 1:
                s = s[avatar_index]
            gaussians.positions = gaussians.positions * s.unsqueeze(0)
            gaussians.scales = gaussians.scales * s.unsqueeze(0)
        if self.avatar_transl is not None:
            t = self.avatar_transl
            if t.ndim == 2:
                t = t[avatar_index]
            gaussians.positions = gaussians.positions + t.unsqueeze(0)
        return gaussians

    def forward_frames(self, data: dict, poses, bg_mode: Optional[str] = None, frozen_avatar: bool = False) -> dict:
        """Playback of F pose frames under one camera (`data`: the loader's dict) or each under its own (`data`: a list of F dicts, as the
        reference's evaluation loader yields them): `animate` per pose, then ONE rasterizer launch chain for all of them
        (renderer.render_frames).  Frame f equals `forward(data, poses[f], use_densifier=False, bg_mode=bg_mode)` bit for bit -- the
        reference's evaluation loop renders such sequences one pose at a time under inference mode (trainer.py:1019-1150).  Single avatar,
        no gradients.  `frozen_avatar`: the avatar's parameters do not change between the frames (a trained avatar playing a motion): the
        pose-independent part of `animate` -- canonical positions, grid encoding, colour / opacity network -- is computed once and kept until
        a parameter changes (avatar.DreamWaltzG.frozen_playback; same bits, 0.3 ms less per 300 k-Gaussian frame).
        -> {'image' | 'image_fg' | 'depth' | 'alpha': [F, H, W, C]}."""
        if self.avatars is not None:
            raise NotImplementedError("forward_frames renders one avatar per frame")
        frames = []
        self.avatar.frozen_playback = bool(frozen_avatar)
        try:
            for pose in poses:
                g = self.avatar_forward(smpl_observed_inputs=pose)
                if self.use_zero_scales:
                    g.scales = g.scales * 0.1
                if self.use_constant_colors:
                    g.colors = self.constant_colors.expand(g.colors.size(0), -1)
                if self.use_constant_opacities:
                    g.opacities = self.constant_opacities.expand(g.opacities.size(0), -1)
                if self.use_fixed_n_gaussians:
                    g = downsample_gaussians(g, self.fixed_n_gaussians)
                frames.append(g)
        finally:
            self.avatar.frozen_playback = False
        outputs = self.renderer.render_frames(data=data, frames=frames)
        if bg_mode in self.pure_colors:
            outputs['image_bg'] = self.pure_colors.get_background_like(bg_mode, outputs['image'])
            outputs['image_fg'] = outputs['image']
            outputs['image'] = outputs['image'] + outputs['image_bg'] * (1 - outputs['alpha'])
        else:
            outputs['image_fg'] = outputs['image']
        return outputs

    def forward_views(self, datas, poses, bg_mode: Optional[str] = None, streams=None) -> list:
        """The V views of a batched multi-view training step (trainer.train_forward_views): `animate` per view (`poses[v]`: that view's
        smpl inputs, or None for the canonical pose) -- each on its own stream when `streams` is given, they are independent chains of
        small launches -- then ONE differentiable rasterizer launch chain for all of them (renderer.render_frames: forward AND backward on
        (work, V) grids).  View v of the result equals `forward(datas[v], poses[v], use_densifier=False, bg_mode=bg_mode)` bit for bit,
        and so do the gradients it sends back.  Single avatar.  -> a list of V output dicts ([1, H, W, C] tensors)."""
        if self.avatars is not None:
            raise NotImplementedError("forward_views renders one avatar per view")
        main = torch.cuda.current_stream(self.avatar.device) if streams else None
        frames = []
        for v, pose in enumerate(poses):
            if streams:
                streams[v].wait_stream(main)
            with (torch.cuda.stream(streams[v]) if streams else contextlib.nullcontext()):
                g = self.avatar_forward(smpl_observed_inputs=pose)
                if self.use_zero_scales:
                    g.scales = g.scales * 0.1
                if self.use_constant_colors:
                    g.colors = self.constant_colors.expand(g.colors.size(0), -1)
                if self.use_constant_opacities:
                    g.opacities = self.constant_opacities.expand(g.opacities.size(0), -1)
                if self.use_fixed_n_gaussians:
                    g = downsample_gaussians(g, self.fixed_n_gaussians)
            if streams:
                for t in (g.positions, g.opacities, g.colors, g.sh_features, g.scales, g.quaternions):
                    if t is not None:
                        t.record_stream(main)
            frames.append(g)
        if streams:
            for side in streams[:len(frames)]:
                main.wait_stream(side)
        out = self.renderer.render_frames(data=list(datas), frames=frames)
        views = []
        for v in range(len(frames)):
            o = {k: t[v:v + 1] for k, t in out.items()}
            if bg_mode in self.pure_colors:
                o['image_bg'] = self.pure_colors.get_background_like(bg_mode, o['image'])
                o['image_fg'] = o['image']
                o['image'] = o['image'] + o['image_bg'] * (1 - o['alpha'])
            else:
                o['image_fg'] = o['image']
            views.append(o)
        return views

    def forward(self, data: dict, smpl_observed_inputs: Optional[dict] = None, use_densifier: bool = True, bg_mode: Optional[str] = None,
                **kwargs):
        if self.avatars is None:
            gaussians = self.avatar_forward(smpl_observed_inputs=smpl_observed_inputs)
        else:
            parts = []
            batch_size = smpl_observed_inputs['body_pose'].size(0)
            assert batch_size <= len(self.avatars), f'Assert num_smplx_inputs: {batch_size} <= num_avatars: {len(self.avatars)}'
            for i, avatar in enumerate(self.avatars):
                parts.append(self.avatar_forward({k: v[i:i + 1, ...] for k, v in smpl_observed_inputs.items()}, avatar=avatar, avatar_index=i))
            gaussians = merge_gaussians(*parts)
        if self.use_zero_scales:
            gaussians.scales = gaussians.scales * 0.1
        if self.use_constant_colors:
            gaussians.colors = self.constant_colors.expand(gaussians.colors.size(0), -1)
        if self.use_constant_opacities:
            gaussians.opacities = self.constant_opacities.expand(gaussians.opacities.size(0), -1)
        if self.use_fixed_n_gaussians:
            gaussians = downsample_gaussians(gaussians, self.fixed_n_gaussians)
        outputs = self.renderer.render(data=data, gaussians=gaussians, return_2d_radii=use_densifier)
        if bg_mode in self.pure_colors:
            outputs['image_bg'] = self.pure_colors.get_background_like(bg_mode, outputs['image'])
            outputs['image_fg'] = outputs['image']
            outputs['image'] = outputs['image'] + outputs['image_bg'] * (1 - outputs['alpha'])
        else:
            outputs['image_fg'] = outputs['image']
        return outputs

    def densify(self, densifiers: dict, render_outputs: dict, spatial_scale: float, train_step: int):
        """scene.py:170-186: the free Gaussians' share of the screen-space gradient / radii goes to the avatar's densifier."""
        if hasattr(self.avatar, 'densification_mask'):
            mask = self.avatar.densification_mask.to(render_outputs['radii'].device)
            viewspace_points = render_outputs['viewspace_points'][mask]
            viewspace_points.grad = render_outputs['viewspace_points'].grad[mask]
            radii = render_outputs['radii'][mask]
        else:
            viewspace_points = render_outputs['viewspace_points']
            radii = render_outputs['radii']
        densifiers['avatar'](viewspace_points=viewspace_points, radii=radii, spatial_extent=spatial_scale, train_step=train_step)

    # -- checkpoints (scene.py:170-208) ---------------------------------------------------------------------------------------
    @staticmethod
    def organize_state_dict(state_dict):
        by_module = {}
        for k, v in state_dict.items():
            module_name = k.split('.')[0]
            by_module.setdefault(module_name, {})[k.replace(f'{module_name}.', '', 1)] = v
        return by_module

    def state_dict(self, *args, **kwargs):
        """Scene.state_dict() of the reference: the avatar under `avatar.` (and `avatars.<i>.` for a multi-avatar scene)."""
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        """scene.py:196-200: resize the per-Gaussian parameters to the checkpoint's count (reset_by_state_dict), then a plain
        load; derived caches / topology of the native path are rebuilt afterwards."""
        by_module = self.organize_state_dict(state_dict)
        if 'avatar' in by_module:
            self.avatar.reset_by_state_dict(by_module['avatar'])
        own = set(super().state_dict().keys())
        filtered = {k: v for k, v in state_dict.items() if k in own}
        res = super().load_state_dict(filtered, strict=False)
        if 'avatar.nerf_bound' in state_dict:
            self.avatar._nerf_bound_host = float(state_dict['avatar.nerf_bound'])
        for a in ([self.avatar] if self.avatars is None else list(self.avatars)):
            a.invalidate_caches()
        unexpected = [k for k in state_dict.keys() if k not in own]
        if strict and (res.missing_keys or unexpected):
            raise RuntimeError("Scene.load_state_dict: missing %s unexpected %s" % (res.missing_keys, unexpected))
        return res
