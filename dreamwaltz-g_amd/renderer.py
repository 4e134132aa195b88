"""Mirror of GaussianRenderer (/root/reference/core/gaussian/gaussian_renderer.py:9-224):
same constructor, same `build_gaussian_rasterizer(data)` / `compute_colors` / `compute_3d_covariance` /
`render(data, gaussians, return_2d_radii, rasterizer)` contract and output dict, bound to the HIP rasterizer (rasterizer.py)
instead of the CUDA package."""
import ctypes
import os
from typing import Optional

import torch

from . import _lib
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, PairCapacity, morton_order
from .rigid import mm4, quaternion_to_matrix

# real spherical harmonics, degree <= 3 (core/gaussian/spherical_harmonics.py:5-40,117-172)
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
      -0.5900435899266435]


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [..., C, (deg+1)^2], dirs [..., 3] unit -> [..., C] (degree <= 3: what the rasterizer supports; the Python-side colour
    path exists for `compute_color_in_rasterizer=False` and the sh_levels=1 background of scene.py:123-132)."""
    assert 0 <= deg <= 3, "SH degree 0..3"
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] +
                      C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10] +
                          C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
                          C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14] +
                          C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def get_colors(sh_features, directions, sh_levels):
    """core/gaussian/gaussian_utils.py:12-17."""
    sh_features = sh_features[:, :sh_levels ** 2]
    shs_view = sh_features.transpose(-1, -2).reshape(-1, 3, sh_levels ** 2)
    return torch.clamp_min(eval_sh(sh_levels - 1, shs_view, directions) + 0.5, 0.0).view(-1, 3)


def stream_keyed(renderer) -> bool:
    return bool(getattr(renderer, "per_stream_pair_states", False))


class _ChainCounts:
    """`last_num_pairs` of a frames chain: (largest pair count after exact culling, the views' mean reference tile-pair count)."""

    def __init__(self, pair_state):
        self.pair_state, self.visit_order = pair_state, None

    @property
    def last_num_pairs(self):
        self.pair_state.resolve()
        return self.pair_state.last_num_pairs, self.pair_state.last_num_pairs_ref


class GaussianRenderer:
    def __init__(self, sh_levels=4, bg_color=(0.0, 0.0, 0.0), compute_color_in_rasterizer=True,
                 compute_covariance_in_rasterizer=True, async_pair_count=False, reorder_every: Optional[int] = None) -> None:
        self.sh_levels = sh_levels
        self.bg_color = torch.tensor(bg_color, dtype=torch.float32)
        self.compute_color_in_rasterizer = compute_color_in_rasterizer
        self.compute_covariance_in_rasterizer = compute_covariance_in_rasterizer
        # opt-in: size the pair workspace without a host synchronisation per frame (rasterizer.PairCapacity); state lives HERE,
        # per renderer and per (device, H, W)
        self.async_pair_count = async_pair_count
        self._pair_states = {}
        self._bg_dev = {}
        self.last_rasterizer = None
        # Binning order (rasterizer.morton_order): refreshed from the current positions every `reorder_every` frames per Gaussian
        # count (0, the default since round 5: never -- index order, coalesced reads; the supertile histograms merge their atomics either way);
        # the images do not depend on it.  DWG_RASTER_REORDER=<frames> switches the refresh on.
        if reorder_every is None:
            reorder_every = int(os.environ.get("DWG_RASTER_REORDER", "0"))
        self.reorder_every = int(reorder_every)
        self._visit_orders = {}

    # -- capacity bookkeeping of the async mode ------------------------------------------------------------------------
    def pair_state_key(self, device, H, W):
        """The key of the PairCapacity a frame rendered NOW (this device, size and current stream) uses: one helper for `pair_state` and for
        whoever installs a state of its own (player.GraphedAnimation's frozen capacity) -- the two cannot disagree about the key's shape."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:          # "cuda" and "cuda:<current>" are one device, one state
            device = torch.device("cuda", torch.cuda.current_device())
        # one state per stream as well once a multi-view step has rendered its views concurrently on side streams: a state's pinned count /
        # overflow words belong to ONE in-flight frame
        if not stream_keyed(self):
            return (str(device), int(H), int(W))
        stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
        return (str(device), int(H), int(W), stream)

    def pair_state(self, device, H, W) -> Optional[PairCapacity]:
        if not self.async_pair_count:
            return None
        key = self.pair_state_key(device, H, W)
        if key not in self._pair_states:
            self._pair_states[key] = PairCapacity()
        return self._pair_states[key]

    def consume_overflow(self) -> bool:
        """True if a frame since the last call was truncated by the pair capacity (to be re-rendered by the caller)."""
        return any([st.consume_overflow() for st in self._pair_states.values()])

    def build_gaussian_rasterizer(self, data: dict, **kwargs) -> GaussianRasterizer:
        """gaussian_renderer.py:23-70.  viewmatrix = extrinsic[0]^T, projmatrix = viewmatrix @ projection[0]^T, campos =
        c2w[0,:3,3]; on the GPU the three come out of one small launch (no library GEMM for a 4x4 product)."""
        world_view_matrix = data['extrinsic'][0]
        projection_matrix = data['projection'][0]
        image_width, image_height = data['image_width'], data['image_height']
        tanfovy = data['tanfov'][0].item()
        tanfovx = data['tanfov_x'][0].item() if 'tanfov_x' in data else tanfovy
        device = world_view_matrix.device
        tanfov_dev = None
        if device.type == 'cuda' and data.get('tanfov_dev') is not None:
            # A camera that lives in DEVICE memory (data['tanfov_dev'] = {tanfovx, tanfovy} next to the matrices): the field of view reaches
            # the kernels through a pointer, not through kernel arguments -- a step captured into a graph follows whatever camera its static
            # tensors hold at replay time (step_graph.GraphedTrainStep.step(pose, camera)); the two host scalars above are then only the
            # settings object's record of the camera it was built with.
            out = torch.empty(37, device=device, dtype=torch.float32)
            tf = data['tanfov_dev']
            p = _lib.ptr
            _lib.check(_lib.lib().dwg_raster_camera_block(p(world_view_matrix.float().contiguous()), p(projection_matrix.float().contiguous()),
                                                          p(data['c2w'][0].float().contiguous()), ctypes.c_void_p(tf.data_ptr() + 4), p(tf), p(out),
                                                          ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                       "dwg_raster_camera_block")
            viewmatrix, projmatrix, campos, tanfov_dev = out[:16].view(4, 4), out[16:32].view(4, 4), out[32:35], out[35:37]
        elif device.type == 'cuda':
            out = torch.empty(35, device=device, dtype=torch.float32)
            p = _lib.ptr
            _lib.check(_lib.lib().dwg_raster_camera_setup(p(world_view_matrix.float().contiguous()), p(projection_matrix.float().contiguous()),
                                                          p(data['c2w'][0].float().contiguous()), p(out),
                                                          ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                       "dwg_raster_camera_setup")
            viewmatrix, projmatrix, campos = out[:16].view(4, 4), out[16:32].view(4, 4), out[32:35]
        else:
            viewmatrix = world_view_matrix.transpose(0, 1)
            projmatrix = mm4(viewmatrix, projection_matrix.transpose(0, 1))
            campos = data['c2w'][0, :3, 3]
        if device not in self._bg_dev:
            self._bg_dev[device] = self.bg_color.to(device)
        raster_settings = {
            "image_height": image_height, "image_width": image_width, "tanfovx": tanfovx, "tanfovy": tanfovy,
            "bg": self._bg_dev[device], "viewmatrix": viewmatrix, "projmatrix": projmatrix, "sh_degree": self.sh_levels - 1,
            "campos": campos, "tanfov_dev": tanfov_dev,
        }
        raster_settings.update(kwargs)
        for key, value in raster_settings.items():
            if isinstance(value, torch.Tensor):
                raster_settings[key] = value.to(device)
        settings = GaussianRasterizationSettings(**raster_settings, scale_modifier=1., prefiltered=False, debug=False)
        return GaussianRasterizer(raster_settings=settings, pair_state=self.pair_state(device, image_height, image_width))

    def _visit_order_for(self, positions: torch.Tensor) -> torch.Tensor:
        key = (str(positions.device), int(positions.shape[0]))
        entry = self._visit_orders.get(key)
        if entry is None or entry[1] >= self.reorder_every:
            order = morton_order(positions)
            ev = None
            if positions.is_cuda:           # frames on OTHER streams (concurrent views of a multi-view step) wait for the permutation to be complete
                ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(positions.device))
            entry = self._visit_orders[key] = [order, 0, ev, torch.cuda.current_stream(positions.device).cuda_stream if positions.is_cuda else 0]
        elif entry[2] is not None and torch.cuda.current_stream(positions.device).cuda_stream != entry[3]:
            torch.cuda.current_stream(positions.device).wait_event(entry[2])
        entry[1] += 1
        return entry[0]

    def compute_colors(self, sh_features, directions=None, positions=None, camera_positions=None, sh_levels=None, sh_rotations=None):
        """gaussian_renderer.py:72-105."""
        if directions is None:
            if positions is None or camera_positions is None:
                raise ValueError("Either directions or positions must be provided.")
            directions = torch.nn.functional.normalize(positions - camera_positions, dim=-1)
        if sh_rotations is not None:
            directions = (directions.unsqueeze(1) @ sh_rotations)[..., 0, :]
        if sh_levels is None:
            sh_levels = self.sh_levels
        return get_colors(sh_features=sh_features, directions=directions, sh_levels=sh_levels)

    def render_frames(self, data: dict, frames) -> dict:
        """F posed sets of Gaussians seen by ONE camera (or each by its own), rasterized by ONE launch chain (rasterizer.rasterize_frames):
        the playback path under inference mode (trainer.py:1019-1150 renders a pose sequence frame by frame), and -- differentiable since
        round 6, the backward on (work, F) grids too -- the V views of a batched multi-view training step.  Frame f of the result is
        bit-identical to `render(data, frames[f])`; what changes is the cost: the rasterizer's seven dependent launches are paid once per
        batch and each runs over F times the work (per frame, 300 k Gaussians at 1024^2: 0.44 ms alone, 0.25 ms at F = 4).
        -> {'image': [F, H, W, 3], 'depth': [F, H, W, 1], 'alpha': [F, H, W, 1]}."""
        from .rasterizer import rasterize_frames
        # one camera for the batch (`data`: the loader's dict), or one per frame (`data`: a list of F such dicts -- the reference's evaluation
        # loader hands a camera with every pose; image size and field of view must agree across the batch)
        datas = list(data) if isinstance(data, (list, tuple)) else [data]
        if len(datas) not in (1, len(frames)):
            raise ValueError("render_frames: %d cameras for %d frames" % (len(datas), len(frames)))
        rows, rs = [], None
        for d in datas:
            r = self.build_gaussian_rasterizer(data=d).raster_settings
            if rs is not None and (r.image_height, r.image_width, r.tanfovx, r.tanfovy) != (rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy):
                raise ValueError("render_frames: the frames of a batch share image size and field of view")
            rs = rs or r
            rows.append(torch.cat([r.viewmatrix.reshape(-1), r.projmatrix.reshape(-1), r.campos.reshape(-1)]).float())
        cam = rows[0] if len(rows) == 1 else torch.stack(rows)
        data = datas[0]
        g0 = frames[0]
        use_colors = g0.colors is not None
        if not use_colors and not self.compute_color_in_rasterizer:
            for f, g in enumerate(frames):
                g.colors = self.compute_colors(sh_features=g.sh_features, positions=g.positions,
                                               camera_positions=datas[f if len(datas) > 1 else 0]['c2w'][:, :3, 3])
            use_colors = True
        use_cov = not self.compute_covariance_in_rasterizer
        if use_cov:
            for g in frames:
                g.cov3D = self.compute_3d_covariance(scales=g.scales, quaternions=g.quaternions)
        st = lambda name: torch.stack([getattr(g, name) for g in frames])          # noqa: E731
        # the pair workspace is sized exactly: ONE read-back of the F pair counts between binning and compositing (per frame before)
        color, radii, depth, alpha, info = rasterize_frames(
            st("positions"), st("opacities"), colors_precomp=st("colors") if use_colors else None,
            shs=None if use_colors else st("sh_features"), scales=None if use_cov else st("scales"),
            rotations=None if use_cov else st("quaternions"), cov3D_precomp=st("cov3D") if use_cov else None,
            cameras=cam, image_height=rs.image_height, image_width=rs.image_width, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
            bg=rs.bg, sh_degree=rs.sh_degree,
            # a training chain (gradients wanted) sizes its pair buffers like the single-frame training path: from the renderer's running state,
            # without a host wait; playback keeps the exact per-chain read-back
            pair_state=self.pair_state(g0.positions.device, rs.image_height, rs.image_width) if torch.is_grad_enabled() and g0.positions.requires_grad else None)
        ps = self.pair_state(g0.positions.device, rs.image_height, rs.image_width) if torch.is_grad_enabled() and g0.positions.requires_grad else None
        if ps is not None:
            self.last_rasterizer = _ChainCounts(ps)             # what `last_rasterizer.last_num_pairs` reports after a training chain
        self.last_frames_headers = info["headers"]                  # device [F, 4]: block pairs, overflow, reference tile pairs, segments
        return {"image": color.permute(0, 2, 3, 1), "depth": depth.permute(0, 2, 3, 1), "alpha": alpha.permute(0, 2, 3, 1)}

    @staticmethod
    def compute_3d_covariance(scales, quaternions):
        """gaussian_renderer.py:107-128: R diag(s^2) R^T, upper triangle (xx, xy, xz, yy, yz, zz)."""
        R = quaternion_to_matrix(quaternions)
        M = R * (scales * scales).unsqueeze(-2)             # R diag(s^2)
        cov = mm4(M, R.transpose(-1, -2))
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1).float()

    def render(self, data: dict, gaussians, return_2d_radii: bool = False,
               rasterizer: Optional[GaussianRasterizer] = None) -> dict:
        """gaussian_renderer.py:130-224 (mutates `gaussians` like the reference: checklist Q9)."""
        if rasterizer is None:
            rasterizer = self.build_gaussian_rasterizer(data=data)
        self.last_rasterizer = rasterizer
        if gaussians.colors is not None:
            gaussians.sh_features = None
        elif not self.compute_color_in_rasterizer:
            gaussians.colors = self.compute_colors(sh_features=gaussians.sh_features, positions=gaussians.positions,
                                                   camera_positions=data['c2w'][:, :3, 3])
            gaussians.sh_features = None
        if not self.compute_covariance_in_rasterizer:
            gaussians.cov3D = self.compute_3d_covariance(scales=gaussians.scales, quaternions=gaussians.quaternions)
            gaussians.quaternions = None
            gaussians.scales = None
        means3D = gaussians.positions
        if rasterizer.visit_order is None and self.reorder_every > 0 and means3D.is_cuda and means3D.shape[0] >= 4096:
            rasterizer.visit_order = self._visit_order_for(means3D)
        screenspace_points = torch.zeros(means3D.shape[0], 3, dtype=means3D.dtype, requires_grad=True, device=means3D.device)
        if return_2d_radii:
            try:
                screenspace_points.retain_grad()
            except Exception:
                print("WARNING: return_2d_radii is True, but failed to retain grad of screenspace_points!")
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
