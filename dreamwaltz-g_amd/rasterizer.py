"""Drop-in for the `diff_gaussian_rasterization` Python module (boundary B1).

Mirrors the interface the reference uses at /root/reference/core/gaussian/gaussian_renderer.py:
  :5      from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
  :60-70  GaussianRasterizationSettings(**kwargs, scale_modifier=1., prefiltered=False, debug=False)
  :186-195 rasterizer(means3D=, means2D=, shs=, colors_precomp=, opacities=, scales=, rotations=, cov3D_precomp=)
           -> (image[3,H,W], radii[G], depth[1,H,W], alpha[1,H,W])
All arithmetic runs in the hand-written HIP kernels of csrc/raster.hip through the C-ABI in
include/dwg_raster.h; there is no CPU fallback.
"""
import ctypes
from typing import NamedTuple, Optional

import torch

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    tanfov_dev: Optional[torch.Tensor] = None      # device float32 [2] = {tanfovx, tanfovy}: read by the kernels INSTEAD of the two scalars (a camera
                                                   # that lives in device memory: include/dwg_raster.h dwg_raster_settings::tanfov)


class PairCapacity:
    """Sizing of the (Gaussian, block) pair workspace WITHOUT a per-frame host synchronisation (opt-in; the default path reads
    the 4-byte pair count back like the reference's extension does).  One object per renderer and (device, H, W) -- no module
    state.  The pair buffers of frame t are sized from the counts seen so far (with head-room; the workspace only costs memory,
    288 GB of HBM make a generous bound free); frame t's real count and overflow flag come back through a pinned 16-byte copy
    that `resolve()` waits on -- called from the backward of the same frame (by then the GPU is far past the forward, so
    the wait is free) or at the next forward.  An overflow (stage B found fewer slots than pairs: the frame was truncated) is
    LATCHED in `overflow`; the owner of the training step re-renders that frame before any optimizer step (`consume_overflow`)."""

    def __init__(self, min_pairs=1 << 20, headroom=4.0):
        self.cap = 0
        self.min_pairs, self.headroom = int(min_pairs), float(headroom)
        self.host = None
        self.event = None
        self.pending = False
        self.overflow = False
        self.frozen = False              # True: fixed capacity, no events, no host waits (a frame captured into a graph, see player.py)
        self.truncated, self.truncated_host = None, None      # frozen states: device count of truncated frames + its pinned mirror
        self.seq = 0                     # frames rendered through this state; `pending_seq` = the frame whose count is in flight
        self.pending_seq = 0
        self.overflow_seqs = set()       # EVERY recent frame that was truncated (several may be in flight through one state before their
                                         # backwards run): each one's backward returns zeros (see _RasterizeGaussians.backward)
        self.last_num_pairs = 0          # pairs after exact culling (what the buffers hold)
        self.last_num_pairs_ref = 0      # sum of 16x16 reference tiles touched (the K of SURVEY 8d's byte formula)

    def seed(self, K):
        self.cap = max(self.cap, int(K * self.headroom), self.min_pairs)

    def resolve(self):
        if not self.pending or self.frozen:
            return
        self.event.synchronize()
        self.pending = False
        K, ovf, Kref = int(self.host[0]), int(self.host[1]), int(self.host[2])
        self.last_num_pairs, self.last_num_pairs_ref = K, Kref
        if ovf:
            self.overflow = True
            self.overflow_seqs.add(self.pending_seq)
            if len(self.overflow_seqs) > 64:                 # bounded: frames this old have had their backward or never will
                self.overflow_seqs = {q for q in self.overflow_seqs if q > self.pending_seq - 64}
        if ovf or K * 2 > self.cap:
            self.cap = max(self.cap, int(K * self.headroom))

    def consume_overflow(self):
        self.resolve()
        o, self.overflow = self.overflow, False
        return o


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.float().contiguous()


def _settings_struct(rs: GaussianRasterizationSettings, device, sh_coeffs: int, keep: list):
    def dev(t):
        t = torch.as_tensor(t, dtype=torch.float32, device=device).contiguous()
        keep.append(t)
        return t.data_ptr()
    c = _lib.RasterSettingsC()
    c.image_height = int(rs.image_height); c.image_width = int(rs.image_width)
    c.tanfovx = float(rs.tanfovx); c.tanfovy = float(rs.tanfovy)
    c.scale_modifier = float(rs.scale_modifier)
    c.sh_degree = int(rs.sh_degree); c.sh_coeffs = int(sh_coeffs)
    c.prefiltered = int(bool(rs.prefiltered)); c.debug = int(bool(rs.debug))
    c.bg = dev(rs.bg); c.viewmatrix = dev(rs.viewmatrix); c.projmatrix = dev(rs.projmatrix)
    c.campos = dev(rs.campos)
    tfd = getattr(rs, "tanfov_dev", None)
    if tfd is not None:
        if not (tfd.is_cuda and tfd.dtype == torch.float32 and tfd.numel() >= 2 and tfd.is_contiguous()):
            raise ValueError("tanfov_dev: a contiguous float32 CUDA tensor {tanfovx, tanfovy}")
        keep.append(tfd)
        c.tanfov = tfd.data_ptr()
    return c


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings: GaussianRasterizationSettings, pair_state: Optional[PairCapacity], info: dict,
                visit_order: Optional[torch.Tensor] = None):
        if not means3D.is_cuda:
            raise RuntimeError("dreamwaltz_g_amd rasterizer runs on the GPU only (HIP kernels); got a CPU tensor")
        L = _lib.lib()
        device = means3D.device
        G = int(means3D.shape[0])
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        means3D = _f32c(means3D)
        sh = _f32c(sh) if sh is not None and sh.numel() > 0 else None
        colors_precomp = _f32c(colors_precomp) if colors_precomp is not None and colors_precomp.numel() > 0 else None
        if G == 0:
            sh, colors_precomp = None, torch.zeros(0, 3, device=device)
        opac = _f32c(opacities).reshape(-1)
        scales = _f32c(scales) if scales is not None and scales.numel() > 0 else None
        rotations = _f32c(rotations) if rotations is not None and rotations.numel() > 0 else None
        cov3D = _f32c(cov3Ds_precomp) if cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0 else None
        if G == 0 and cov3D is None:
            scales = torch.zeros(0, 3, device=device); rotations = torch.zeros(0, 4, device=device)
        keep = []
        M = int(sh.shape[1]) if sh is not None else 0
        cfg = _settings_struct(raster_settings, device, M, keep)
        if visit_order is not None:         # binning walks the Gaussians in this order (a permutation; the images do not depend on it)
            if visit_order.dtype != torch.int32 or visit_order.device != device or visit_order.numel() != G or not visit_order.is_contiguous():
                raise ValueError("visit_order: a contiguous int32 permutation of the %d Gaussians on %s" % (G, device))
            cfg.visit_order = visit_order.data_ptr()
        gb, pb, ib = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(L.dwg_raster_workspace_sizes(G, H, W, 0, ctypes.byref(gb), ctypes.byref(pb), ctypes.byref(ib)),
                   "dwg_raster_workspace_sizes")
        ws_geom = torch.empty(gb.value, dtype=torch.uint8, device=device)
        ws_image = torch.empty(ib.value, dtype=torch.uint8, device=device)
        radii = torch.empty(G, dtype=torch.int32, device=device)         # k_preprocess writes every entry (0 for what it culls)
        st = _stream(device)
        p = _lib.ptr
        if pair_state is not None:
            pair_state.resolve()        # the previous frame's count (long complete): keeps `cap` current, latches an overflow
        _lib.check(L.dwg_raster_forward_bin(ctypes.byref(cfg), G, p(means3D), p(sh), p(colors_precomp), p(opac),
                                            p(scales), p(rotations), p(cov3D), p(radii), p(ws_geom), st),
                   "dwg_raster_forward_bin")
        if pair_state is not None and pair_state.cap > 0:
            cap = pair_state.cap        # no host synchronisation
        else:
            # one 16-byte read-back sizes the pair buffers exactly (the reference's extension does the same D2H copy)
            hdr = ws_geom[:16].view(torch.int32).cpu()
            K = int(hdr[0])
            info["num_pairs"], info["num_pairs_ref"] = K, int(hdr[2])
            cap = max(K, 1)
            if pair_state is not None:
                pair_state.seed(K)
                pair_state.last_num_pairs, pair_state.last_num_pairs_ref = K, int(hdr[2])
                cap = pair_state.cap
        _lib.check(L.dwg_raster_workspace_sizes(G, H, W, cap, ctypes.byref(gb), ctypes.byref(pb), ctypes.byref(ib)),
                   "dwg_raster_workspace_sizes")
        ws_pairs = torch.empty(pb.value, dtype=torch.uint8, device=device)
        color = torch.empty(3, H, W, dtype=torch.float32, device=device)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=device)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=device)
        _lib.check(L.dwg_raster_forward_render(ctypes.byref(cfg), G, p(ws_geom), p(ws_pairs), cap, p(ws_image),
                                               p(color), p(depth), p(alpha), st), "dwg_raster_forward_render")
        if pair_state is not None:
            if pair_state.host is None:
                pair_state.host = torch.zeros(4, dtype=torch.int32).pin_memory()
            pair_state.host.copy_(ws_geom[:16].view(torch.int32), non_blocking=True)
            if pair_state.frozen:
                # frames of a captured graph: nobody waits per frame, so truncations are COUNTED on the device and the count is what the
                # owner's check() reads (the per-frame words above only describe the last frame)
                if pair_state.truncated is None:
                    pair_state.truncated = torch.zeros(1, dtype=torch.int32, device=device)
                    pair_state.truncated_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                pair_state.truncated.add_(ws_geom[:16].view(torch.int32)[1:2])
                pair_state.truncated_host.copy_(pair_state.truncated, non_blocking=True)
            pair_state.seq += 1
            if not pair_state.frozen:
                ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(device))
                pair_state.event, pair_state.pending, pair_state.pending_seq = ev, True, pair_state.seq
        ctx.raster_settings = raster_settings
        ctx.set_materialize_grads(False)        # an output the loss never touched (depth / alpha in the SDS path) arrives as None, not as zeros
        ctx.cap = cap
        ctx.pair_state = pair_state
        ctx.frame_seq = pair_state.seq if pair_state is not None else 0
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3D is not None)
        ctx.save_for_backward(means3D, sh, colors_precomp, opac, scales, rotations, cov3D, ws_geom, ws_pairs, ws_image)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        L = _lib.lib()
        means3D, sh, colors_precomp, opac, scales, rotations, cov3D, ws_geom, ws_pairs, ws_image = ctx.saved_tensors
        rs = ctx.raster_settings
        device = means3D.device
        G = int(means3D.shape[0])
        if ctx.pair_state is not None:
            ctx.pair_state.resolve()     # this frame's forward finished long ago: free, and latches a capacity overflow
            if ctx.frame_seq in ctx.pair_state.overflow_seqs:
                # THIS frame was truncated by the pair capacity: its owner renders it again (SDSTrainer.train_step).  Its gradient is
                # ZERO, not the truncated frame's partial one -- a multi-view step accumulates several frames into one gradient buffer
                # before the optimizer runs, and the re-rendered frame must be the only contribution of its view.
                z = lambda t: None if t is None else torch.zeros_like(t)     # noqa: E731
                return (z(means3D), torch.zeros(G, 3, device=device), z(sh), z(colors_precomp), torch.zeros(G, 1, device=device),
                        z(scales), z(rotations), z(cov3D), None, None, None, None)
        keep = []
        M = int(sh.shape[1]) if sh is not None else 0
        cfg = _settings_struct(rs, device, M, keep)
        H, W = int(rs.image_height), int(rs.image_width)
        g_color = torch.zeros(3, H, W, device=device) if g_color is None else _f32c(g_color)
        g_depth = None if g_depth is None else _f32c(g_depth)
        g_alpha = None if g_alpha is None else _f32c(g_alpha)
        d_means3D = torch.empty(G, 3, device=device)
        d_means2D = torch.empty(G, 3, device=device)
        d_opac = torch.empty(G, 1, device=device)
        d_sh = torch.empty_like(sh) if sh is not None else None
        d_colors = torch.empty(G, 3, device=device) if colors_precomp is not None else None
        d_scales = torch.empty(G, 3, device=device) if scales is not None else None
        d_rots = torch.empty(G, 4, device=device) if rotations is not None else None
        d_cov = torch.empty(G, 6, device=device) if cov3D is not None else None
        ws_grad = torch.empty(max(G, 1) * 12, dtype=torch.float32, device=device)
        p = _lib.ptr
        _lib.check(L.dwg_raster_backward(ctypes.byref(cfg), G, p(means3D), p(sh), p(colors_precomp), p(opac), p(scales),
                                         p(rotations), p(cov3D), p(ws_geom), p(ws_pairs), ctx.cap, p(ws_image),
                                         p(ws_grad), p(g_color), p(g_depth), p(g_alpha), p(d_means3D), p(d_means2D),
                                         p(d_sh), p(d_colors), p(d_opac), p(d_scales), p(d_rots), p(d_cov),
                                         _stream(device)), "dwg_raster_backward")
        return d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rots, d_cov, None, None, None, None


class _RasterizeFrames(torch.autograd.Function):
    """F frames per launch chain, forward AND backward (include/dwg_raster.h dwg_raster_*_frames): the (work, F) grids of the seven forward
    and three backward launches.  Inputs [F, G, ...] (a posed set of Gaussians per frame) or [G, ...] (shared: their gradients are the sum
    over the frames, added in frame order); frame f's images and gradient rows are bit for bit those of the single-frame call."""

    @staticmethod
    def forward(ctx, means3D, opac, colors_precomp, shs, scales, rotations, cov3D, cameras, H, W, tanfovx, tanfovy, bg, sh_degree,
                scale_modifier, pair_capacity, info, pair_state=None):
        L = _lib.lib()
        device = means3D.device
        per_frame = means3D.dim() == 3
        F = int(means3D.shape[0]) if per_frame else (int(cameras.shape[0]) if cameras.dim() == 2 else 1)
        G = int(means3D.shape[-2])
        M = int(shs.shape[-2]) if shs is not None else 0
        keep = []
        rs = GaussianRasterizationSettings(H, W, float(tanfovx), float(tanfovy), bg, float(scale_modifier), cameras.reshape(-1)[0:16],
                                           cameras.reshape(-1)[16:32], int(sh_degree), cameras.reshape(-1)[32:35], False, False)
        cfg = _settings_struct(rs, device, M, keep)
        fr = _lib.RasterFramesC(F, G if per_frame else 0, 35 if (cameras.dim() == 2 and cameras.shape[0] == F and F > 1) else 0)
        gb, pb, ib = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(L.dwg_raster_workspace_sizes(G, H, W, 0, ctypes.byref(gb), ctypes.byref(pb), ctypes.byref(ib)), "dwg_raster_workspace_sizes")
        ws_geom = torch.empty(F * gb.value, dtype=torch.uint8, device=device)
        ws_image = torch.empty(F * ib.value, dtype=torch.uint8, device=device)
        radii = torch.empty(F, G, dtype=torch.int32, device=device)          # k_preprocess writes every entry (0 for what it culls)
        st, p = _stream(device), _lib.ptr
        _lib.check(L.dwg_raster_forward_bin_frames(ctypes.byref(cfg), ctypes.byref(fr), G, p(means3D), p(shs), p(colors_precomp), p(opac),
                                                   p(scales), p(rotations), p(cov3D), p(radii), p(ws_geom), st), "dwg_raster_forward_bin_frames")
        if pair_state is not None:
            pair_state.resolve()               # the previous chain's counts (long complete): keeps `cap` current, latches an overflow
        if pair_capacity is not None:
            cap = int(pair_capacity)
        elif pair_state is not None and pair_state.cap > 0:
            cap = pair_state.cap               # no host synchronisation (PairCapacity): every frame of the chain gets the state's capacity
        else:                                  # one read-back of the F pair counts sizes the shared capacity exactly
            hdrs = ws_geom.view(F, gb.value)[:, :16].contiguous().view(torch.int32)
            cap = max(int(hdrs[:, 0].max().item()), 1)
            if pair_state is not None:
                pair_state.seed(cap)
                cap = pair_state.cap
        _lib.check(L.dwg_raster_workspace_sizes(G, H, W, cap, ctypes.byref(gb), ctypes.byref(pb), ctypes.byref(ib)), "dwg_raster_workspace_sizes")
        ws_pairs = torch.empty(F * pb.value, dtype=torch.uint8, device=device)
        color = torch.empty(F, 3, H, W, dtype=torch.float32, device=device)
        depth = torch.empty(F, 1, H, W, dtype=torch.float32, device=device)
        alpha = torch.empty(F, 1, H, W, dtype=torch.float32, device=device)
        _lib.check(L.dwg_raster_forward_render_frames(ctypes.byref(cfg), ctypes.byref(fr), G, p(ws_geom), p(ws_pairs), cap, p(ws_image),
                                                      p(color), p(depth), p(alpha), st), "dwg_raster_forward_render_frames")
        info["headers"] = ws_geom.view(F, -1)[:, :16].contiguous().view(torch.int32)     # device tensor: [F, 4] = K, overflow, K_ref, segments
        info["capacity"] = cap
        if pair_state is not None and not pair_state.frozen:
            # the chain's largest pair count and whether ANY frame was truncated travel to the host behind the launches (pinned 16 bytes)
            if pair_state.host is None:
                pair_state.host = torch.zeros(4, dtype=torch.int32).pin_memory()
            hd = info["headers"]
            pair_state.host.copy_(torch.stack((hd[:, 0].max(), hd[:, 1].max(), hd[:, 2].sum() // F, hd[:, 3].max())), non_blocking=True)     # [2]: the views' mean K_ref (what the byte formula of a view uses)
            pair_state.seq += 1
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(device))
            pair_state.event, pair_state.pending, pair_state.pending_seq = ev, True, pair_state.seq
        ctx.pair_state = pair_state
        ctx.frame_seq = pair_state.seq if pair_state is not None else 0
        ctx.geo = (F, G, H, W, M, per_frame, cap, float(tanfovx), float(tanfovy), int(sh_degree), float(scale_modifier))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(means3D, opac, colors_precomp, shs, scales, rotations, cov3D, cameras, bg if torch.is_tensor(bg) else None,
                              ws_geom, ws_pairs, ws_image)
        ctx.bg = bg
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        L = _lib.lib()
        means3D, opac, colors_precomp, shs, scales, rotations, cov3D, cameras, bg_t, ws_geom, ws_pairs, ws_image = ctx.saved_tensors
        F, G, H, W, M, per_frame, cap, tanfovx, tanfovy, sh_degree, scale_modifier = ctx.geo
        device = means3D.device
        if ctx.pair_state is not None:
            ctx.pair_state.resolve()
            if ctx.frame_seq in ctx.pair_state.overflow_seqs:
                # a frame of this chain was truncated by the pair capacity: its owner renders the step's views again (SDSTrainer.train_step);
                # the chain contributes ZEROS, as a truncated single frame does
                z = lambda t: None if t is None else torch.zeros_like(t)     # noqa: E731
                return (z(means3D), z(opac), z(colors_precomp), z(shs), z(scales), z(rotations), z(cov3D)) + (None,) * 11
        keep = []
        rs = GaussianRasterizationSettings(H, W, tanfovx, tanfovy, bg_t if bg_t is not None else ctx.bg, scale_modifier, cameras.reshape(-1)[0:16],
                                           cameras.reshape(-1)[16:32], sh_degree, cameras.reshape(-1)[32:35], False, False)
        cfg = _settings_struct(rs, device, M, keep)
        fr = _lib.RasterFramesC(F, G if per_frame else 0, 35 if (cameras.dim() == 2 and cameras.shape[0] == F and F > 1) else 0)
        g_color = torch.zeros(F, 3, H, W, device=device) if g_color is None else _f32c(g_color)
        g_depth = None if g_depth is None else _f32c(g_depth)
        g_alpha = None if g_alpha is None else _f32c(g_alpha)
        new = lambda *tail: torch.empty((F, G) + tail, device=device)      # noqa: E731
        d_means3D, d_means2D, d_opac = new(3), new(3), new()
        d_sh = new(M, 3) if shs is not None else None
        d_colors = new(3) if colors_precomp is not None else None
        d_scales = new(3) if scales is not None else None
        d_rots = new(4) if rotations is not None else None
        d_cov = new(6) if cov3D is not None else None
        ws_grad = torch.empty(F * max(G, 1) * 12, dtype=torch.float32, device=device)
        p = _lib.ptr
        _lib.check(L.dwg_raster_backward_frames(ctypes.byref(cfg), ctypes.byref(fr), G, p(means3D), p(shs), p(colors_precomp), p(opac), p(scales),
                                                p(rotations), p(cov3D), p(ws_geom), p(ws_pairs), cap, p(ws_image), p(ws_grad), p(g_color),
                                                p(g_depth), p(g_alpha), p(d_means3D), p(d_means2D), p(d_sh), p(d_colors), p(d_opac), p(d_scales),
                                                p(d_rots), p(d_cov), _stream(device)), "dwg_raster_backward_frames")

        def fold(t, like):
            """[F, G, ...] gradient rows -> the input's shape: per-frame inputs take their rows, shared inputs the sum over the frames."""
            if t is None or like is None:
                return None
            if not per_frame:
                acc = t[0]
                for f in range(1, F):           # frame order, one add per frame: the same bits as F single-frame backwards accumulated in turn
                    acc = acc + t[f]
                t = acc
            return t.reshape(like.shape)
        return (fold(d_means3D, means3D), fold(d_opac, opac), fold(d_colors, colors_precomp), fold(d_sh, shs), fold(d_scales, scales),
                fold(d_rots, rotations), fold(d_cov, cov3D)) + (None,) * 11


def rasterize_frames(means3D, opacities, colors_precomp=None, shs=None, scales=None, rotations=None, cov3D_precomp=None, *,
                     cameras, image_height, image_width, tanfovx, tanfovy, bg, sh_degree=0, scale_modifier=1.0, pair_capacity=None,
                     pair_state: Optional[PairCapacity] = None):
    """F frames in ONE launch chain (include/dwg_raster.h `dwg_raster_frames`), differentiable: the playback path calls it under no_grad,
    the batched multi-view step with the V views of the step as its frames.

    Per-Gaussian inputs are either [F, G, ...] (one posed set of Gaussians per frame) or [G, ...] (shared by all frames);
    `cameras` is [F, 35] or [35] float32 = [viewmatrix 16 | projmatrix 16 | campos 3] rows as `dwg_raster_camera_setup` writes them.
    Frame f's outputs -- and, in the backward, its gradient rows -- are bit-identical to a single-frame call with its inputs (the
    reference's call at /root/reference/core/gaussian/gaussian_renderer.py:186-195, once per frame).
    -> color [F,3,H,W], radii [F,G] int32, depth [F,1,H,W], alpha [F,1,H,W], info dict(headers [F,4] device int32, capacity)."""
    if not means3D.is_cuda:
        raise RuntimeError("dreamwaltz_g_amd rasterizer runs on the GPU only (HIP kernels); got a CPU tensor")
    per_frame = means3D.dim() == 3
    cameras = _f32c(cameras)
    F = int(means3D.shape[0]) if per_frame else (int(cameras.shape[0]) if cameras.dim() == 2 else 1)
    if cameras.dim() == 2 and int(cameras.shape[0]) not in (1, F):
        raise ValueError("rasterize_frames: %d cameras for %d frames" % (cameras.shape[0], F))
    G = int(means3D.shape[-2])

    def arr(t, tail):
        if t is None:
            return None
        t = _f32c(t)
        want = ((F, G) if per_frame else (G,)) + tail
        if tuple(t.shape) != want and not (tail == () and tuple(t.shape) == want + (1,)):
            raise ValueError("rasterize_frames: expected shape %s, got %s" % (want, tuple(t.shape)))
        return t
    means3D = arr(means3D, (3,)); opac = arr(opacities, ())
    colors_precomp = arr(colors_precomp, (3,)); scales = arr(scales, (3,)); rotations = arr(rotations, (4,)); cov3D = arr(cov3D_precomp, (6,))
    if shs is not None:
        shs = _f32c(shs)
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    info = {}
    color, radii, depth, alpha = _RasterizeFrames.apply(means3D, opac, colors_precomp, shs, scales, rotations, cov3D, cameras, int(image_height),
                                                        int(image_width), tanfovx, tanfovy, bg, sh_degree, scale_modifier, pair_capacity, info, pair_state)
    return color, radii, depth, alpha, info


def morton_order(positions: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """int32 permutation that walks `positions` [G,3] along a Z-order curve of their bounding box (3 x `bits` bits).  Gaussians that
    are neighbours in the walk are neighbours in space, hence on screen: a workgroup of the binning stages then touches a few
    hundred 8x8 blocks instead of thousands, and its block-private histogram merges the global atomics (include/dwg_raster.h,
    `visit_order`).  A handful of small torch launches; callers refresh it every few dozen frames, not per frame."""
    with torch.no_grad():
        p = positions.detach().float()
        lo, hi = p.amin(dim=0), p.amax(dim=0)
        q = ((p - lo) / (hi - lo).clamp_min(1e-12) * float((1 << bits) - 1)).to(torch.int64).clamp_(0, (1 << bits) - 1)
        if bits > 10:
            raise ValueError("morton_order: at most 10 bits per axis")
        for shift, mask in ((16, 0x030000FF), (8, 0x0300F00F), (4, 0x030C30C3), (2, 0x09249249)):      # spread the bits 3 apart
            q = (q | (q << shift)) & mask
        code = q[:, 0] | (q[:, 1] << 1) | (q[:, 2] << 2)
        return torch.argsort(code).to(torch.int32).contiguous()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, pair_state=None, info=None, visit_order=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, pair_state, {} if info is None else info, visit_order)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, pair_state: Optional[PairCapacity] = None,
                 visit_order: Optional[torch.Tensor] = None):
        super().__init__()
        self.raster_settings = raster_settings
        self.visit_order = visit_order      # optional int32 permutation: the order in which binning walks the Gaussians (see morton_order)
        self.pair_state = pair_state        # None: exact sizing through a 16-byte read-back per frame (the reference's behaviour)
        self.info = {}                      # num_pairs / num_pairs_ref of the last synchronous forward

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test used by some callers of the original package: p_view.z > 0.2."""
        with torch.no_grad():
            v = self.raster_settings.viewmatrix
            z = positions[:, 0] * v[0, 2] + positions[:, 1] * v[1, 2] + positions[:, 2] * v[2, 2] + v[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   rs, self.pair_state, self.info, self.visit_order)

    @property
    def last_num_pairs(self):
        """(pairs after exact culling, pairs by the reference's 3-sigma tile count) of the most recent resolved forward."""
        if self.pair_state is not None:
            self.pair_state.resolve()
            return self.pair_state.last_num_pairs, self.pair_state.last_num_pairs_ref
        return self.info.get("num_pairs", 0), self.info.get("num_pairs_ref", 0)
