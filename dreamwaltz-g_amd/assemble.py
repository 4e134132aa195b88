"""Per-Gaussian activations and non-rigid composition of DreamWaltzG.animate on the HIP kernels of csrc/assemble.hip
(include/dwg_gaussian.h): non_rigid_transform with the default flags (/root/reference/core/system/avatar.py:1464-1498),
static_mlp_forward (:1283-1290) and the GaussianModel activations (core/gaussian/gaussian_model.py:25-56) -- one launch forward,
one backward, instead of ~45 element-wise PyTorch kernels including autograd."""
import ctypes

import torch

from . import _lib


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _Assemble(torch.autograd.Function):
    """`mlp`: the deformation network's packed output [N, ld] = [warp 3 | scaling 3 | rest]: its column blocks are read in place and their
    gradient comes back as ONE [N, ld] tensor (zeros in the unused columns) -- no slice copies, no slice-backward zero-fill + copy pairs."""

    @staticmethod
    def forward(ctx, positions, mlp, log_scales, quaternions, h, init_offset, init_scale):
        for t in (positions, mlp, log_scales, quaternions, h):
            if not t.is_cuda:
                raise RuntimeError("dreamwaltz_g_amd.assemble: HIP-only path, got a CPU tensor (no CPU fallback)")
        c = lambda t: t.contiguous().float()  # noqa: E731
        positions, mlp, log_scales, quaternions, h = map(c, (positions, mlp, log_scales, quaternions, h))
        n, nt, ld = positions.shape[0], h.shape[0], int(mlp.shape[1])
        assert mlp.shape[0] == n and ld >= 6, mlp.shape
        dev = h.device
        pos, scl = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        qn = torch.empty(n, 4, device=dev)
        col, op = torch.empty(nt, 3, device=dev), torch.empty(nt, 1, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_gaussian_assemble_forward_ld(n, nt, p(positions), p(mlp), init_offset, p(log_scales),
                                                               ctypes.c_void_p(mlp.data_ptr() + 12), ld, init_scale, p(quaternions), p(h), p(pos),
                                                               p(scl), p(qn), p(col), p(op), _st(h)), "dwg_gaussian_assemble_forward_ld")
        ctx.save_for_backward(log_scales, quaternions, h)
        ctx.k = (n, nt, ld, init_offset, init_scale)
        return pos, scl, qn, col, op

    @staticmethod
    def backward(ctx, g_pos, g_scl, g_qn, g_col, g_op):
        log_scales, quaternions, h = ctx.saved_tensors
        n, nt, ld, io, isc = ctx.k
        dev = h.device
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        g_pos, g_scl, g_qn, g_col, g_op = map(c, (g_pos, g_scl, g_qn, g_col, g_op))
        d_p, d_ls = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        d_mlp = torch.empty(n, ld, device=dev)
        d_q, d_h = torch.empty(n, 4, device=dev), torch.empty(nt, 4, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_gaussian_assemble_backward_ld(n, nt, io, p(log_scales), isc, p(quaternions), p(h), p(g_pos), p(g_scl),
                                                                p(g_qn), p(g_col), p(g_op), p(d_p), p(d_mlp), p(d_ls),
                                                                ctypes.c_void_p(d_mlp.data_ptr() + 12), ld, ld - 6, p(d_q), p(d_h), _st(h)),
                   "dwg_gaussian_assemble_backward_ld")
        return d_p, d_mlp, d_ls, d_q, d_h, None, None


def assemble(positions, offsets, log_scales, mlp_scales, quaternions, h, init_offset, init_scale):
    """-> (pos [N,3], scales [N,3], unit quaternions [N,4], colours [Nt,3], opacities [Nt,1]); h = static-MLP output [Nt,4], rows
    >= N are mesh-bound Gaussians (colours only, opacity 1)."""
    return _Assemble.apply(positions, torch.cat([offsets, mlp_scales], dim=1), log_scales, quaternions, h, float(init_offset), float(init_scale))


def assemble_packed(positions, mlp, log_scales, quaternions, h, init_offset, init_scale):
    """`assemble` with offsets = mlp[:, 0:3] and mlp_scales = mlp[:, 3:6] read in place (the deformation network's packed output)."""
    return _Assemble.apply(positions, mlp, log_scales, quaternions, h, float(init_offset), float(init_scale))


class _CopySegments(torch.autograd.Function):
    """Several row-block concatenations in ONE launch (dwg_copy_segments): outputs[k] = cat(parts of group k, dim 0); with `bound`
    every value goes out as (v + bound) / (2 bound).  Backward: row slices of the incoming gradients (views; one division when normalised)."""

    @staticmethod
    def forward(ctx, bound, sizes, *parts):
        dev = parts[0].device
        outs, segs, k = [], [], 0
        parts = [t.contiguous().float() for t in parts]
        for n in sizes:
            group = parts[k:k + n]
            k += n
            rows = sum(int(t.shape[0]) for t in group)
            out = torch.empty((rows,) + tuple(group[0].shape[1:]), device=dev)
            off = 0
            for t in group:
                if t.numel():
                    segs.append((out.data_ptr() + off * 4, t.data_ptr(), t.numel()))
                off += t.numel()
            outs.append(out)
        if len(segs) > 12:
            raise ValueError("copy_segments: at most 12 blocks per launch")
        arr = (_lib.SegmentC * max(len(segs), 1))(*[_lib.SegmentC(d, s_, c) for d, s_, c in segs])
        add, div = (float(bound), 2.0 * float(bound)) if bound is not None else (0.0, 0.0)
        _lib.check(_lib.lib().dwg_copy_segments(len(segs), ctypes.cast(arr, ctypes.c_void_p), add, div, _st(parts[0])), "dwg_copy_segments")
        ctx.bound, ctx.sizes, ctx.rows = bound, tuple(sizes), [int(t.shape[0]) for t in parts]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        res, k = [], 0
        for n, g in zip(ctx.sizes, grads):
            if g is not None and ctx.bound is not None:
                g = g / (2.0 * float(ctx.bound))
            off = 0
            for r in ctx.rows[k:k + n]:
                res.append(None if g is None else g[off:off + r])
                off += r
            k += n
        return (None, None) + tuple(res)


def concat_rows(groups, bound=None):
    """[[a0, a1, ...], [b0, b1, ...], ...] -> (cat(a*), cat(b*), ...) along dim 0 in one launch; `bound`: normalise (v + bound) / (2 bound)."""
    sizes = [len(g) for g in groups]
    return _CopySegments.apply(bound, sizes, *[t for g in groups for t in g])
