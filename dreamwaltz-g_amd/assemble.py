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
    @staticmethod
    def forward(ctx, positions, offsets, log_scales, mlp_scales, quaternions, h, init_offset, init_scale):
        for t in (positions, offsets, log_scales, mlp_scales, quaternions, h):
            if not t.is_cuda:
                raise RuntimeError("dreamwaltz_g_amd.assemble: HIP-only path, got a CPU tensor (no CPU fallback)")
        c = lambda t: t.contiguous().float()  # noqa: E731
        positions, offsets, log_scales, mlp_scales, quaternions, h = map(c, (positions, offsets, log_scales, mlp_scales, quaternions, h))
        n, nt = positions.shape[0], h.shape[0]
        dev = h.device
        pos, scl = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        qn = torch.empty(n, 4, device=dev)
        col, op = torch.empty(nt, 3, device=dev), torch.empty(nt, 1, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_gaussian_assemble_forward(n, nt, p(positions), p(offsets), init_offset, p(log_scales), p(mlp_scales),
                                                            init_scale, p(quaternions), p(h), p(pos), p(scl), p(qn), p(col), p(op),
                                                            _st(h)), "dwg_gaussian_assemble_forward")
        ctx.save_for_backward(log_scales, quaternions, h)
        ctx.k = (n, nt, init_offset, init_scale)
        return pos, scl, qn, col, op

    @staticmethod
    def backward(ctx, g_pos, g_scl, g_qn, g_col, g_op):
        log_scales, quaternions, h = ctx.saved_tensors
        n, nt, io, isc = ctx.k
        dev = h.device
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        g_pos, g_scl, g_qn, g_col, g_op = map(c, (g_pos, g_scl, g_qn, g_col, g_op))
        d_p, d_off, d_ls, d_ms = (torch.empty(n, 3, device=dev) for _ in range(4))
        d_q, d_h = torch.empty(n, 4, device=dev), torch.empty(nt, 4, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_gaussian_assemble_backward(n, nt, io, p(log_scales), isc, p(quaternions), p(h), p(g_pos), p(g_scl),
                                                             p(g_qn), p(g_col), p(g_op), p(d_p), p(d_off), p(d_ls), p(d_ms), p(d_q),
                                                             p(d_h), _st(h)), "dwg_gaussian_assemble_backward")
        return d_p, d_off, d_ls, d_ms, d_q, d_h, None, None


def assemble(positions, offsets, log_scales, mlp_scales, quaternions, h, init_offset, init_scale):
    """-> (pos [N,3], scales [N,3], unit quaternions [N,4], colours [Nt,3], opacities [Nt,1]); h = static-MLP output [Nt,4], rows
    >= N are mesh-bound Gaussians (colours only, opacity 1)."""
    return _Assemble.apply(positions, offsets, log_scales, mlp_scales, quaternions, h, float(init_offset), float(init_scale))
