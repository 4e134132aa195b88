"""Per-Gaussian MLPs of the avatar on the f32-exact MFMA GEMM (csrc/gemm.hip), with hand-written autograd.

Class / parameter names mirror the reference so its state_dicts load unchanged:
  MLP            /root/reference/core/nerf/nerf_model.py:12-33           (`net.{l}.weight/bias`, ReLU between layers)
  DeformNetwork  /root/reference/core/deformation/deform_model.py:61-143 (`layers.{i}`, `gaussian_warp`,
                 `gaussian_rotation`, `gaussian_scaling`; leaky-ReLU; xyz_input_ch given -> identity embedding;
                 D=4, W=64, no skips, is_6dof=False -- the configuration DreamWaltzG builds at avatar.py:1171-1174)
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib, gemm


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _splitk(M):
    return max(1, min(256, (M + 1023) // 1024))


class _Linear(torch.autograd.Function):
    """y = act(x @ w[:, :Kx]^T + c), c = b + w[:, Kx:] @ extra  (extra: a vector broadcast to every row, e.g. the body pose
    that deform_model.py:113-115 expands and concatenates)."""

    @staticmethod
    def forward(ctx, x, w, b, extra, act):
        x = x.contiguous().float()
        Kx = x.shape[1]
        wc = w.contiguous().float()
        bias = b.float() if b is not None else torch.zeros(w.shape[0], device=x.device)
        if extra is not None:
            e = extra.reshape(1, -1).contiguous().float()
            # c = extra @ w[:, Kx:]^T + b   (1 x Ke) x (Ke x N)
            c = torch.empty(1, w.shape[0], device=x.device)
            gemm.gemm_raw(e, wc[:, Kx:], c, 1, w.shape[0], e.shape[1], (e.shape[1], 1), (wc.stride(0), 1), w.shape[0],
                          bias=bias.contiguous(), name="mlp_pose_bias")
            bias = c.reshape(-1)
        y = torch.empty(x.shape[0], w.shape[0], device=x.device)
        gemm.gemm_raw(x, wc, y, x.shape[0], w.shape[0], Kx, (Kx, 1), (wc.stride(0), 1), w.shape[0], bias=bias.contiguous(),
                      act=act, name="mlp_fwd")
        ctx.save_for_backward(x, wc, y, extra)
        ctx.act = act
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, extra = ctx.saved_tensors
        M, Kx = x.shape
        N = w.shape[0]
        dy = dy.contiguous().float()
        dz = torch.empty_like(dy) if ctx.act else dy
        colsum = torch.zeros(N, device=x.device)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_act_backward_colsum(M, N, gemm.ACT[ctx.act], p(dy), p(y) if ctx.act else None,
                                                      p(dz) if ctx.act else None, p(colsum), _st(x)), "dwg_act_backward_colsum")
        dw = torch.zeros_like(w)
        # dW[:, :Kx] = dz^T x   (contraction over the M rows of two row-major operands)
        if N <= 64 and Kx <= 64:
            L = _lib.lib()
            ws = torch.empty(L.dwg_mlp_wgrad_workspace_floats(M), device=x.device, dtype=torch.float32)
            _lib.check(L.dwg_mlp_wgrad(M, N, Kx, p(dz), N, p(x), Kx, p(dw), w.shape[1], p(ws), _st(x)), "dwg_mlp_wgrad")
        else:
            gemm.gemm_raw(dz, x, dw, N, Kx, M, (1, N), (1, Kx), w.shape[1], splitk=_splitk(M), name="mlp_wgrad")
        if extra is not None:
            dw[:, Kx:] = colsum[:, None] * extra.reshape(1, -1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm.gemm_raw(dz, w, dx, M, Kx, N, (N, 1), (1, w.stride(0)), Kx, name="mlp_dgrad")
        return dx, dw, (colsum if ctx.has_b else None), None, None


def linear(x, w, b=None, act=None, extra=None):
    return _Linear.apply(x, w, b, extra, act)


class MLP(nn.Module):
    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        net = []
        for l in range(num_layers):
            net.append(nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=bias))
        self.net = nn.ModuleList(net)

    def forward(self, x):
        for l in range(self.num_layers):
            x = linear(x, self.net[l].weight, self.net[l].bias, act="relu" if l != self.num_layers - 1 else None)
        return x


class DeformNetwork(nn.Module):
    def __init__(self, xyz_input_ch=32, pose_input_ch=63, D=4, W=64, multires=10, residual=False, is_6dof=False):
        super().__init__()
        if residual or is_6dof or xyz_input_ch is None:
            raise NotImplementedError("only the configuration DreamWaltzG uses (avatar.py:1171-1174) is on the hot path")
        self.D, self.W = D, W
        self.input_ch = xyz_input_ch + pose_input_ch
        self.xyz_input_ch = xyz_input_ch
        self.layers = nn.ModuleList([nn.Linear(self.input_ch, W)] + [nn.Linear(W, W) for _ in range(D - 1)])
        self.gaussian_warp = nn.Linear(W, 3)
        self.gaussian_rotation = nn.Linear(W, 4)
        self.gaussian_scaling = nn.Linear(W, 3)

    def forward(self, x, body_pose):
        h = linear(x, self.layers[0].weight, self.layers[0].bias, act="leaky_relu", extra=body_pose)
        for i in range(1, self.D):
            h = linear(h, self.layers[i].weight, self.layers[i].bias, act="leaky_relu")
        # the three heads share one 64 -> 10 product (warp 3 | scaling 3 | rotation 4)
        w = torch.cat([self.gaussian_warp.weight, self.gaussian_scaling.weight, self.gaussian_rotation.weight], 0)
        b = torch.cat([self.gaussian_warp.bias, self.gaussian_scaling.bias, self.gaussian_rotation.bias], 0)
        o = linear(h, w, b)
        return o[:, 0:3], o[:, 3:6], o[:, 6:10]
