"""Python bindings of include/dwg_nn.h (GroupNorm, LayerNorm, GEGLU, fused attention, row softmax)."""
import ctypes

import torch

from . import _lib


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(t, dtype=torch.bfloat16):
    if not t.is_cuda:
        raise RuntimeError("dreamwaltz_g_amd nn ops run on the GPU only (HIP kernels)")
    assert t.dtype == dtype and t.is_contiguous(), (t.dtype, t.is_contiguous())


def groupnorm(x, gamma, beta, groups=32, eps=1e-5, silu=False, out=None, stats=None):
    """x [B, HW, C] (or [B,H,W,C]) bf16 NHWC -> same shape; returns (y, stats[B,G,2])."""
    _chk(x)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x) if out is None else out
    stats = torch.empty(B, groups, 2, device=x.device, dtype=torch.float32) if stats is None else stats
    ws = torch.empty(_lib.lib().dwg_groupnorm_workspace_floats(B, groups), device=x.device, dtype=torch.float32)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_groupnorm_forward(B, HW, C, groups, p(x), p(gamma), p(beta), eps, int(silu), p(y), p(stats), p(ws),
                                                _st(x)), "dwg_groupnorm_forward")
    return y, stats


def groupnorm_backward(x, dy, stats, gamma, beta, groups=32, eps=1e-5, silu=False, residual=None):
    _chk(x); _chk(dy)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    dx = torch.empty_like(x)
    scratch = torch.empty(B, groups, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().dwg_groupnorm_workspace_floats(B, groups), device=x.device, dtype=torch.float32)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_groupnorm_backward(B, HW, C, groups, p(x), p(dy), p(stats), p(gamma), p(beta), eps, int(silu), p(dx),
                                                 p(scratch), p(ws), p(residual), _st(x)), "dwg_groupnorm_backward")
    return dx


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _chk(x)
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x) if out is None else out
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_layernorm_forward(M, C, p(x), p(gamma), p(beta), eps, p(y), _st(x)), "dwg_layernorm_forward")
    return y


def geglu(x, out=None):
    _chk(x)
    F2 = x.shape[-1]
    M = x.numel() // F2
    y = torch.empty(*x.shape[:-1], F2 // 2, device=x.device, dtype=x.dtype) if out is None else out
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_geglu_forward(M, F2 // 2, p(x), p(y), _st(x)), "dwg_geglu_forward")
    return y


def attention(q, k, v, heads, out=None, scale=None):
    """q [B, Nq, H*d], k/v [B, Nk, H*d] bf16 or fp16 (last dim contiguous; may be column slices of a fused projection)."""
    B, Nq, HD = q.shape
    Nk = k.shape[1]
    d = HD // heads
    o = torch.empty(B, Nq, HD, device=q.device, dtype=q.dtype) if out is None else out
    scale = d ** -0.5 if scale is None else scale
    for t in (q, k, v, o):
        assert t.stride(2) == 1 and t.dtype == q.dtype and t.dtype in (torch.bfloat16, torch.float16)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_attention_forward_dt(1 if q.dtype == torch.bfloat16 else 2, B, heads, Nq, Nk, d, p(q), q.stride(1), q.stride(0), p(k), k.stride(1), k.stride(0),
                                                p(v), v.stride(1), v.stride(0), p(o), o.stride(1), o.stride(0), scale, _st(q)),
               "dwg_attention_forward_dt")
    return o


def softmax_rows(S, scale=1.0, out=None):
    rows, n = S.shape
    P = torch.empty(rows, n, device=S.device, dtype=torch.bfloat16) if out is None else out
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_softmax_rows_forward(rows, n, scale, p(S), S.stride(0), p(P), P.stride(0), _st(S)),
               "dwg_softmax_rows_forward")
    return P


def softmax_rows_backward(P, dP, scale=1.0):
    rows, n = P.shape
    dS = torch.empty(rows, n, device=P.device, dtype=torch.bfloat16)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_softmax_rows_backward(rows, n, scale, p(P), P.stride(0), p(dP), dP.stride(0), p(dS), dS.stride(0),
                                                    _st(P)), "dwg_softmax_rows_backward")
    return dS
