"""Python binding of the MFMA GEMM / implicit-GEMM conv primitive (include/dwg_gemm.h, csrc/gemm.hip)."""
import ctypes

import torch

from . import _lib

F32, BF16, F16, F32X = 0, 1, 2, 3        # DWG_DTYPE_* (include/dwg_types.h); F32X tensors are int32-typed (xfmt.py)
ACT = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2, "silu": 3, "gelu": 4, "sigmoid": 5, "geglu_pair": 6}


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("a_row_stride", ctypes.c_int64), ("a_k_stride", ctypes.c_int64), ("b_row_stride", ctypes.c_int64),
        ("b_k_stride", ctypes.c_int64), ("ldc", ctypes.c_int64), ("ldr", ctypes.c_int64),
        ("batch1", ctypes.c_int32), ("batch2", ctypes.c_int32),
        ("a_batch1_stride", ctypes.c_int64), ("a_batch2_stride", ctypes.c_int64), ("b_batch1_stride", ctypes.c_int64),
        ("b_batch2_stride", ctypes.c_int64), ("c_batch1_stride", ctypes.c_int64), ("c_batch2_stride", ctypes.c_int64),
        ("r_batch1_stride", ctypes.c_int64), ("r_batch2_stride", ctypes.c_int64),
        ("dtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32), ("residual_dtype", ctypes.c_int32),
        ("act", ctypes.c_int32), ("alpha", ctypes.c_float), ("bias_per_row", ctypes.c_int32),
        ("splitk", ctypes.c_int32), ("accumulate", ctypes.c_int32),
        ("conv_enabled", ctypes.c_int32), ("conv_cin", ctypes.c_int32), ("conv_hin", ctypes.c_int32),
        ("conv_win", ctypes.c_int32), ("conv_hout", ctypes.c_int32), ("conv_wout", ctypes.c_int32),
        ("conv_kh", ctypes.c_int32), ("conv_kw", ctypes.c_int32), ("conv_stride", ctypes.c_int32),
        ("conv_pad_t", ctypes.c_int32), ("conv_pad_l", ctypes.c_int32), ("conv_in_dilation", ctypes.c_int32),
        ("conv_in_upsample", ctypes.c_int32), ("A2", ctypes.c_void_p), ("conv_cin1", ctypes.c_int32),
        ("bias_row_div", ctypes.c_int32), ("bias_ld", ctypes.c_int64),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("force_register_staging", ctypes.c_int32), ("workspace_counters", ctypes.c_int32),
        ("name", ctypes.c_char_p),
    ]


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.int32:
        return F32X
    raise TypeError("dwg gemm supports float32, bfloat16, float16 and f32x (int32-typed, xfmt.py) tensors, got %s" % t.dtype)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def gemm_raw(A, B, C, M, N, K, a_strides, b_strides, ldc, bias=None, residual=None, ldr=0, act=None, alpha=1.0,
             batch=(1, 1), a_batch=(0, 0), b_batch=(0, 0), c_batch=(0, 0), r_batch=(0, 0), bias_per_row=False, splitk=1,
             accumulate=False, conv=None, name=None, conv_upsample=1, A2=None, cin1=0, bias_row_div=0, bias_ld=0, run=True):
    """Thin descriptor builder; all strides in elements.  A/B/C/bias/residual are CUDA tensors (used for their pointers)."""
    if not A.is_cuda:
        raise RuntimeError("dreamwaltz_g_amd GEMM runs on the GPU only (HIP kernels)")
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.M, d.N, d.K = M, N, K
    d.a_row_stride, d.a_k_stride = a_strides
    d.b_row_stride, d.b_k_stride = b_strides
    d.ldc, d.ldr = ldc, ldr
    d.batch1, d.batch2 = batch
    d.a_batch1_stride, d.a_batch2_stride = a_batch
    d.b_batch1_stride, d.b_batch2_stride = b_batch
    d.c_batch1_stride, d.c_batch2_stride = c_batch
    d.r_batch1_stride, d.r_batch2_stride = r_batch
    d.dtype = _dt(A)
    assert _dt(B) == d.dtype, "A and B must share a dtype"
    d.out_dtype = _dt(C)
    d.residual_dtype = _dt(residual) if residual is not None else F32
    d.act = ACT[act]
    d.alpha = alpha
    d.bias_per_row = int(bias_per_row)
    d.splitk = splitk
    d.accumulate = int(accumulate)
    if conv is not None:
        d.conv_enabled = 1
        (d.conv_cin, d.conv_hin, d.conv_win, d.conv_hout, d.conv_wout, d.conv_kh, d.conv_kw, d.conv_stride, d.conv_pad_t,
         d.conv_pad_l, d.conv_in_dilation) = conv
    d.conv_in_upsample = conv_upsample
    if A2 is not None:
        d.A2 = A2.data_ptr(); d.conv_cin1 = cin1
    d.bias_row_div = bias_row_div
    d.bias_ld = bias_ld
    d.name = name.encode() if name else None
    if not run:
        return d
    _lib.check(_lib.lib().dwg_gemm(ctypes.byref(d), _stream(A)), "dwg_gemm")
    return C


def run_desc(d, stream):
    """Launch a prebuilt descriptor (static plans: no per-step Python work besides this call)."""
    rc = _lib.lib().dwg_gemm(ctypes.byref(d), stream)
    if rc != 0:
        raise RuntimeError("dwg_gemm failed with DWG error %d" % rc)


def linear(x, w, bias=None, act=None, out_dtype=None, residual=None, out=None, name=None):
    """y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ residual).  x may have leading dims; last-dim contiguous."""
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    M, N = x2.shape[0], w.shape[0]
    out_dtype = out_dtype or x.dtype
    y = out if out is not None else torch.empty(M, N, device=x.device, dtype=out_dtype)
    r2 = None if residual is None else residual.reshape(M, N)
    gemm_raw(x2, w, y, M, N, K, (x2.stride(0), x2.stride(1)), (w.stride(0), w.stride(1)), y.stride(0), bias=bias, residual=r2,
             ldr=0 if r2 is None else r2.stride(0), act=act, name=name)
    return y.reshape(*x.shape[:-1], N)


def conv2d_nhwc(x, w, bias=None, stride=1, pad=(1, 1), act=None, residual=None, out_hw=None, in_dilation=1, out_dtype=None,
                name=None):
    """x [B,H,W,Cin] bf16 NHWC, w [Cout,KH,KW,Cin] bf16 -> [B,Ho,Wo,Cout].  pad = (top, left); bottom/right are implied by
    out_hw (defaults to the symmetric-padding size)."""
    Bn, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Hv, Wv = (H - 1) * in_dilation + 1, (W - 1) * in_dilation + 1
    if out_hw is None:
        Ho = (Hv + 2 * pad[0] - KH) // stride + 1
        Wo = (Wv + 2 * pad[1] - KW) // stride + 1
    else:
        Ho, Wo = out_hw
    y = torch.empty(Bn, Ho, Wo, Cout, device=x.device, dtype=out_dtype or x.dtype)
    M, N, K = Bn * Ho * Wo, Cout, KH * KW * Cin
    gemm_raw(x, w, y, M, N, K, (0, 1), (K, 1), Cout, bias=bias, residual=residual, ldr=Cout if residual is not None else 0,
             act=act, conv=(Cin, H, W, Ho, Wo, KH, KW, stride, pad[0], pad[1], in_dilation), name=name)
    return y
