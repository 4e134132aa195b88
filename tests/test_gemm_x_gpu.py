"""-m gpu parity tests of the split-precision ("f32x", DWG_DTYPE_F32X) unit of the MFMA GEMM / implicit-GEMM convolution (csrc/gemm_x.hip,
csrc/dwg_xfmt.h) against float64 references on the SAME fp32 inputs: operands are packed into hi / lo fp16 halves, every product is three
16-bit MFMAs with fp32 accumulation.  The bar is the one of the exact-f32 MFMA kernels (tests/test_gemm_gpu.py: 1e-5 of the output scale) --
fp32-grade results, which is what the reference's fp32 guidance stage (configs/__init__.py:236,241) produces."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5        # relative to the output scale; measured 2e-7 .. 1.5e-6 (profiles/r04_parity_f32x.json)


def _rel(a, r):
    return float((a.double() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-20))


def _x(t):
    from dreamwaltz_g_amd import xfmt
    return xfmt.pack(t).cuda()


def _u(t):
    from dreamwaltz_g_amd import xfmt
    return xfmt.unpack(t.cpu())


def test_pack_unpack_round_trip_on_the_gpu_matches_the_format():
    from dreamwaltz_g_amd import xfmt
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 5, 64, generator=g) * torch.logspace(-3, 4, 64)
    y = xfmt.unpack(xfmt.pack(x.cuda())).cpu()
    assert float(((y - x).abs() / x.abs().clamp_min(1e-4)).max()) < 5e-7      # 22 significand bits
    assert float(xfmt.unpack(xfmt.pack(torch.full((8,), 1e9))).max()) == 65504.0   # saturates instead of overflowing


@pytest.mark.parametrize("M,N,K", [(8192, 320, 320), (2048, 1280, 640), (154, 768, 320), (100, 40, 4096), (1, 8, 8), (300, 72, 64)])
def test_x_linear_bias_act_residual(M, N, K):
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.silu(x.double() @ w.double().t() + b.double()) + r.double()
    xc, wc, bc, rx = _x(x), _x(w), b.cuda(), _x(r)
    y = gemm.linear(xc, wc, bc, act="silu", residual=rx, out_dtype=torch.float32)          # fp32 out, f32x residual
    assert y.dtype == torch.float32 and _rel(y.cpu(), ref) < TOL
    yx = gemm.linear(xc, wc, bc, act="silu", residual=rx)                                   # f32x out
    assert yx.dtype == torch.int32 and _rel(_u(yx), ref) < TOL
    yf = gemm.linear(xc, wc, bc, act="silu", residual=r.cuda(), out_dtype=torch.float32)    # fp32 residual
    assert _rel(yf.cpu(), ref) < TOL


def test_x_small_magnitudes_keep_fp32_grade_precision():
    """Operands far below 1 (weights ~ 1e-3, gradients ~ 1e-4): the scaled lo plane keeps 22 bits down to |x| = 6e-5, where a plain
    fp16 residual would already be subnormal."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 256, 1024
    x = torch.randn(M, K, generator=g) * 3e-4; w = torch.randn(N, K, generator=g) * 1e-3
    ref = x.double() @ w.double().t()
    y = gemm.linear(_x(x), _x(w), None, out_dtype=torch.float32)
    assert _rel(y.cpu(), ref) < TOL


def test_x_batched_attention_scores_in_place_heads():
    """QK^T for [B, N, heads*d] projections addressed in place through (image, head) batch strides (d = 40: five 8-channel groups)."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(1)
    Bn, Nq, Nk, Hh, d = 2, 256, 77, 8, 40
    q = torch.randn(Bn, Nq, Hh * d, generator=g); k = torch.randn(Bn, Nk, Hh * d, generator=g)
    S = torch.empty(Bn, Hh, Nq, Nk, device="cuda", dtype=torch.float32)
    gemm.gemm_raw(_x(q), _x(k), S, Nq, Nk, d, (Hh * d, 1), (Hh * d, 1), Nk, alpha=d ** -0.5, batch=(Bn, Hh),
                  a_batch=(Nq * Hh * d, d), b_batch=(Nk * Hh * d, d), c_batch=(Hh * Nq * Nk, Nq * Nk))
    qh = q.double().view(Bn, Nq, Hh, d).permute(0, 2, 1, 3); kh = k.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
    assert _rel(S.cpu(), qh @ kh.transpose(-1, -2) * d ** -0.5) < TOL


@pytest.mark.parametrize("Cin,Cout,H,stride,k", [(8, 320, 64, 1, 3), (320, 320, 32, 2, 3), (64, 128, 17, 1, 3), (640, 320, 16, 1, 1),
                                                 (128, 128, 64, 2, 3), (16, 32, 40, 2, 3), (128, 128, 128, 1, 3), (256, 512, 64, 1, 3)])
def test_x_conv_forward(Cin, Cout, H, stride, k):
    """generic im2col loader (Cin = 8, 16), the fast loader (Cin % 32 == 0), 1x1, stride 2, and the LDS-patch kernel (M >= 8192)."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(Cin + Cout)
    Bn = 2
    x = torch.randn(Bn, Cin, H, H, generator=g); w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = k // 2
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    xc, wc = _x(x.permute(0, 2, 3, 1)), _x(w.permute(0, 2, 3, 1))
    y = gemm.conv2d_nhwc(xc, wc, b.cuda(), stride=stride, pad=(pad, pad), out_dtype=torch.float32)
    assert _rel(y.permute(0, 3, 1, 2).cpu(), ref) < TOL
    yx = gemm.conv2d_nhwc(xc, wc, b.cuda(), stride=stride, pad=(pad, pad))
    assert _rel(_u(yx).permute(0, 3, 1, 2), ref) < TOL


def test_x_conv_asymmetric_pad_and_input_gradient():
    """VAE down-sampling conv (F.pad (0,1,0,1) + 3x3 stride 2 pad 0) and the input-gradient of a conv as the dilated gather form."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(3)
    Bn, Cin, Cout, H = 1, 16, 24, 20
    x = torch.randn(Bn, Cin, H, H, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / 12
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xd, (0, 1, 0, 1)), w.double(), stride=2)
    y = gemm.conv2d_nhwc(_x(x.permute(0, 2, 3, 1)), _x(w.permute(0, 2, 3, 1)), None, stride=2, pad=(0, 0), out_hw=(H // 2, H // 2),
                         out_dtype=torch.float32)
    assert _rel(y.permute(0, 3, 1, 2).cpu(), ref) < TOL
    gy = torch.randn(ref.shape, generator=g)
    (gx_ref,) = torch.autograd.grad(ref, xd, gy.double())
    wf = _x(w.flip(2, 3).permute(1, 2, 3, 0))
    gx = gemm.conv2d_nhwc(_x(gy.permute(0, 2, 3, 1)), wf, None, stride=1, pad=(2, 2), out_hw=(H, H), in_dilation=2, out_dtype=torch.float32)
    assert _rel(gx.permute(0, 3, 1, 2).cpu(), gx_ref) < TOL


def test_x_conv_upsample_and_concat_sources():
    """Upsample2D fused into the conv's loader, and the skip concatenation as two NHWC sources (UNet decoder)."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(9)
    Bn, C1, C2, Cout, H = 2, 64, 32, 64, 16
    a = torch.randn(Bn, C1, H, H, generator=g); b2 = torch.randn(Bn, C2, H, H, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / 30
    ref = torch.nn.functional.conv2d(torch.cat([a, b2], 1).double(), w.double(), padding=1)
    y = torch.empty(Bn, H, H, Cout, device="cuda")
    K = 9 * (C1 + C2)
    ax, bx, wx = _x(a.permute(0, 2, 3, 1)), _x(b2.permute(0, 2, 3, 1)), _x(w.permute(0, 2, 3, 1))
    gemm.gemm_raw(ax, wx, y, Bn * H * H, Cout, K, (0, 1), (K, 1), Cout, conv=(C1 + C2, H, H, H, H, 3, 3, 1, 1, 1, 1), A2=bx, cin1=C1)
    assert _rel(y.permute(0, 3, 1, 2).cpu(), ref) < TOL
    wu = torch.randn(Cout, C1, 3, 3, generator=g) / 24
    refu = torch.nn.functional.conv2d(torch.nn.functional.interpolate(a.double(), scale_factor=2, mode="nearest"), wu.double(), padding=1)
    yu = torch.empty(Bn, 2 * H, 2 * H, Cout, device="cuda")
    gemm.gemm_raw(ax, _x(wu.permute(0, 2, 3, 1)), yu, Bn * 4 * H * H, Cout, 9 * C1, (0, 1), (9 * C1, 1), Cout,
                  conv=(C1, H, H, 2 * H, 2 * H, 3, 3, 1, 1, 1, 1), conv_upsample=2)
    assert _rel(yu.permute(0, 3, 1, 2).cpu(), refu) < TOL


def _with_library_splitk(d):
    from dreamwaltz_g_amd import _lib
    d.splitk = 0
    need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(need, 16) // 4, device="cuda")
    if need > 0:
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    else:
        d.splitk = 1
    return need, ws


def test_x_splitk_slab_epilogue_small_m():
    """Small-M layers (8x8 / 16x16 latents): library-chosen split-K with fp32 slabs + the reduce pass that splits its f32x output."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(4)
    M, N, K = 128, 1280, 11520
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.silu(x.double() @ w.double().t() + b.double()) + r.double()
    xc, wc, bc, rx = _x(x), _x(w), b.cuda(), _x(r)
    y = torch.empty(M, N, device="cuda", dtype=torch.int32)
    d = gemm.gemm_raw(xc, wc, y, M, N, K, (K, 1), (K, 1), N, bias=bc, residual=rx, ldr=N, act="silu", run=False)
    need, ws = _with_library_splitk(d)
    assert need > 0
    gemm.run_desc(d, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert _rel(_u(y), ref) < TOL


def test_x_geglu_pair_epilogue():
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(8)
    M, C = 1000, 320
    x = torch.randn(M, C, generator=g)
    w = torch.randn(8 * C, C, generator=g) / C ** 0.5; b = torch.randn(8 * C, generator=g) * 0.1
    h = x.double() @ w.double().t() + b.double()
    hid, gate = h.chunk(2, dim=-1)
    ref = hid * torch.nn.functional.gelu(gate)
    F_ = 4 * C
    idx = torch.arange(F_).view(-1, 32)
    perm = torch.cat([idx, idx + F_], dim=1).reshape(-1)
    y = torch.empty(M, F_, device="cuda", dtype=torch.int32)
    gemm.gemm_raw(_x(x), _x(w[perm]), y, M, 8 * C, C, (C, 1), (C, 1), F_, bias=b[perm].contiguous().cuda(), act="geglu_pair")
    assert _rel(_u(y), ref) < 2e-6 + TOL        # + the branch-free erf of the GEGLU epilogue (6e-7 absolute, dwg_common.h)


@pytest.mark.parametrize("C,H,patch_min_m", [(640, 32, 512), (1280, 16, 512), (640, 32, 8192), (320, 64, 512)])
def test_x_conv3x3_small_latents_splitk_paths(C, H, patch_min_m):
    """3x3 convolutions of the 32x32 / 16x16 latent levels with the library-chosen split-K, through the LDS-patch kernel with split-K over
    the channel slabs (DWG_CONV_PATCH_MINM=512) and through the im2col loader (default); per-image channel bias (time embedding)."""
    import os
    from dreamwaltz_g_amd import gemm, _lib
    g = torch.Generator().manual_seed(C + H)
    Bn = 2
    x = torch.randn(Bn, C, H, H, generator=g); w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    bimg = torch.randn(Bn, C, generator=g); r = torch.randn(Bn, H, H, C, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=1).permute(0, 2, 3, 1) + bimg.double()[:, None, None, :] + r.double()
    xc, wc = _x(x.permute(0, 2, 3, 1)), _x(w.permute(0, 2, 3, 1))
    y = torch.empty(Bn, H, H, C, device="cuda", dtype=torch.int32)
    M, K = Bn * H * H, 9 * C
    os.environ["DWG_CONV_PATCH_MINM"] = str(patch_min_m)
    try:
        bc, rcu = bimg.cuda(), _x(r)
        d = gemm.gemm_raw(xc, wc, y, M, C, K, (0, 1), (K, 1), C, bias=bc, bias_row_div=H * H, bias_ld=C, residual=rcu, ldr=C,
                          conv=(C, H, H, H, H, 3, 3, 1, 1, 1, 1), run=False)
        need, ws = _with_library_splitk(d)
        _lib.prof_enable(True)
        gemm.run_desc(d, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        syms = _lib.prof_symbols(); _lib.prof_enable(False)
    finally:
        os.environ.pop("DWG_CONV_PATCH_MINM", None)
    used_patch = any(k.startswith("k_conv3x3_patch") for k in syms)
    assert used_patch == (M >= patch_min_m), (syms.keys(), M, patch_min_m)
    assert _rel(_u(y), ref) < TOL


@pytest.mark.parametrize("case", ["gemm_small_m", "gemm_qkv", "gemm_ff_out", "conv_1280_16", "conv_640_32", "conv_320_64", "conv_im2col"])
def test_x_splitk_reduce_inside_the_kernel_matches_the_reduce_pass_bit_for_bit(case):
    """dwg_gemm_desc::workspace_counters (round 6): the slice that arrives last at a tile sums the slabs inside the GEMM kernel.  Same
    additions in the same order as k_splitk_epilogue: identical bits, whichever slice is last; the counters are zero again afterwards (the
    second and third run on the same workspace agree); no reduce launch."""
    import os
    from dreamwaltz_g_amd import gemm, _lib
    g = torch.Generator().manual_seed(len(case))
    os.environ.pop("DWG_CONV_PATCH_MINM", None)
    if case.startswith("gemm"):
        M, N, K = {"gemm_small_m": (128, 1280, 11520), "gemm_qkv": (512, 3840, 1280), "gemm_ff_out": (2048, 640, 2560)}[case]
        x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
        xc, wc, bc, rx = _x(x), _x(w), b.cuda(), _x(r)
        shape = (M, N)
        make = lambda y: gemm.gemm_raw(xc, wc, y, M, N, K, (K, 1), (K, 1), N, bias=bc, residual=rx, ldr=N, act="silu", run=False)  # noqa: E731
    else:
        C, H = {"conv_1280_16": (1280, 16), "conv_640_32": (640, 32), "conv_320_64": (320, 64), "conv_im2col": (640, 32)}[case]
        if case == "conv_im2col":
            os.environ["DWG_CONV_PATCH_MINM"] = "1000000"
        Bn = 2
        x = torch.randn(Bn, C, H, H, generator=g); w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
        bimg = torch.randn(Bn, C, generator=g); r = torch.randn(Bn, H, H, C, generator=g)
        xc, wc, bc, rcu = _x(x.permute(0, 2, 3, 1)), _x(w.permute(0, 2, 3, 1)), bimg.cuda(), _x(r)
        M, K = Bn * H * H, 9 * C
        shape = (Bn, H, H, C)
        make = lambda y: gemm.gemm_raw(xc, wc, y, M, C, K, (0, 1), (K, 1), C, bias=bc, bias_row_div=H * H, bias_ld=C, residual=rcu, ldr=C,  # noqa: E731
                                       conv=(C, H, H, H, H, 3, 3, 1, 1, 1, 1), run=False)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        outs, launches = [], []
        for counters in (0, 1, 1, 1):
            y = torch.empty(shape, device="cuda", dtype=torch.int32)
            d = make(y)
            d.splitk = 0
            need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
            if need == 0:
                pytest.skip("the library does not split this shape")
            if counters == 0 or len(outs) == 1:
                ws = torch.zeros(need // 4, device="cuda")          # the fused runs share ONE workspace: counters must come back to zero
            d.workspace, d.workspace_bytes, d.workspace_counters = ws.data_ptr(), need, counters
            _lib.prof_enable(True)
            gemm.run_desc(d, st)
            torch.cuda.synchronize()
            launches.append(_lib.prof_symbols()); _lib.prof_enable(False)
            outs.append(y.clone())
            if counters:
                assert int(ws[:4096].view(torch.int32).abs().sum()) == 0, "tile counters not back at zero"
    finally:
        os.environ.pop("DWG_CONV_PATCH_MINM", None)
    assert any("splitk_epilogue" in k for k in launches[0]), launches[0].keys()
    for k in (1, 2, 3):
        assert not any("splitk_epilogue" in s_ for s_ in launches[k]), launches[k].keys()
        assert torch.equal(outs[0], outs[k]), (case, k)


def test_x_rejects_what_the_format_cannot_express():
    """K-strided operands and channel counts that are not whole 8-groups are argument errors, not silent fallbacks."""
    from dreamwaltz_g_amd import gemm
    x = torch.zeros(64, 64, dtype=torch.int32, device="cuda"); y = torch.empty(64, 64, device="cuda")
    with pytest.raises(RuntimeError):
        gemm.gemm_raw(x, x, y, 64, 64, 64, (64, 1), (1, 64), 64)          # B(n, k) strided along k
    with pytest.raises(RuntimeError):
        gemm.gemm_raw(x, x, y, 64, 64, 60, (64, 1), (64, 1), 64)          # K % 8 != 0
