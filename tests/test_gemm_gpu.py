"""-m gpu parity tests of the MFMA GEMM / implicit-GEMM conv primitive against plain PyTorch fp32 references.
f32 path (v_mfma_f32_32x32x2_f32) is an exact-f32 fma chain: tolerance 1e-5 relative.  bf16 path: inputs are rounded
to bf16 first so the only difference to the fp32 reference is accumulation order: 2e-3 relative to the output scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16", "f16"])
def half(request):
    """The 16-bit operand types: bf16 (gemm.hip / attention.hip) and fp16 (their _f16 units: the same kernels on _Float16)."""
    return {"bf16": torch.bfloat16, "f16": torch.float16}[request.param]



def _rel(a, r):
    return float((a.float() - r.float()).abs().max() / r.float().abs().max().clamp_min(1e-20))


@pytest.mark.parametrize("M,N,K,act", [(1000, 64, 32, "relu"), (257, 64, 95, "leaky_relu"), (4096, 4, 64, None),
                                      (130, 200, 52, "sigmoid"), (1, 1, 1, None), (300, 10, 64, None)])
def test_f32_linear(M, N, K, act):
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = {None: lambda t: t, "relu": torch.relu, "leaky_relu": torch.nn.functional.leaky_relu, "sigmoid": torch.sigmoid}[act](ref)
    y = gemm.linear(x.cuda(), w.cuda(), b.cuda(), act=act)
    assert _rel(y.cpu(), ref) < 1e-5


def test_f32_transposed_operands_and_splitk_wgrad():
    """dW[n,k] = sum_m dY[m,n] X[m,k] : both operands are row(m)-major -> reduction index is the slow one; split-K atomics."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(0)
    M, N, K = 20000, 64, 32
    dY = torch.randn(M, N, generator=g); X = torch.randn(M, K, generator=g)
    ref = dY.double().t() @ X.double()
    dYc, Xc = dY.cuda(), X.cuda()
    out = torch.zeros(N, K, device="cuda")
    gemm.gemm_raw(dYc, Xc, out, N, K, M, (1, N), (1, K), K, splitk=16)
    assert _rel(out.cpu(), ref) < 1e-4
    # dgrad: dX[m,k] = sum_n dY[m,n] W[n,k]  (B operand is k-major: b_row_stride = 1)
    W = torch.randn(N, K, generator=g)
    out2 = torch.empty(M, K, device="cuda")
    gemm.gemm_raw(dYc, W.cuda(), out2, M, K, N, (N, 1), (1, K), K)
    assert _rel(out2.cpu(), dY.double() @ W.double()) < 1e-5
    # accumulate
    gemm.gemm_raw(dYc, W.cuda(), out2, M, K, N, (N, 1), (1, K), K, accumulate=True)
    assert _rel(out2.cpu(), 2 * (dY.double() @ W.double())) < 1e-5


@pytest.mark.parametrize("M,N,K", [(8192, 320, 320), (2048, 1280, 640), (154, 768, 320), (100, 40, 4096)])
def test_bf16_linear(M, N, K, half):
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(half); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(half)
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g).to(half)
    ref = torch.nn.functional.silu(x.double() @ w.double().t() + b.double()) + r.double()
    y = gemm.linear(x.cuda(), w.cuda(), b.cuda(), act="silu", residual=r.cuda(), out_dtype=torch.float32)
    assert _rel(y.cpu(), ref) < 2e-3
    yb = gemm.linear(x.cuda(), w.cuda(), b.cuda(), act="silu", residual=r.cuda())
    assert yb.dtype == half and _rel(yb.cpu(), ref) < 1.2e-2   # + bf16 output rounding (2^-8)


def test_bf16_batched_attention_products(half):
    """QK^T and PV for [B, N, heads*d] projections addressed in place through (image, head) batch strides."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(1)
    Bn, Nq, Nk, Hh, d = 2, 256, 77, 8, 40
    q = torch.randn(Bn, Nq, Hh * d, generator=g).to(half); k = torch.randn(Bn, Nk, Hh * d, generator=g).to(half)
    v = torch.randn(Bn, Nk, Hh * d, generator=g).to(half)
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    S = torch.empty(Bn, Hh, Nq, Nk, device="cuda", dtype=torch.float32)
    gemm.gemm_raw(qc, kc, S, Nq, Nk, d, (Hh * d, 1), (Hh * d, 1), Nk, alpha=d ** -0.5, batch=(Bn, Hh),
                  a_batch=(Nq * Hh * d, d), b_batch=(Nk * Hh * d, d), c_batch=(Hh * Nq * Nk, Nq * Nk))
    qh = q.double().view(Bn, Nq, Hh, d).permute(0, 2, 1, 3); kh = k.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
    vh = v.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
    Sref = qh @ kh.transpose(-1, -2) * d ** -0.5
    assert _rel(S.cpu(), Sref) < 2e-3
    P = torch.softmax(S, -1).to(half)
    Pp = torch.zeros(Bn, Hh, Nq, 80, device="cuda", dtype=half)   # key dim padded to a multiple of 8
    Pp[..., :Nk] = P
    O = torch.empty(Bn, Nq, Hh * d, device="cuda", dtype=half)
    gemm.gemm_raw(Pp, vc, O, Nq, d, Nk, (80, 1), (1, Hh * d), Hh * d, batch=(Bn, Hh), a_batch=(Hh * Nq * 80, Nq * 80),
                  b_batch=(Nk * Hh * d, d), c_batch=(Nq * Hh * d, d))
    Oref = (P.double().cpu() @ vh).permute(0, 2, 1, 3).reshape(Bn, Nq, Hh * d)
    assert _rel(O.cpu(), Oref) < 1.2e-2


@pytest.mark.parametrize("Cin,Cout,H,stride,k", [(8, 320, 64, 1, 3), (320, 320, 32, 2, 3), (64, 128, 17, 1, 3), (640, 320, 16, 1, 1),
                                                 (128, 128, 64, 2, 3)])
def test_bf16_conv_forward(Cin, Cout, H, stride, k, half):
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(Cin + Cout)
    Bn = 2
    x = torch.randn(Bn, Cin, H, H, generator=g).to(half); w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(half)
    b = torch.randn(Cout, generator=g)
    pad = k // 2
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    y = gemm.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda(),
                         stride=stride, pad=(pad, pad), out_dtype=torch.float32)
    assert _rel(y.permute(0, 3, 1, 2).cpu(), ref) < 2e-3


def test_bf16_conv_asymmetric_pad_and_input_gradient(half):
    """VAE down-sampling conv: F.pad (0,1,0,1) + 3x3 stride 2 pad 0, and the input-gradient of a conv as the dilated
    gather form (transposed convolution)."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(3)
    Bn, Cin, Cout, H = 1, 16, 24, 20
    x = torch.randn(Bn, Cin, H, H, generator=g).to(half); w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 12).to(half)
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xd, (0, 1, 0, 1)), w.double(), stride=2)
    y = gemm.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), None, stride=2,
                         pad=(0, 0), out_hw=(H // 2, H // 2), out_dtype=torch.float32)
    assert _rel(y.permute(0, 3, 1, 2).cpu(), ref) < 2e-3
    gy = torch.randn(ref.shape, generator=g).to(half)
    (gx_ref,) = torch.autograd.grad(ref, xd, gy.double())
    # dgrad: flipped, channel-transposed weights [Cin, KH, KW, Cout]; stride 1, pad KH-1-pad_t, input dilation = stride
    wf = w.flip(2, 3).permute(1, 2, 3, 0).contiguous()
    gx = gemm.conv2d_nhwc(gy.permute(0, 2, 3, 1).contiguous().cuda(), wf.cuda(), None, stride=1, pad=(2, 2), out_hw=(H, H),
                          in_dilation=2, out_dtype=torch.float32)
    assert _rel(gx.permute(0, 3, 1, 2).cpu(), gx_ref) < 2e-3
    # stride-1 symmetric conv input gradient
    ref1 = torch.nn.functional.conv2d(xd, w.double(), padding=1)
    gy1 = torch.randn(ref1.shape, generator=g).to(half)
    (gx1_ref,) = torch.autograd.grad(ref1, xd, gy1.double())
    gx1 = gemm.conv2d_nhwc(gy1.permute(0, 2, 3, 1).contiguous().cuda(), wf.cuda(), None, stride=1, pad=(1, 1), out_dtype=torch.float32)
    assert _rel(gx1.permute(0, 3, 1, 2).cpu(), gx1_ref) < 2e-3


def test_bf16_splitk_slab_epilogue_small_m(half):
    """Small-M layers (8x8 / 16x16 latents): library-chosen split-K with fp32 slabs + fused epilogue pass."""
    import ctypes
    from dreamwaltz_g_amd import gemm, _lib
    g = torch.Generator().manual_seed(4)
    M, N, K = 128, 1280, 11520
    x = torch.randn(M, K, generator=g).to(half); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(half)
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g).to(half)
    ref = torch.nn.functional.silu(x.double() @ w.double().t() + b.double()) + r.double()
    xc, wc, bc, rc_ = x.cuda(), w.cuda(), b.cuda(), r.cuda()
    y = torch.empty(M, N, device="cuda", dtype=half)
    d = gemm.gemm_raw(xc, wc, y, M, N, K, (K, 1), (K, 1), N, bias=bc, residual=rc_, ldr=N, act="silu", run=False)
    d.splitk = 0
    need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
    assert need > 0
    ws = torch.empty(need // 4, device="cuda")
    d.workspace, d.workspace_bytes = ws.data_ptr(), need
    gemm.run_desc(d, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert _rel(y.cpu(), ref) < 1.2e-2


def test_bf16_geglu_pair_epilogue(half):
    """Fused GEGLU: hidden * gelu(gate) formed in the projection's epilogue from a 32-interleaved [hidden|gate] weight."""
    from dreamwaltz_g_amd import gemm
    g = torch.Generator().manual_seed(8)
    M, C = 1000, 320
    x = torch.randn(M, C, generator=g).to(half)
    w = (torch.randn(8 * C, C, generator=g) / C ** 0.5).to(half); b = torch.randn(8 * C, generator=g) * 0.1
    h = x.double() @ w.double().t() + b.double()
    hid, gate = h.chunk(2, dim=-1)
    ref = hid * torch.nn.functional.gelu(gate)
    F_ = 4 * C
    idx = torch.arange(F_).view(-1, 32)
    perm = torch.cat([idx, idx + F_], dim=1).reshape(-1)
    wp, bp = w[perm].contiguous().cuda(), b[perm].contiguous().cuda()
    y = torch.empty(M, F_, device="cuda", dtype=half)
    gemm.gemm_raw(x.cuda(), wp, y, M, 8 * C, C, (C, 1), (C, 1), F_, bias=bp, act="geglu_pair")
    assert _rel(y.cpu(), ref) < 1.2e-2


@pytest.mark.parametrize("C,H,patch_min_m", [(640, 32, 512), (1280, 16, 512), (640, 32, 8192), (320, 64, 512)])
def test_bf16_conv3x3_small_latents_splitk_paths(C, H, patch_min_m, half):
    """3x3 / stride 1 / pad 1 convolutions of the 32x32 and 16x16 latent levels with the library-chosen split-K: through the
    LDS-patch kernel with split-K over the 64-channel slabs (DWG_CONV_PATCH_MINM=512) and through the im2col loader (default)."""
    import ctypes
    import os
    from dreamwaltz_g_amd import gemm, _lib
    g = torch.Generator().manual_seed(C + H)
    Bn = 2
    x = torch.randn(Bn, C, H, H, generator=g).to(half); w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(half)
    b = torch.randn(C, generator=g); r = torch.randn(Bn, H, H, C, generator=g).to(half)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + r.double()
    xc, wc = x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.empty(Bn, H, H, C, device="cuda", dtype=half)
    M, K = Bn * H * H, 9 * C
    os.environ["DWG_CONV_PATCH_MINM"] = str(patch_min_m)
    try:
        bc, rcu = b.cuda(), r.cuda()      # the descriptor holds raw pointers: these must outlive the launch
        d = gemm.gemm_raw(xc, wc, y, M, C, K, (0, 1), (K, 1), C, bias=bc, residual=rcu, ldr=C,
                          conv=(C, H, H, H, H, 3, 3, 1, 1, 1, 1), run=False)
        d.splitk = 0
        need = _lib.lib().dwg_gemm_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(need, 16) // 4, device="cuda")
        if need > 0:
            d.workspace, d.workspace_bytes = ws.data_ptr(), need
        else:
            d.splitk = 1
        _lib.prof_enable(True)
        gemm.run_desc(d, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        syms = _lib.prof_symbols(); _lib.prof_enable(False)
    finally:
        os.environ.pop("DWG_CONV_PATCH_MINM", None)
    used_patch = any(k.startswith("k_conv3x3_patch") for k in syms)
    assert used_patch == (M >= patch_min_m), (syms.keys(), M, patch_min_m)
    assert _rel(y.cpu(), ref) < 1.2e-2
