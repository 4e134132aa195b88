"""-m gpu parity tests of the norm / activation / attention kernels against plain PyTorch fp32 references.
Inputs are bf16-rounded first; outputs are bf16 (2^-8 relative rounding), statistics/softmax in fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16", "f16"])
def half(request):
    """The 16-bit operand types: bf16 (gemm.hip / attention.hip) and fp16 (their _f16 units: the same kernels on _Float16)."""
    return {"bf16": torch.bfloat16, "f16": torch.float16}[request.param]



def _err(a, r):
    a = a.float().cpu(); r = r.float().cpu()
    return float((a - r).abs().max() / r.abs().max().clamp_min(1e-20))


@pytest.mark.parametrize("B,H,C,silu,eps", [(2, 64, 320, True, 1e-5), (2, 16, 1280, False, 1e-6), (1, 96, 128, True, 1e-6),
                                            (2, 8, 2560, True, 1e-5), (1, 33, 64, False, 1e-5), (2, 32, 960, True, 1e-5),
                                            (1, 128, 256, True, 1e-6),    # 128^2 x 8 ch per group: the three-kernel path
                                            # the one-launch small-image kernel (k_gn_small): group bundles of 1 / 2 / 4 groups, 5-15 chunks wide
                                            (2, 32, 640, True, 1e-5), (2, 32, 1280, True, 1e-5), (2, 16, 1920, True, 1e-5), (2, 8, 960, False, 1e-5),
                                            (1, 32, 320, True, 1e-5), (2, 16, 2560, True, 1e-5), (3, 8, 1280, True, 1e-6)])
def test_groupnorm_forward_backward(B, H, C, silu, eps):
    from dreamwaltz_g_amd import nn_ops
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3).bfloat16()
    gamma = torch.randn(C, generator=g) * 0.5 + 1.0; beta = torch.randn(C, generator=g) * 0.2
    xd = x.double().requires_grad_(True)
    ref = F.group_norm(xd, 32, gamma.double(), beta.double(), eps)
    if silu:
        ref = F.silu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    y, stats = nn_ops.groupnorm(xn, gamma.cuda(), beta.cuda(), 32, eps, silu)
    assert _err(y.permute(0, 3, 1, 2), ref) < 1.2e-2
    dy = torch.randn(ref.shape, generator=g).bfloat16()
    (gx,) = torch.autograd.grad(ref, xd, dy.double())
    dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = nn_ops.groupnorm_backward(xn, dyn, stats, gamma.cuda(), beta.cuda(), 32, eps, silu)
    assert _err(dx.permute(0, 3, 1, 2), gx) < 1.5e-2
    # fused skip-connection gradient: dx + residual in the same pass
    res = torch.randn(xn.shape, generator=g).bfloat16().cuda()
    dx2 = nn_ops.groupnorm_backward(xn, dyn, stats, gamma.cuda(), beta.cuda(), 32, eps, silu, residual=res)
    assert _err(dx2.permute(0, 3, 1, 2), gx + res.permute(0, 3, 1, 2).cpu().double()) < 1.5e-2


@pytest.mark.parametrize("M,C", [(8192, 320), (300, 1280), (5, 640)])
def test_layernorm_and_geglu(M, C):
    from dreamwaltz_g_amd import nn_ops
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).bfloat16()
    gamma = torch.randn(C, generator=g) * 0.3 + 1; beta = torch.randn(C, generator=g) * 0.1
    ref = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5)
    y = nn_ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda())
    assert _err(y, ref) < 1.2e-2
    h = torch.randn(M, 2 * C, generator=g).bfloat16()
    a, b = h.double().chunk(2, dim=-1)
    assert _err(nn_ops.geglu(h.cuda()), a * F.gelu(b)) < 1.2e-2


@pytest.mark.parametrize("B,Hh,Nq,Nk,d", [(2, 8, 4096, 4096, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 256, 160), (1, 8, 64, 64, 160),
                                          (2, 4, 200, 77, 40), (1, 2, 130, 33, 16), (2, 8, 256, 77, 160)])
def test_flash_attention(B, Hh, Nq, Nk, d, half):
    from dreamwaltz_g_amd import nn_ops
    g = torch.Generator().manual_seed(Nq + Nk + d)
    q = torch.randn(B, Nq, Hh * d, generator=g).to(half); k = torch.randn(B, Nk, Hh * d, generator=g).to(half)
    v = torch.randn(B, Nk, Hh * d, generator=g).to(half)
    if Nq == 256 and Nk == 256:
        k[:, 5] *= 6.0   # a spiky key: forces large running-max jumps in the online softmax
    sp = lambda t, n: t.double().view(B, n, Hh, d).permute(0, 2, 1, 3)  # noqa: E731
    ref = F.scaled_dot_product_attention(sp(q, Nq), sp(k, Nk), sp(v, Nk)).permute(0, 2, 1, 3).reshape(B, Nq, Hh * d)
    o = nn_ops.attention(q.cuda(), k.cuda(), v.cuda(), Hh)
    assert _err(o, ref) < 2e-2, _err(o, ref)


def test_flash_attention_on_fused_qkv_slices(half):
    """q/k/v as column slices of one [B, N, 3C] projection output (strided rows, no copies)."""
    from dreamwaltz_g_amd import nn_ops
    g = torch.Generator().manual_seed(0)
    B, N, Hh, d = 2, 512, 8, 40
    C = Hh * d
    qkv = torch.randn(B, N, 3 * C, generator=g).to(half)
    sp = lambda t: t.double().view(B, N, Hh, d).permute(0, 2, 1, 3)  # noqa: E731
    ref = F.scaled_dot_product_attention(sp(qkv[..., :C]), sp(qkv[..., C:2 * C]), sp(qkv[..., 2 * C:]))
    ref = ref.permute(0, 2, 1, 3).reshape(B, N, C)
    qc = qkv.cuda()
    o = nn_ops.attention(qc[..., :C], qc[..., C:2 * C], qc[..., 2 * C:], Hh)
    assert _err(o, ref) < 2e-2


def test_softmax_rows_forward_backward():
    from dreamwaltz_g_amd import nn_ops
    g = torch.Generator().manual_seed(2)
    S = torch.randn(300, 4096, generator=g) * 3
    Sd = S.double().requires_grad_(True)
    ref = torch.softmax(Sd * 0.044, -1)
    P = nn_ops.softmax_rows(S.cuda(), 0.044)
    assert _err(P, ref) < 1e-2
    dP = torch.randn(300, 4096, generator=g)
    (gS,) = torch.autograd.grad(ref, Sd, dP.double())
    dS = nn_ops.softmax_rows_backward(P, dP.cuda(), 0.044)
    assert _err(dS, gS) < 2e-2


@pytest.mark.parametrize("B,R,C", [(1, 4096, 512), (2, 72, 200), (3, 8, 8), (1, 512, 4096)])
def test_transpose_2byte(B, R, C, half):
    """out[b][c][r] = in[b][r][c] on 2-byte elements, bit-exact, incl. ragged 64 x 64 tiles and a strided (row-sliced) source."""
    import ctypes
    from dreamwaltz_g_amd import _lib
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(B, R, C + 8, generator=g).to(half).cuda()
    src = x[..., :C]                                   # row stride C + 8
    out = torch.empty(B, C, R, device="cuda", dtype=half)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().dwg_transpose_2byte(B, R, C, _lib.ptr(src), src.stride(1), src.stride(0), _lib.ptr(out), R, C * R, st), "dwg_transpose_2byte")
    assert torch.equal(out, src.transpose(1, 2).contiguous())
