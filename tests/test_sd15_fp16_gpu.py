"""-m gpu: the fp16-storage denoiser / VAE plans -- the reference's `--guide.dtype fp16` (core/guidance/basic.py:24-27,233: the UNet,
ControlNet and VAE are loaded in torch.float16; configs/__init__.py:236 default 'fp32') and its autocast storage type under `--optim.fp16`
(configs/__init__.py:462).  Same kernels as the bf16 default compiled
for _Float16 operands (csrc/gemm_f16.hip, attention_f16.hip: v_mfma_f32_32x32x16_f16), fp32 accumulation, fp32 norm statistics.

  * single blocks at SD-1.5 widths against the fp32 PyTorch-CPU oracle (oracle/sd15.py): 10 mantissa bits -> tighter than the bf16 bars;
  * whole ControlNet + UNet CFG pass and whole VAE encoder (forward + image gradient) against the fp32 plans ON THE GPU, with the same
    quantities the bf16 trade is reported in (tests/test_sd15_fp32_gpu.py), written to gpurun_out/parity_fp32.json next to them;
  * a whole SDS step through ControlNetScoreDistillation(dtype="f16").
Tolerances are stated per assert; the fp32 plans themselves are pinned to the oracle in tests/test_sd15_fp32_gpu.py."""
import os

import pytest
import torch

from tests.test_sd15_fp32_gpu import _block_plan, _cos, _nhwc, _note, _rel, _sds

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,hw", [(1280, 1280, 8), (640, 320, 64), (2560, 1280, 16)])
def test_fp16_resnet_block(cin, cout, hw):
    from dreamwaltz_g_amd import sd15
    from oracle import sd15 as osd
    sh = {}
    sd15._resnet_shapes(sh, "r", cin, cout, 1280)
    sd = sd15.random_state_dict(sh, seed=cin + hw)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, cin, hw, hw, generator=g)
    temb = torch.randn(2, 1280, generator=g)
    ref = osd.resnet(x, sd, "r", temb, 32, 1e-5)
    plan, w, b = _block_plan(sd, "f16")
    xin = plan.buf(2, hw, hw, cin)
    assert xin.dtype == torch.float16
    tb = (torch.nn.functional.linear(torch.nn.functional.silu(temb), sd["r.time_emb_proj.weight"], sd["r.time_emb_proj.bias"])
          + sd["r.conv1.bias"]).cuda().contiguous()
    y = b.resnet(xin, "r", (tb, cout))
    xin.copy_(_nhwc(x))
    plan.run_eager()
    e = _rel(y.permute(0, 3, 1, 2), ref)
    _note("fp16_resnet_%dto%d_r%d" % (cin, cout, hw), rel_l2=e)
    assert e < 2e-3, e              # fp16 storage of x, weights and 4 intermediate activations: ~2^-11 each


@pytest.mark.parametrize("c,hw", [(320, 32), (1280, 16), (1280, 8)])
def test_fp16_transformer_block(c, hw):
    """Head sizes 40 / 160 through the fp16 unit of the fused attention kernel."""
    from dreamwaltz_g_amd import sd15
    from oracle import sd15 as osd
    sh = {}
    sd15._transformer_shapes(sh, "a", c, 768)
    sd = sd15.random_state_dict(sh, seed=c + hw)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, c, hw, hw, generator=g)
    text = torch.randn(2, 77, 768, generator=g)
    ref = osd.transformer(x, sd, "a", text, 8, 32)
    plan, w, b = _block_plan(sd, "f16")
    xin = plan.buf(2, hw, hw, c)
    tin = plan.buf(2, 77, 768)
    y = b.transformer(xin, "a", tin, 8)
    xin.copy_(_nhwc(x)); tin.copy_(text)
    plan.run_eager()
    e = _rel(y.permute(0, 3, 1, 2), ref)
    _note("fp16_transformer_c%d_r%d" % (c, hw), rel_l2=e)
    assert e < 3e-3, e


@pytest.mark.slow
def test_full_width_denoiser_fp16_vs_fp32_plan():
    """fp16 plan vs fp32 plan ON THE GPU at t in {20, 500, 980} x 3 seeds -- the sweep the bf16 default is reported on."""
    from dreamwaltz_g_amd import sd15
    ucfg = sd15.UNetConfig()
    usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=0)
    csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=1)
    dev = torch.device("cuda")
    p32 = sd15.DenoiserPlan(ucfg, usd, csd, dev, batch=2, latent_hw=64, dtype="f32")
    p16 = sd15.DenoiserPlan(ucfg, usd, csd, dev, batch=2, latent_hw=64, dtype="f16")

    def draw(seed):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(1, 4, 64, 64, generator=g).repeat(2, 1, 1, 1), torch.randn(2, 77, 768, generator=g),
                torch.rand(1, 3, 512, 512, generator=g), torch.randn(1, 4, 64, 64, generator=g))

    rows = []
    for seed in (5, 6, 7):
        lat, text, cond, noise = draw(seed)
        for tt in (20, 500, 980):
            t = torch.tensor([tt])
            outs = []
            for p in (p32, p16):
                p.set_inputs(lat.cuda(), t.cuda(), text.cuda(), cond.cuda())
                outs.append(p.run().float().cpu().clone())
            a32, a16 = outs
            assert torch.isfinite(a16).all()
            g32, d32 = _sds(a32, noise); g16, _ = _sds(a16, noise)
            eps_norm = float(a32[0].double().norm())
            rows.append(dict(seed=seed, t=tt, rel_l2_eps=_rel(a16, a32), rel_l2_sds_gradients=_rel(g16, g32), cosine_sds_gradients=_cos(g16, g32),
                             sds_error_over_50_eps=float((g16 - g32).double().norm()) / (50.0 * eps_norm)))
    worst = {k: max(r[k] for r in rows) for k in ("rel_l2_eps", "rel_l2_sds_gradients", "sds_error_over_50_eps")}
    worst["cosine_sds_gradients_min"] = min(r["cosine_sds_gradients"] for r in rows)
    _note("denoiser_fp16_vs_fp32_sweep", rows=rows, worst=worst)
    # 8x finer storage than bf16 (whose bars are 3e-2 / 4e-2 / 0.09 / 0.996 on the same sweep)
    assert worst["rel_l2_eps"] < 5e-3, worst
    assert worst["sds_error_over_50_eps"] < 6e-3, worst
    assert worst["rel_l2_sds_gradients"] < 1.5e-2 and worst["cosine_sds_gradients_min"] > 0.9999, worst


@pytest.mark.slow
def test_full_width_vae_encoder_fp16_vs_fp32_plan():
    from dreamwaltz_g_amd import sd15
    vcfg = sd15.VAEConfig()
    sd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=2)
    g = torch.Generator().manual_seed(6)
    img = torch.rand(1, 3, 512, 512, generator=g)
    dev = torch.device("cuda")
    p32 = sd15.VAEEncoderPlan(vcfg, sd, dev, image_hw=512, dtype="f32")
    got = p32.encode(img.cuda()).float().cpu().clone()
    gm = torch.randn(got.shape, generator=g)
    gimg = p32.backward(gm.cuda()).float().cpu().clone()
    p16 = sd15.VAEEncoderPlan(vcfg, sd, dev, image_hw=512, dtype="f16")
    got16 = p16.encode(img.cuda()).float().cpu().clone()
    g16 = p16.backward(gm.cuda()).float().cpu().clone()
    assert torch.isfinite(got16).all() and torch.isfinite(g16).all()
    t_f, t_b, t_c = _rel(got16, got), _rel(g16, gimg), _cos(g16, gimg)
    _note("vae_encoder_fp16_vs_fp32", rel_l2_moments=t_f, rel_l2_image_grad=t_b, cosine_image_grad=t_c)
    assert t_f < 4e-3 and t_b < 6e-3 and t_c > 0.99998, (t_f, t_b, t_c)     # bf16 bars: 2.5e-2 / 3.5e-2 / 0.9995


def test_sds_guidance_call_fp16_reduced_width():
    """A whole guidance call (VAE encode in autograd -> add noise -> ControlNet + UNet CFG -> SpecifyGradient -> backward to the image)
    with fp16 plans against the same call with fp32 plans: reduced width, same weights, timestep and noises."""
    from dreamwaltz_g_amd import guidance, sd15
    from tests.test_guidance_gpu import _small
    ucfg, vcfg = _small()
    usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=1)
    csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=2)
    vsd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    img = torch.rand(1, 3, 128, 128, generator=g)
    text = torch.randn(2, 77, ucfg.cross_dim, generator=g)
    cond = torch.rand(1, 3, 128, 128, generator=g)
    noise = torch.randn(1, 4, 16, 16, generator=g); vnoise = torch.randn(1, 4, 16, 16, generator=g)
    outs = {}
    for dt in ("f32", "f16"):
        gd = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, usd, csd, vsd, image_hw=128, dtype=dt)
        ic = img.cuda().requires_grad_(True)
        out = gd(ic, {"neg": text[:1].cuda(), "text": text[1:].cuda()}, cond_inputs=cond.cuda(), timestep=torch.tensor([500], device=dev),
                 noise=noise.cuda(), posterior_noise=vnoise.cuda())
        out["diffusion_loss"].backward()
        assert torch.isfinite(ic.grad).all()
        outs[dt] = (out["gradients"].float().cpu().clone(), ic.grad.float().cpu().clone())
    eg, ei = _rel(outs["f16"][0], outs["f32"][0]), _rel(outs["f16"][1], outs["f32"][1])
    _note("sds_guidance_fp16_vs_fp32_reduced_width", rel_l2_latent_gradients=eg, rel_l2_image_gradients=ei)
    assert eg < 2e-2 and ei < 3e-2, (eg, ei)           # the same call with bf16 plans: 1.5e-1 / 2e-1 (tests/test_guidance_gpu.py)
