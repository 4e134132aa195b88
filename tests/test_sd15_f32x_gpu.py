"""-m gpu parity of the SPLIT-PRECISION ("f32x") SD-1.5 plans against the fp32 PyTorch-CPU oracle (oracle/sd15.py) with the same seeded random
weights.  f32x is how this build reaches the precision the reference runs the guidance stage in (fp32: /root/reference/configs/__init__.py:236,241;
scripts/train_w_expr.sh:56-94 never pass --optim.fp16) at the 16-bit MFMA rate: every value is hi + 2^-11 lo fp16 halves (csrc/dwg_xfmt.h), every
product three v_mfma_f32_32x32x16_f16 with fp32 accumulation.  The bars are fp32-grade, not "16-bit storage" grade:

  * blocks at SD-1.5 widths (ResNet, transformer with the fused split-precision attention at head sizes 40 / 160, VAE down block): 2e-5 rel-L2;
  * whole ControlNet + UNet CFG pass (1.22 G parameters): eps <= 1e-4, SDS gradient under CFG 50 <= 5e-4 vs the CPU oracle (the exact-f32
    plans measure 4e-6 / 1.3e-5 there, the fp16 plans 2e-3 / 6e-3, the bf16 plans 1.5e-2 / 4e-2);
  * whole VAE encoder forward and image gradient (incl. its single-head N = 4096 attention with the format-aware transposes): 1e-4 / 2e-4;
  * f32x == exact-f32 plans ON THE GPU over t in {20, 500, 980} x 3 seeds.

Measured values land in gpurun_out/parity_f32x.json (copied to profiles/r04_parity_f32x.json)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, r):
    a = a.detach().double().cpu(); r = r.detach().double().cpu()
    return float((a - r).norm() / r.norm().clamp_min(1e-30))


def _cos(a, r):
    a = a.detach().double().cpu().reshape(-1); r = r.detach().double().cpu().reshape(-1)
    return float((a @ r) / (a.norm() * r.norm()).clamp_min(1e-30))


def _note(name, **kw):
    path = os.path.join(ROOT, "gpurun_out", "parity_f32x.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[name] = kw
    with open(path, "w") as f:
        json.dump(d, f, indent=1)
    print("[parity-f32x]", name, kw)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _block_plan(sd):
    from dreamwaltz_g_amd import sd15
    dev = torch.device("cuda")
    plan = sd15.Plan(dev, "f32x")
    w = sd15.Weights(sd, dev, "f32x")
    return plan, w, sd15.Builder(plan, w, 32, "t")


def test_f32x_plan_buffers_are_opaque_words_of_the_fp32_size():
    from dreamwaltz_g_amd import sd15, xfmt
    plan = sd15.Plan(torch.device("cuda"), "f32x")
    b = plan.buf(2, 8, 8, 320)
    assert b.dtype == xfmt.DTYPE and b.element_size() == 4 and plan.esize == 4 and plan.dt == 3
    x = torch.randn(2, 8, 8, 320, device="cuda")
    plan.store(b, x)
    assert float((plan.load(b) - x).abs().max() / x.abs().max()) < 5e-7
    # fewer channels than the buffer: padded through the staging tensor, whole 8-groups written
    b8 = plan.buf(2, 4, 4, 8, zero=True); st = plan.stage_like(b8)
    plan.store(b8, x[:, :4, :4, :3], st)
    got = plan.load(b8)
    assert float((got[..., :3] - x[:, :4, :4, :3]).abs().max()) < 1e-6 and float(got[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,hw", [(1280, 1280, 8), (640, 320, 64), (2560, 1280, 16)])
def test_f32x_resnet_block(cin, cout, hw):
    from dreamwaltz_g_amd import sd15
    from oracle import sd15 as osd
    sh = {}
    sd15._resnet_shapes(sh, "r", cin, cout, 1280)
    sd = sd15.random_state_dict(sh, seed=cin + hw)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, cin, hw, hw, generator=g)
    temb = torch.randn(2, 1280, generator=g)
    ref = osd.resnet(x, sd, "r", temb, 32, 1e-5)
    plan, w, b = _block_plan(sd)
    xin = plan.buf(2, hw, hw, cin)
    tb = (torch.nn.functional.linear(torch.nn.functional.silu(temb), sd["r.time_emb_proj.weight"], sd["r.time_emb_proj.bias"])
          + sd["r.conv1.bias"]).cuda().contiguous()
    y = b.resnet(xin, "r", (tb, cout))
    plan.store(xin, _nhwc(x).cuda())
    plan.run_eager()
    e = _rel(plan.load(y).permute(0, 3, 1, 2), ref)
    _note("f32x_resnet_%dto%d_r%d" % (cin, cout, hw), rel_l2=e)
    assert e < 2e-5, e


@pytest.mark.parametrize("c,hw", [(320, 32), (640, 16), (1280, 16), (1280, 8)])
def test_f32x_transformer_block(c, hw):
    """Head sizes 40 / 80 / 160 through the fused split-precision attention (self: hw^2 keys, cross: 77), GEGLU in the projection's epilogue."""
    from dreamwaltz_g_amd import sd15
    from oracle import sd15 as osd
    sh = {}
    sd15._transformer_shapes(sh, "a", c, 768)
    sd = sd15.random_state_dict(sh, seed=c + hw)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, c, hw, hw, generator=g)
    text = torch.randn(2, 77, 768, generator=g)
    ref = osd.transformer(x, sd, "a", text, 8, 32)
    plan, w, b = _block_plan(sd)
    xin = plan.buf(2, hw, hw, c)
    tin = plan.buf(2, 77, 768)
    y = b.transformer(xin, "a", tin, 8)
    plan.store(xin, _nhwc(x).cuda()); plan.store(tin, text.cuda())
    plan.run_eager()
    e = _rel(plan.load(y).permute(0, 3, 1, 2), ref)
    _note("f32x_transformer_c%d_r%d" % (c, hw), rel_l2=e)
    assert e < 2e-5, e


def test_f32x_attention_kernel_alone_long_and_ragged():
    """The fused kernel against softmax(QK^T/sqrt(d))V in float64: 4096 keys at d = 40 (the 64x64 level), ragged 77 keys, d = 80 / 160."""
    import ctypes
    from dreamwaltz_g_amd import _lib, xfmt
    L = _lib.lib()
    worst = {}
    for (Bn, Hh, Nq, Nk, d) in [(1, 8, 4096, 4096, 40), (2, 8, 300, 77, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 77, 160), (1, 2, 64, 64, 160)]:
        g = torch.Generator().manual_seed(Nq + d)
        q = torch.randn(Bn, Nq, Hh * d, generator=g); k = torch.randn(Bn, Nk, Hh * d, generator=g); v = torch.randn(Bn, Nk, Hh * d, generator=g)
        qx, kx, vx = xfmt.pack(q).cuda(), xfmt.pack(k).cuda(), xfmt.pack(v).cuda()
        o = torch.empty(Bn, Nq, Hh * d, device="cuda", dtype=xfmt.DTYPE)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        rc = L.dwg_attention_forward_dt(3, Bn, Hh, Nq, Nk, d, pp(qx), Hh * d, Nq * Hh * d, pp(kx), Hh * d, Nk * Hh * d, pp(vx), Hh * d, Nk * Hh * d,
                                        pp(o), Hh * d, Nq * Hh * d, float(d) ** -0.5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        qh = q.double().view(Bn, Nq, Hh, d).permute(0, 2, 1, 3); kh = k.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
        vh = v.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(Bn, Nq, Hh * d)
        worst["N%d_K%d_d%d" % (Nq, Nk, d)] = _rel(xfmt.unpack(o.cpu()), ref)
    _note("f32x_attention_kernel", **worst)
    assert max(worst.values()) < 5e-6, worst


def test_f32x_attention_with_the_keys_split_over_workgroups():
    """dwg_attention_forward_ws: launches whose query blocks do not fill the chip (the 32x32 / 16x16 levels' self-attention, ragged query
    counts) split the keys over workgroups and merge the ranges in a second launch -- fp32-grade against float64 like the unsplit launch,
    bit-reproducible run to run, and a launch the heuristic leaves alone (many query blocks; 77 keys) asks for no workspace."""
    import ctypes
    from dreamwaltz_g_amd import _lib, xfmt
    L = _lib.lib()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    assert L.dwg_attention_split_workspace_bytes(3, 2, 8, 4096, 4096, 40) == 0         # 512 query blocks: no split
    assert L.dwg_attention_split_workspace_bytes(3, 2, 8, 1024, 77, 80) == 0           # three key tiles: nothing to split
    assert L.dwg_attention_split_workspace_bytes(1, 2, 8, 1024, 1024, 80) == 0         # bf16 unit: not built there
    worst = {}
    for (Bn, Hh, Nq, Nk, d) in [(2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160), (1, 8, 300, 333, 40), (1, 2, 130, 1000, 64), (2, 8, 64, 250, 160)]:
        g = torch.Generator().manual_seed(Nq + Nk + d)
        q = torch.randn(Bn, Nq, Hh * d, generator=g); k = torch.randn(Bn, Nk, Hh * d, generator=g); v = torch.randn(Bn, Nk, Hh * d, generator=g)
        qx, kx, vx = xfmt.pack(q).cuda(), xfmt.pack(k).cuda(), xfmt.pack(v).cuda()
        need = int(L.dwg_attention_split_workspace_bytes(3, Bn, Hh, Nq, Nk, d))
        assert need > 0, (Nq, Nk, d)
        ws = torch.empty(need // 4, device="cuda")
        outs = []
        for rep in range(2):
            o = torch.empty(Bn, Nq, Hh * d, device="cuda", dtype=xfmt.DTYPE)
            ws.fill_(float("nan"))                                          # nothing may be read that this launch did not write
            _lib.prof_enable(True)
            rc = L.dwg_attention_forward_ws(3, Bn, Hh, Nq, Nk, d, pp(qx), Hh * d, Nq * Hh * d, pp(kx), Hh * d, Nk * Hh * d, pp(vx), Hh * d,
                                            Nk * Hh * d, pp(o), Hh * d, Nq * Hh * d, float(d) ** -0.5, pp(ws), need, st())
            assert rc == 0
            torch.cuda.synchronize()
            names = _lib.prof_table(); _lib.prof_enable(False)
            assert "flash_attn_merge" in names, names.keys()                # the split path ran
            outs.append(o)
        assert torch.equal(outs[0], outs[1])
        qh = q.double().view(Bn, Nq, Hh, d).permute(0, 2, 1, 3); kh = k.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
        vh = v.double().view(Bn, Nk, Hh, d).permute(0, 2, 1, 3)
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(Bn, Nq, Hh * d)
        worst["N%d_K%d_d%d" % (Nq, Nk, d)] = _rel(xfmt.unpack(outs[0].cpu()), ref)
        # without the workspace the same entry point runs the unsplit launch
        o1 = torch.empty(Bn, Nq, Hh * d, device="cuda", dtype=xfmt.DTYPE)
        rc = L.dwg_attention_forward_ws(3, Bn, Hh, Nq, Nk, d, pp(qx), Hh * d, Nq * Hh * d, pp(kx), Hh * d, Nk * Hh * d, pp(vx), Hh * d,
                                        Nk * Hh * d, pp(o1), Hh * d, Nq * Hh * d, float(d) ** -0.5, None, 0, st())
        assert rc == 0 and _rel(xfmt.unpack(o1.cpu()), ref) < 5e-6
    _note("f32x_attention_key_split", **worst)
    assert max(worst.values()) < 5e-6, worst


def test_f32x_vae_down_block():
    from dreamwaltz_g_amd import sd15
    from oracle import sd15 as osd
    import torch.nn.functional as F
    sh = {}
    sd15._resnet_shapes(sh, "r", 128, 128, 0)
    sh["d.weight"] = (128, 128, 3, 3); sh["d.bias"] = (128,)
    sd = sd15.random_state_dict(sh, seed=9)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 128, 256, 256, generator=g)
    r = osd.resnet(x, sd, "r", None, 32, 1e-6)
    ref = F.conv2d(F.pad(r, (0, 1, 0, 1)), sd["d.weight"], sd["d.bias"], stride=2)
    plan, w, b = _block_plan(sd)
    xin = plan.buf(1, 256, 256, 128)
    y = b.resnet(xin, "r", None, eps=1e-6)
    z = b.conv(y, "d", stride=2, pad=0, out_hw=(128, 128))
    plan.store(xin, _nhwc(x).cuda())
    plan.run_eager()
    e1, e2 = _rel(plan.load(y).permute(0, 3, 1, 2), r), _rel(plan.load(z).permute(0, 3, 1, 2), ref)
    _note("f32x_vae_down_block_r256", rel_l2_resnet=e1, rel_l2_downsample=e2)
    assert e1 < 2e-5 and e2 < 2e-5, (e1, e2)


def _sds(eps2, noise):
    d = eps2[1] - eps2[0]
    return eps2[0] + 50.0 * d - noise[0], d


@pytest.mark.slow
def test_full_width_denoiser_f32x_vs_oracle_and_vs_the_exact_f32_plans():
    """(1) f32x plan vs the fp32 CPU oracle at t = 500 (whole ControlNet + UNet, CFG batch 2): eps <= 1e-4, SDS gradient <= 5e-4;
    (2) f32x plan vs the exact-f32 plan ON THE GPU at t in {20, 500, 980} x 3 seeds; graph replay == eager bit for bit."""
    from dreamwaltz_g_amd import sd15
    from tests import sd15_cases as cases
    ucfg, usd, csd = cases.denoiser_weights()
    dev = torch.device("cuda")
    px = sd15.DenoiserPlan(ucfg, usd, csd, dev, batch=2, latent_hw=64, dtype="f32x")
    lat, text, cond, noise = cases.denoiser_draw(5)
    t = torch.tensor([500])
    ref = cases.denoiser_oracle(5, 500)
    px.set_inputs(lat.cuda(), t.cuda(), text.cuda(), cond.cuda())
    got = px.run().float().cpu().clone()
    gx, _ = _sds(got, noise); gref, dref = _sds(ref, noise)
    e, eg, cg = _rel(got, ref), _rel(gx, gref), _cos(gx, gref)
    _note("denoiser_f32x_vs_oracle", rel_l2_eps=e, rel_l2_sds_gradients=eg, cosine_sds_gradients=cg,
          cfg_difference_over_eps=float(dref.norm() / ref[0].norm()))
    assert e < 1e-4 and eg < 5e-4 and cg > 0.9999999, (e, eg, cg)
    # hipGraph replay of the same plan: bit-identical
    with torch.cuda.stream(torch.cuda.Stream()):
        px.plan.capture()
        px.set_inputs(lat.cuda(), t.cuda(), text.cuda(), cond.cuda())
        rep = px.run().float().cpu().clone()
    assert torch.equal(rep, got)
    px.plan.use_graph = False
    p32 = sd15.DenoiserPlan(ucfg, usd, csd, dev, batch=2, latent_hw=64, dtype="f32")
    rows = []
    for seed in (5, 6, 7):
        lat, text, cond, noise = cases.denoiser_draw(seed)
        for tt in (20, 500, 980):
            t = torch.tensor([tt])
            outs = []
            for p in (p32, px):
                p.set_inputs(lat.cuda(), t.cuda(), text.cuda(), cond.cuda())
                outs.append(p.run().float().cpu().clone())
            a32, ax = outs
            g32, d32 = _sds(a32, noise); gxx, _ = _sds(ax, noise)
            eps_norm = float(a32[0].double().norm())
            rows.append(dict(seed=seed, t=tt, rel_l2_eps=_rel(ax, a32), rel_l2_sds_gradients=_rel(gxx, g32), cosine_sds_gradients=_cos(gxx, g32),
                             sds_error_over_50_eps=float((gxx - g32).double().norm()) / (50.0 * eps_norm)))
    worst = {k: max(r[k] for r in rows) for k in ("rel_l2_eps", "rel_l2_sds_gradients", "sds_error_over_50_eps")}
    worst["cosine_sds_gradients_min"] = min(r["cosine_sds_gradients"] for r in rows)
    _note("denoiser_f32x_vs_fp32_sweep", rows=rows, worst=worst)
    assert worst["rel_l2_eps"] < 1e-4 and worst["rel_l2_sds_gradients"] < 5e-4 and worst["sds_error_over_50_eps"] < 1e-4, worst


@pytest.mark.slow
def test_full_width_vae_encoder_f32x_vs_oracle():
    from dreamwaltz_g_amd import sd15
    from tests import sd15_cases as cases
    vcfg, sd, img, gm, ref, gref = cases.vae_case()
    dev = torch.device("cuda")
    px = sd15.VAEEncoderPlan(vcfg, sd, dev, image_hw=512, dtype="f32x")
    got = px.encode(img.cuda()).float().cpu().clone()
    gimg = px.backward(gm.cuda()).float().cpu().clone()
    e_f, e_b, c_b = _rel(got, ref), _rel(gimg, gref), _cos(gimg, gref)
    _note("vae_encoder_f32x_vs_oracle", rel_l2_moments=e_f, rel_l2_image_grad=e_b, cosine_image_grad=c_b)
    assert e_f < 1e-4 and e_b < 2e-4 and c_b > 0.9999999, (e_f, e_b, c_b)
    # Gradients of ANY scale (round 5; verdict round 4, item 3): the backward is linear, so it runs on 2^k x the gradient with k chosen on the
    # device (sd15.VAEEncoderPlan.backward: max |g| -> [32, 64]) and the answer is scaled back -- the fp32 reference has no floor at fp16's
    # normal range and neither may this.  Round 4 measured 1.2e-4 at scale 1e-4 (bar 1e-3) without the pre-scale.  Bar per decade: the
    # scale-1 bar, 2e-5 (every decade runs the SAME scaled problem: what may differ is the rounding of the scaled-back result).
    sweep = {}
    for dec in (0, -1, -2, -3, -4, -5, -6, 2):
        sc = 10.0 ** dec
        gs = px.backward((gm * sc).cuda()).float().cpu().clone()
        sweep["1e%d" % dec] = _rel(gs / sc, gref)
    _note("vae_encoder_f32x_gradient_scale_sweep", rel_l2_image_grad_by_scale=sweep)
    assert max(sweep.values()) < 2e-5, sweep
    rep = px.bwd.range_report()
    _note("vae_encoder_f32x_range_report", backward={k: v for k, v in rep.items() if k != "worst"},
          forward={k: v for k, v in px.fwd.range_report().items() if k != "worst"})
    assert rep["saturated"] == 0 and rep["nonfinite"] == 0 and rep["max_abs"] < 65504.0 / 8, rep


@pytest.mark.parametrize("scale", [1.0, 3e-5, 0.0, 7e3])
def test_f32x_vae_boundary_converters_equal_the_torch_statements(scale):
    """dwg_vae_image_pack / dwg_vae_grad_prescale_pack / dwg_vae_dx_unpack (one launch each; sd15.VAEEncoderPlan.encode / backward) against the
    element-wise statements they replace: same values bit for bit -- 2 v - 1, the power-of-two pre-scale chosen from max |g|, the scale back."""
    import ctypes
    from dreamwaltz_g_amd import _lib, xfmt
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(3)
    B, H, h = 2, 48, 6
    img = torch.rand(B, 3, H, H, generator=g).cuda()
    x = torch.full((B, H, H, 8), 12345, dtype=torch.int32, device="cuda")
    _lib.check(L.dwg_vae_image_pack(B, H, H, _lib.ptr(img), _lib.ptr(x), st), "dwg_vae_image_pack")
    want = torch.zeros(B, H, H, 8, device="cuda"); want[..., :3] = (img * 2.0 - 1.0).permute(0, 2, 3, 1)
    assert torch.equal(x, xfmt.pack(want))
    gm = (torch.randn(B, 8, h, h, generator=g) * scale).cuda()
    dst = torch.empty(B, h, h, 8, dtype=torch.int32, device="cuda"); inv = torch.empty(1, device="cuda")
    _lib.check(L.dwg_vae_grad_prescale_pack(B, h * h, _lib.ptr(gm), 64.0, _lib.ptr(dst), _lib.ptr(inv), st), "dwg_vae_grad_prescale_pack")
    amax = gm.abs().amax()
    k = torch.floor(torch.log2(64.0 / amax.clamp_min(1e-30))).clamp(-60.0, 100.0)
    k = torch.where(amax > 0, k, torch.zeros_like(k))
    assert torch.equal(dst, xfmt.pack((gm * torch.exp2(k)).permute(0, 2, 3, 1).contiguous())), (float(k), float(inv))
    assert float(inv) == float(torch.exp2(1.0 - k))
    if scale > 0:
        assert 32.0 <= float(xfmt.unpack(dst).abs().max()) <= 64.0
    _lib.check(L.dwg_vae_grad_prescale_pack(B, h * h, _lib.ptr(gm), 0.0, _lib.ptr(dst), _lib.ptr(inv), st), "dwg_vae_grad_prescale_pack")   # no pre-scale
    assert torch.equal(dst, xfmt.pack(gm.permute(0, 2, 3, 1).contiguous())) and float(inv) == 2.0
    dx = xfmt.pack(torch.randn(B, H, H, 8, generator=g).cuda())
    out = torch.empty(B, 3, H, H, device="cuda"); inv.fill_(0.375)
    _lib.check(L.dwg_vae_dx_unpack(B, H, H, _lib.ptr(dx), _lib.ptr(inv), _lib.ptr(out), st), "dwg_vae_dx_unpack")
    assert torch.equal(out, (xfmt.unpack(dx)[..., :3].permute(0, 3, 1, 2) * inv).contiguous())


def test_f32x_range_report_names_the_layer_that_saturates():
    """Round 5: range telemetry of the f32x plans (sd15.Plan.range_report, csrc/elementwise.hip k_x_range_scan).  A ResNet block fed (a)
    ordinary activations, (b) heavy-tailed ones -- a few channels 1e3 times the rest, as SD-1.5's known outlier channels are -- and (c)
    activations large enough to push a convolution's output past fp16's 65504: (a) and (b) stay fp32-grade against the fp32 oracle with
    nothing saturated; (c) is REPORTED (the stored values sit at +-65504) with the convolution's name, instead of passing silently.  Weights
    spanning 1e-6 .. 1e2 keep the fp32-grade bar too (their packed range is part of the report)."""
    from dreamwaltz_g_amd import sd15
    from oracle import sd15 as osd
    cin, cout, hw = 320, 320, 16
    sh = {}
    sd15._resnet_shapes(sh, "r", cin, cout, 0)
    g = torch.Generator().manual_seed(5)
    sd = sd15.random_state_dict(sh, seed=3)
    # weights over eight decades: every output channel of conv1 gets its own magnitude in 1e-6 .. 1e2 (the norm layer behind it rescales)
    mag = 10.0 ** (torch.rand(cout, generator=g) * 8.0 - 6.0)
    sd["r.conv1.weight"] = sd["r.conv1.weight"] / sd["r.conv1.weight"].abs().amax(dim=(1, 2, 3), keepdim=True) * mag.view(-1, 1, 1, 1)
    x0 = torch.randn(2, cin, hw, hw, generator=g)
    heavy = x0.clone(); heavy[:, ::37] *= 1e3                           # outlier channels
    cases = {"plain": x0, "heavy_tailed": heavy}
    res = {}
    for name, x in cases.items():
        ref = osd.resnet(x, sd, "r", None, 32, 1e-5)
        plan, w, b = _block_plan(sd)
        xin = plan.buf(2, hw, hw, cin)
        y = b.resnet(xin, "r", None)
        plan.store(xin, _nhwc(x).cuda())
        plan.run_eager()
        rep = plan.range_report()
        res[name] = dict(rel_l2=_rel(plan.load(y).permute(0, 3, 1, 2), ref), saturated=rep["saturated"], nonfinite=rep["nonfinite"],
                         subnormal_frac=rep["subnormal"] / max(1, rep["elements"]), max_abs=rep["max_abs"])
        assert rep["saturated"] == 0 and rep["nonfinite"] == 0, (name, rep)
        assert res[name]["rel_l2"] < 2e-5, (name, res[name])
        assert w.range["saturated"] == 0 and w.range["max_abs"] <= 100.0 * 1.0001 and w.range["subnormal"] > 0      # 1e-6-scale weights ARE below 6.1e-5
    # (c) a convolution whose output passes 65504: conv2 behind a norm that cannot tame it because its OWN weights are huge
    sd2 = dict(sd); sd2["r.conv2.weight"] = sd["r.conv2.weight"] * 3e5
    plan, w, b = _block_plan(sd2)
    xin = plan.buf(2, hw, hw, cin)
    y = b.resnet(xin, "r", None)
    plan.store(xin, _nhwc(x0).cuda())
    plan.run_eager()
    rep = plan.range_report()
    res["overflowing_conv2"] = dict(saturated=rep["saturated"], worst=[(d["layer"], d["saturated"]) for d in rep["worst"][:2]])
    _note("f32x_range_report", **res)
    assert rep["saturated"] > 0 and rep["nonfinite"] == 0, rep           # saturates (does not overflow to inf) -- and says so
    assert rep["worst"][0]["layer"] == "r.conv2" and rep["worst"][0]["saturated"] > 0, rep["worst"][:2]
