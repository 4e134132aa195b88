"""oracle/sd15.py -- plain PyTorch (CPU, fp32/fp64) restatement of the SD-1.5 UNet, ControlNet and VAE encoder that the
reference runs through `diffusers` (call sites /root/reference/core/guidance/controlnet.py:98-114, vae.py:34-40).
TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: diffusers==0.24.0 (requirements.txt:4; install.sh:27 pulls git HEAD) and the HF weights are not
available here; the layer graph follows the published SD-1.5 architecture [3P-memory] (UNet2DConditionModel with
CrossAttnDownBlock2D x3 + DownBlock2D, UNetMidBlock2DCrossAttn, mirrored up blocks; ControlNetModel = the same encoder +
conditioning embedding + zero convs; AutoencoderKL encoder with a single-head mid attention) and reads a diffusers-format
state_dict, so real checkpoints can be dropped in when they become available.
"""
import math

import torch
import torch.nn.functional as F


def _gn(x, sd, name, groups, eps):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def _conv(x, sd, name, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _lin(x, sd, name, bias=True):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias else None)


def timestep_embedding(t, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def resnet(x, sd, pre, temb, groups, eps):
    h = _conv(F.silu(_gn(x, sd, pre + ".norm1", groups, eps)), sd, pre + ".conv1")
    if temb is not None:
        h = h + _lin(F.silu(temb), sd, pre + ".time_emb_proj")[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, pre + ".norm2", groups, eps)), sd, pre + ".conv2")
    if pre + ".conv_shortcut.weight" in sd:
        x = _conv(x, sd, pre + ".conv_shortcut", padding=0)
    return x + h


def _attn(q, k, v, heads):
    B, Nq, C = q.shape
    d = C // heads
    sp = lambda t: t.view(B, -1, heads, d).permute(0, 2, 1, 3)  # noqa: E731
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return o.permute(0, 2, 1, 3).reshape(B, Nq, C)


def transformer(x, sd, pre, text, heads, groups):
    B, C, H, W = x.shape
    res = x
    h = _conv(_gn(x, sd, pre + ".norm", groups, 1e-6), sd, pre + ".proj_in", padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = pre + ".transformer_blocks.0"
    ln = lambda v, n: F.layer_norm(v, (C,), sd[t + n + ".weight"], sd[t + n + ".bias"], 1e-5)  # noqa: E731
    n1 = ln(h, ".norm1")
    a = _attn(_lin(n1, sd, t + ".attn1.to_q", False), _lin(n1, sd, t + ".attn1.to_k", False), _lin(n1, sd, t + ".attn1.to_v", False), heads)
    h = _lin(a, sd, t + ".attn1.to_out.0") + h
    n2 = ln(h, ".norm2")
    a = _attn(_lin(n2, sd, t + ".attn2.to_q", False), _lin(text, sd, t + ".attn2.to_k", False), _lin(text, sd, t + ".attn2.to_v", False), heads)
    h = _lin(a, sd, t + ".attn2.to_out.0") + h
    n3 = ln(h, ".norm3")
    f = _lin(n3, sd, t + ".ff.net.0.proj")
    hs, gate = f.chunk(2, dim=-1)
    h = _lin(hs * F.gelu(gate), sd, t + ".ff.net.2") + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _conv(h, sd, pre + ".proj_out", padding=0) + res


def _encoder(cfg, sd, x, temb, text, hint=None):
    h = _conv(x, sd, "conv_in")
    if hint is not None:
        h = h + hint
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet(h, sd, "down_blocks.%d.resnets.%d" % (i, j), temb, cfg.groups, 1e-5)
            if cfg.attn_blocks[i]:
                h = transformer(h, sd, "down_blocks.%d.attentions.%d" % (i, j), text, cfg.heads, cfg.groups)
            skips.append(h)
        if i != nb - 1:
            h = _conv(h, sd, "down_blocks.%d.downsamplers.0.conv" % i, stride=2)
            skips.append(h)
    h = resnet(h, sd, "mid_block.resnets.0", temb, cfg.groups, 1e-5)
    h = transformer(h, sd, "mid_block.attentions.0", text, cfg.heads, cfg.groups)
    h = resnet(h, sd, "mid_block.resnets.1", temb, cfg.groups, 1e-5)
    return skips, h


def _temb(cfg, sd, t, B, dtype):
    te = timestep_embedding(t.reshape(-1).expand(B), cfg.block_out_channels[0]).to(dtype)
    return _lin(F.silu(_lin(te, sd, "time_embedding.linear_1")), sd, "time_embedding.linear_2")


def controlnet_forward(cfg, sd, x, t, text, cond):
    """x [B,4,h,w], cond [B,3,8h,8w] in [0,1] -> (12 down residuals, mid residual)."""
    temb = _temb(cfg, sd, t, x.shape[0], x.dtype)
    e = "controlnet_cond_embedding"
    hnt = F.silu(_conv(cond, sd, e + ".conv_in"))
    for k in range(2 * (len(cfg.cond_channels) - 1)):
        hnt = F.silu(_conv(hnt, sd, "%s.blocks.%d" % (e, k), stride=2 if k % 2 == 1 else 1))
    hnt = _conv(hnt, sd, e + ".conv_out")
    skips, mid = _encoder(cfg, sd, x, temb, text, hint=hnt)
    down = [_conv(s, sd, "controlnet_down_blocks.%d" % k, padding=0) for k, s in enumerate(skips)]
    return down, _conv(mid, sd, "controlnet_mid_block", padding=0)


def unet_forward(cfg, sd, x, t, text, down_res=None, mid_res=None):
    temb = _temb(cfg, sd, t, x.shape[0], x.dtype)
    skips, h = _encoder(cfg, sd, x, temb, text)
    if down_res is not None:
        skips = [s + r for s, r in zip(skips, down_res)]
        h = h + mid_res
    nb = len(cfg.block_out_channels)
    rev_attn = list(reversed(cfg.attn_blocks))
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(h, sd, "up_blocks.%d.resnets.%d" % (i, j), temb, cfg.groups, 1e-5)
            if rev_attn[i]:
                h = transformer(h, sd, "up_blocks.%d.attentions.%d" % (i, j), text, cfg.heads, cfg.groups)
        if i != nb - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, "up_blocks.%d.upsamplers.0.conv" % i)
    return _conv(F.silu(_gn(h, sd, "conv_norm_out", cfg.groups, 1e-5)), sd, "conv_out")


def predict_noise(cfg, unet_sd, cn_sd, latents, t, text, cond):
    """ControlNetScoreDistillation._predict (controlnet.py:83-114) with conditioning_scale = 1."""
    down, mid = controlnet_forward(cfg, cn_sd, latents, t, text, cond)
    return unet_forward(cfg, unet_sd, latents, t, text, down, mid)


def vae_encode_moments(cfg, sd, image01):
    """AutoencoderKL.encode(2x-1) -> moments [B, 8, h, w] (mean | logvar)."""
    g = cfg.groups
    x = image01 * 2.0 - 1.0
    h = _conv(x, sd, "encoder.conv_in")
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet(h, sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), None, g, 1e-6)
        if i != nb - 1:
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, stride=2, padding=0)
    h = resnet(h, sd, "encoder.mid_block.resnets.0", None, g, 1e-6)
    a = "encoder.mid_block.attentions.0"
    B, C, H, W = h.shape
    n = _gn(h, sd, a + ".group_norm", g, 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
    o = _attn(_lin(n, sd, a + ".to_q"), _lin(n, sd, a + ".to_k"), _lin(n, sd, a + ".to_v"), 1)
    h = _lin(o, sd, a + ".to_out.0").reshape(B, H, W, C).permute(0, 3, 1, 2) + h
    h = resnet(h, sd, "encoder.mid_block.resnets.1", None, g, 1e-6)
    h = _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", g, 1e-6)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def sample_latents(moments, noise, scaling_factor=0.18215):
    """DiagonalGaussianDistribution.sample() * scaling_factor (vae.py:39-40)."""
    mean, logvar = moments.chunk(2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return (mean + std * noise) * scaling_factor


def sd15_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """scaled-linear beta schedule of SD-1.5."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def sds_step(ucfg, vcfg, unet_sd, cn_sd, vae_sd, image01, text_neg_pos, cond, t, noise, vae_noise, guidance_scale=50.0):
    """BasicScoreDistillation.__call__ default branch (basic.py:778-917): returns (gradients [1,4,h,w], d loss / d image).
    image01 [1,3,H,W] requires grad; text_neg_pos [2,77,768] = (neg, text)."""
    moments = vae_encode_moments(vcfg, vae_sd, image01)
    latents = sample_latents(moments, vae_noise, vcfg.scaling_factor)
    ac = sd15_alphas_cumprod().to(latents.dtype)
    with torch.no_grad():
        a = ac[int(t)]
        noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
        tt = torch.tensor([int(t)])
        pred = predict_noise(ucfg, unet_sd, cn_sd, torch.cat([noisy] * 2), tt, text_neg_pos, torch.cat([cond] * 2))
        un, tx = pred.chunk(2)
        noise_pred = un + guidance_scale * (tx - un)
        gradients = noise_pred - noise                       # weight_type 'sjc' -> w = 1
    (g_img,) = torch.autograd.grad(latents, image01, gradients)   # SpecifyGradient: d loss / d latents = gradients
    return gradients, g_img
