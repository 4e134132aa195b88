"""Shared full-width cases of the SD-1.5 parity tests: seeded weights / inputs and the fp32 CPU-oracle outputs for them.  The oracle runs
of the whole denoiser (~1 minute of CPU arithmetic) and of the whole VAE encoder are computed once per machine and cached under the system
temp directory, so that the fp32 / f32x / fp16 test files compare against the SAME oracle tensors without paying for them three times."""
import os
import tempfile

import torch

_CACHE = os.path.join(tempfile.gettempdir(), "dwg_oracle_cache")


def _cached(name, fn):
    os.makedirs(_CACHE, exist_ok=True)
    path = os.path.join(_CACHE, name + ".pt")
    if os.path.exists(path):
        return torch.load(path)
    out = fn()
    torch.save(out, path + ".tmp")
    os.replace(path + ".tmp", path)
    return out


def denoiser_weights():
    from dreamwaltz_g_amd import sd15
    ucfg = sd15.UNetConfig()
    return ucfg, sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=0), sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=1)


def denoiser_draw(seed):
    """(latents [2,4,64,64] (the CFG pair shares them), text [2,77,768], cond [1,3,512,512], noise [1,4,64,64])"""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, 4, 64, 64, generator=g).repeat(2, 1, 1, 1), torch.randn(2, 77, 768, generator=g),
            torch.rand(1, 3, 512, 512, generator=g), torch.randn(1, 4, 64, 64, generator=g))


def denoiser_oracle(seed=5, t=500):
    """eps [2,4,64,64] of oracle/sd15.py (fp32, CPU) for denoiser_weights() on denoiser_draw(seed) at timestep t."""
    def run():
        from oracle import sd15 as osd
        torch.set_num_threads(min(64, max(1, os.cpu_count() or 1)))
        ucfg, usd, csd = denoiser_weights()
        lat, text, cond, _ = denoiser_draw(seed)
        with torch.no_grad():
            return osd.predict_noise(ucfg, usd, csd, lat, torch.tensor([t]), text, cond.repeat(2, 1, 1, 1))
    return _cached("denoiser_s%d_t%d" % (seed, t), run)


def vae_case():
    """(cfg, state dict, image [1,3,512,512], moment gradient) and the oracle's (moments, image gradient)."""
    from dreamwaltz_g_amd import sd15
    vcfg = sd15.VAEConfig()
    sd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=2)
    g = torch.Generator().manual_seed(6)
    img = torch.rand(1, 3, 512, 512, generator=g)
    gm = torch.randn(1, 8, 64, 64, generator=g)

    def run():
        from oracle import sd15 as osd
        torch.set_num_threads(min(64, max(1, os.cpu_count() or 1)))
        imgr = img.clone().requires_grad_(True)
        ref = osd.vae_encode_moments(vcfg, sd, imgr)
        (gref,) = torch.autograd.grad(ref, imgr, gm)
        return ref.detach(), gref.detach()
    ref, gref = _cached("vae_s2_img6", run)
    return vcfg, sd, img, gm, ref, gref
