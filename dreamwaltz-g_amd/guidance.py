"""Mirror of the reference's SDS guidance seam on the HIP denoiser (boundary B4/B5).

  ControlNetScoreDistillation.__call__      /root/reference/core/guidance/basic.py:778-917 (default branch: loss_type 'sds',
                                            CFG with negative text, weight_type 'sjc', guidance_scale 50)
  ._predict(latents, text, cond)            /root/reference/core/guidance/controlnet.py:83-114
  .encode_images(images)  (differentiable)  /root/reference/core/guidance/vae.py:34-40
  SpecifyGradient                           /root/reference/core/guidance/basic.py:213-226
  TimePrioritizedScheduler ('uniform')      /root/reference/core/guidance/time_prior.py:321-352
Device RNG draw order per call is the reference's (checklist Q12): VAE posterior noise -> timestep -> latent noise.
"""
from typing import Dict, Optional

import torch

from . import sd15


class SpecifyGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_tensor, gt_grad):
        ctx.save_for_backward(gt_grad)
        return torch.ones([1], device=input_tensor.device, dtype=input_tensor.dtype)

    @staticmethod
    def backward(ctx, grad_scale):
        (gt_grad,) = ctx.saved_tensors
        return gt_grad * grad_scale, None


class _VAEEncode(torch.autograd.Function):
    """AutoencoderKL.encode(...).latent_dist moments with the frozen encoder inside the autograd graph."""

    @staticmethod
    def forward(ctx, images, plan):
        ctx.plan = plan
        return plan.encode(images).clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.plan.backward(g), None


def sd15_alphas_cumprod(device, n=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).to(device)


class ControlNetScoreDistillation:
    def __init__(self, device, unet_cfg: Optional[sd15.UNetConfig] = None, vae_cfg: Optional[sd15.VAEConfig] = None,
                 unet_sd=None, controlnet_sd=None, vae_sd=None, image_hw=512, guidance_scale=50.0, min_timestep=0.02,
                 max_timestep=0.98, seed=0):
        self.device = device
        self.unet_cfg = unet_cfg or sd15.UNetConfig()
        self.vae_cfg = vae_cfg or sd15.VAEConfig()
        if unet_sd is None:        # random-init weights of the SD-1.5 architecture (no checkpoints offline)
            unet_sd = sd15.random_state_dict(sd15.unet_param_shapes(self.unet_cfg), seed=seed)
        if controlnet_sd is None:
            controlnet_sd = sd15.random_state_dict(sd15.controlnet_param_shapes(self.unet_cfg), seed=seed + 1)
        if vae_sd is None:
            vae_sd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(self.vae_cfg), seed=seed + 2)
        self.image_hw = image_hw
        down = 2 ** (len(self.vae_cfg.block_out_channels) - 1)
        self.latent_hw = image_hw // down
        self.denoiser = sd15.DenoiserPlan(self.unet_cfg, unet_sd, controlnet_sd, device, batch=2, latent_hw=self.latent_hw)
        self.vae = sd15.VAEEncoderPlan(self.vae_cfg, vae_sd, device, image_hw=image_hw)
        self.alphas_cumprod = sd15_alphas_cumprod(device)
        self.num_train_timesteps = 1000
        self.guidance_scale = guidance_scale
        self.min_step = int(self.num_train_timesteps * min_timestep)
        self.max_step = int(self.num_train_timesteps * max_timestep)
        self.vae_scale_factor = down
        self.scaling_factor = self.vae_cfg.scaling_factor

    def plans(self):
        return (self.denoiser.plan, self.vae.fwd, self.vae.bwd)

    def capture_graphs(self):
        for p in self.plans():
            p.capture()

    def set_use_graphs(self, on: bool):
        for p in self.plans():
            p.use_graph = bool(on)

    # -- vae.py:34-40
    def encode_images(self, images: torch.Tensor, posterior_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        moments = _VAEEncode.apply(images, self.vae)
        mean, logvar = moments.chunk(2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        if posterior_noise is None:
            posterior_noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype)      # RNG draw #1
        return (mean + std * posterior_noise) * self.scaling_factor

    # -- controlnet.py:83-114
    @torch.no_grad()
    def _predict(self, latents_model_input, text_embeddings, cond_inputs, timestep):
        """latents [2,4,h,w], text [2,77,768], cond [1,3,8h,8w] float in [0,1] (the PIL->tensor conversion of
        controlnet.py:33-55 belongs to the data layer)."""
        self.denoiser.set_inputs(latents_model_input, timestep, text_embeddings, cond_inputs)
        return self.denoiser.run()

    def get_timestep(self):
        return torch.randint(self.min_step, self.max_step + 1, (1,), dtype=torch.long, device=self.device)   # RNG draw #2

    def __call__(self, inputs: torch.Tensor, text_embeds_dict: Dict[str, torch.Tensor], train_step: int = 0,
                 max_iteration: int = 1, cond_inputs=None, timestep=None, noise=None, posterior_noise=None, **_unused):
        """inputs [1,3,H,W] rendered image in [0,1] (requires grad).  Returns the reference's result dict."""
        if inputs.shape[-1] != self.image_hw or inputs.shape[-2] != self.image_hw:
            inputs = torch.nn.functional.interpolate(inputs, (self.image_hw, self.image_hw), mode="bilinear", align_corners=False)
        latents = self.encode_images(inputs, posterior_noise)
        t = self.get_timestep() if timestep is None else timestep
        with torch.no_grad():
            if noise is None:
                noise = torch.randn_like(latents)                                                  # RNG draw #3
            a = self.alphas_cumprod[t].reshape(-1, 1, 1, 1)
            latents_noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
            text = torch.cat([text_embeds_dict['neg'], text_embeds_dict['text']], dim=0)         # ('neg','text') basic.py:546-600
            pred = self._predict(torch.cat([latents_noisy] * 2), text, cond_inputs, t)
            noise_pred_uncond, noise_pred_text = pred.chunk(2)
            noise_pred = noise_pred_uncond + self.guidance_scale * (noise_pred_text - noise_pred_uncond)
            gradients = noise_pred - noise                                                         # weight_type 'sjc': w = 1
        loss = SpecifyGradient.apply(latents, gradients)
        return {"diffusion_loss": loss, "gradients": gradients, "timestep": t, "latents": latents, "sources": latents_noisy,
                "targets": noise_pred}
