/* dwg_sds.h -- the latent algebra around the denoiser call of one SDS step (boundary B4), one launch per reference statement group:
 *   posterior sample    /root/reference/core/guidance/vae.py:34-40  (AutoencoderKL.encode(...).latent_dist.sample() * scaling_factor:
 *                       DiagonalGaussianDistribution: mean, logvar = moments.chunk(2, 1); logvar clamped to [-30, 20]; std = exp(0.5 logvar))
 *   add_noise           DDPMScheduler.add_noise as /root/reference/core/guidance/basic.py:833-835 uses it
 *   SDS gradient        /root/reference/core/guidance/basic.py:602-646 ('sds' branch: classifier-free combination, minus the noise,
 *                       weight by timestep, nan_to_num)
 * All tensors fp32, NCHW-contiguous; V = views (images) per call; n = elements per image of a 4-channel latent (4 h w).
 * Error codes: dwg_types.h. */
#ifndef DWG_SDS_H
#define DWG_SDS_H
#include <stddef.h>
#include <stdint.h>
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* latents[v] = (mean + exp(0.5 clamp(logvar, -30, 20)) * noise[v]) * scale, moments [V, 8, h w] = [mean (4 channels) | logvar (4)]. */
int dwg_sds_posterior_sample(int32_t V, int64_t n, const float* moments, const float* noise, float scale, float* latents,
                             dwg_stream_t stream);
/* g_moments [V, 8, h w] from g_latents [V, 4, h w]: d mean = g scale; d logvar = g scale noise 0.5 std where the clamp passes, else 0. */
int dwg_sds_posterior_sample_backward(int32_t V, int64_t n, const float* moments, const float* noise, float scale, const float* g_latents,
                                      float* g_moments, dwg_stream_t stream);
/* out[v] = sqrt(a) latents[v] + sqrt(1 - a) noise[v], a = alphas_cumprod[timestep[v]] (timestep: int64 on the device, 0 <= t < n_steps). */
int dwg_sds_add_noise(int32_t V, int64_t n, const float* latents, const float* noise, const float* alphas_cumprod, int32_t n_steps,
                      const int64_t* timestep, float* out, dwg_stream_t stream);
#define DWG_SDS_WEIGHT_NONE 0        /* 'sjc' / None */
#define DWG_SDS_WEIGHT_DREAMFUSION 1 /* 1 - a */
#define DWG_SDS_WEIGHT_LATENT_NERF 2 /* (1 - a) sqrt(a) */
#define DWG_SDS_WEIGHT_ISM 3         /* sqrt((1 - a) / a) */
/* eps [2 V, n]: the V unconditional predictions, then the V text-conditioned ones.  noise_pred[v] = u + guidance_scale (t - u);
 * gradients[v] = w(a) (noise_pred[v] - noise[v]), torch.nan_to_num'ed when nan_to_num != 0 (nan -> 0, +-inf -> +-FLT_MAX).
 * noise_pred may be NULL. */
int dwg_sds_gradient(int32_t V, int64_t n, const float* eps, const float* noise, const float* alphas_cumprod, int32_t n_steps,
                     const int64_t* timestep, float guidance_scale, int32_t weight_type, int32_t nan_to_num, float* gradients,
                     float* noise_pred, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
