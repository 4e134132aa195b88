/*
 * include/dwg_gaussian.h -- per-Gaussian activations and non-rigid composition of DreamWaltzG.animate in one launch per direction
 * (SURVEY.md section 8a rows L11 activations, L13, L16).  Replaces the element-wise PyTorch chains of
 *   DreamWaltzG.non_rigid_transform  /root/reference/core/system/avatar.py:1464-1498 (default flags)
 *   DreamWaltzG.static_mlp_forward   /root/reference/core/system/avatar.py:1283-1290
 *   GaussianModel activations        /root/reference/core/gaussian/gaussian_model.py:25-56
 * Rows [0, n_free) are the free Gaussians (geometry + appearance); rows [n_free, n_total) are mesh-bound Gaussians, which only
 * take their colours from the static MLP and have opacity fixed to 1 (avatar.py:1328-1355).  fp32, dense row-major.
 */
#ifndef DWG_GAUSSIAN_H
#define DWG_GAUSSIAN_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* pos = positions + offsets*init_offset; scales = exp(log_scales) + mlp_scales*init_scale; quats = normalize(quaternions);
 * colors = sigmoid(h[:,1:4]); opacities = sigmoid(h[:,0]) for free rows, 1 for mesh-bound rows.  h = static MLP output [n_total,4]. */
int dwg_gaussian_assemble_forward(int32_t n_free, int32_t n_total, const float* positions, const float* offsets, float init_offset,
                                  const float* log_scales, const float* mlp_scales, float init_scale, const float* quaternions,
                                  const float* h, float* pos_out /*[n_free,3]*/, float* scales_out /*[n_free,3]*/,
                                  float* quats_out /*[n_free,4]*/, float* colors_out /*[n_total,3]*/, float* opac_out /*[n_total,1]*/,
                                  dwg_stream_t stream);

/* Gradients of all differentiable inputs; any incoming gradient may be NULL (zero).  Every output is fully written. */
int dwg_gaussian_assemble_backward(int32_t n_free, int32_t n_total, float init_offset, const float* log_scales, float init_scale,
                                   const float* quaternions, const float* h, const float* g_pos, const float* g_scales,
                                   const float* g_quats, const float* g_colors, const float* g_opac, float* d_positions,
                                   float* d_offsets, float* d_log_scales, float* d_mlp_scales, float* d_quaternions, float* d_h,
                                   dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
