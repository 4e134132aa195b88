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

/* The same two calls with `offsets` / `mlp_scales` as COLUMNS of one row-major [n_free, mlp_ld] tensor (the deformation network's packed
 * output [warp 3 | scaling 3 | rotation 4]: avatar.py:1500-1588 slices it; here the slices are never materialised).  mlp_ld = row stride in
 * floats (3: the dense layout of the calls above).  Backward: d_offsets / d_mlp_scales point into ONE [n_free, mlp_ld] gradient tensor, and
 * the `mlp_tail` columns right behind d_mlp_scales (the unused rotation columns) are written as zeros by the same launch. */
int dwg_gaussian_assemble_forward_ld(int32_t n_free, int32_t n_total, const float* positions, const float* offsets, float init_offset,
                                     const float* log_scales, const float* mlp_scales, int32_t mlp_ld, float init_scale,
                                     const float* quaternions, const float* h, float* pos_out, float* scales_out, float* quats_out,
                                     float* colors_out, float* opac_out, dwg_stream_t stream);
int dwg_gaussian_assemble_backward_ld(int32_t n_free, int32_t n_total, float init_offset, const float* log_scales, float init_scale,
                                      const float* quaternions, const float* h, const float* g_pos, const float* g_scales,
                                      const float* g_quats, const float* g_colors, const float* g_opac, float* d_positions,
                                      float* d_offsets, float* d_log_scales, float* d_mlp_scales, int32_t mlp_ld, int32_t mlp_tail,
                                      float* d_quaternions, float* d_h, dwg_stream_t stream);

/* Up to DWG_MAX_SEGMENTS copies of `count` floats each in ONE launch: the row-block merges of the avatar's Gaussian sets
 * (merge_gaussians, /root/reference/core/gaussian/gaussian_utils.py:56-68: one torch.cat per tensor there).  div != 0: every value goes out
 * as (v + add) * (1 / div) -- the grid encoder's input normalisation (core/nerf/gridencoder/grid.py `(inputs + bound) / (2 * bound)`) in the
 * form torch's elementwise kernels evaluate a division by a scalar (one fp32 reciprocal, a multiply per element): the same bits. */
#define DWG_MAX_SEGMENTS 12
typedef struct dwg_segment { void* dst; const void* src; int64_t count; } dwg_segment;
int dwg_copy_segments(int32_t count, const dwg_segment* segs, float add, float div, dwg_stream_t stream);
/* The reverse hand-off: dst[i] += src[i] for every segment, one launch (the segments must not overlap each other). */
int dwg_add_segments(int32_t count, const dwg_segment* segs, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
