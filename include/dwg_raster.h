/*
 * include/dwg_raster.h -- C-ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * Drop-in boundary B1: replaces the third-party CUDA extension
 * `diff_gaussian_rasterization` that the reference imports at
 *   /root/reference/core/gaussian/gaussian_renderer.py:5   (GaussianRasterizationSettings, GaussianRasterizer)
 * and calls at gaussian_renderer.py:60-70 (settings) and :186-195 (forward).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - all tensors are dense fp32 row-major, exactly the layouts the reference hands to its extension:
 *      means3D [G,3], colors_precomp [G,3] | shs [G,M,3], opacities [G] (the reference's [G,1]),
 *      scales [G,3], rotations [G,4] (real-first, used UN-normalised), cov3D_precomp [G,6],
 *      viewmatrix / projmatrix [4,4] in the reference's row-vector form
 *      (gaussian_renderer.py:38-39: viewmatrix = extrinsic^T, projmatrix = viewmatrix @ projection^T);
 *  - outputs: color [3,H,W], depth [1,H,W], alpha [1,H,W], radii [G] int32;
 *  - nothing is allocated inside; the caller supplies workspaces sized by dwg_raster_workspace_sizes();
 *  - every launch goes to the caller's hipStream_t; no host synchronisation happens inside
 *    (the number of (Gaussian,tile) pairs is left in the geometry workspace, see dwg_raster_num_pairs_ptr);
 *  - return value: 0 on success, negative DWG_E_* on error (bad arguments, launch failure).
 */
#ifndef DWG_RASTER_H
#define DWG_RASTER_H

#include "dwg_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors GaussianRasterizationSettings (kwargs at gaussian_renderer.py:43-64). */
typedef struct dwg_raster_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;      /* active SH degree (0..3); ignored when colors_precomp is given */
    int32_t sh_coeffs;      /* M: coefficients per Gaussian in `shs` (>= (sh_degree+1)^2) */
    int32_t prefiltered;    /* accepted for API parity; no effect */
    int32_t debug;          /* accepted for API parity; no effect */
    const float* bg;         /* [3]  device */
    const float* viewmatrix; /* [16] device */
    const float* projmatrix; /* [16] device */
    const float* campos;     /* [3]  device (only read for SH colours) */
    const int32_t* visit_order; /* [G] device or NULL: a PERMUTATION of 0..G-1, the order in which the binning stages walk the
                                 * Gaussians.  Results do not depend on it (every per-block list is sorted by (depth, index)).
                                 * Since round 5 the binning granule is the 32x32-pixel supertile, whose workgroup-private
                                 * histograms merge their global atomics in index order as well: NULL (index order, coalesced
                                 * reads of the five input arrays) is the default; a spatially coherent order is still honoured. */
    const float* tanfov;        /* [2] device or NULL: {tanfovx, tanfovy} read by the kernels INSTEAD of the two host scalars above (which
                                 * must still be positive).  With it -- and viewmatrix / projmatrix / campos being device pointers anyway --
                                 * nothing about the camera is a kernel argument: a step captured into a graph follows a camera that is
                                 * re-sampled every step (/root/reference/data/camera/__init__.py:124-165) by refreshing the device block.
                                 * Per frame at camera_stride like the matrices (dwg_raster_frames). */
} dwg_raster_settings;

/* Several frames per launch chain (round 5): frame f reads per-Gaussian input row g at [f * gaussian_stride + g] of every input array
 * (0: all frames share one set of Gaussians, e.g. several cameras on one pose) and its camera at viewmatrix / projmatrix / campos +
 * f * camera_stride floats (0: one camera; 35 for the [viewmatrix 16 | projmatrix 16 | campos 3] blocks dwg_raster_camera_setup writes).
 * Workspaces and outputs are frame-major: frame f owns bytes [f * size, (f + 1) * size) of each workspace (sizes from
 * dwg_raster_workspace_sizes for ONE frame), radii [F, G], color [F, 3, H, W], depth / alpha [F, 1, H, W].  Every frame of a call has
 * the same Gaussian count, image size, field of view and pair capacity.  The semantics of one frame are those of the single-frame
 * call (/root/reference/core/gaussian/gaussian_renderer.py:186-195 once per frame), bit for bit. */
typedef struct dwg_raster_frames {
    int32_t num_frames;
    int64_t gaussian_stride;
    int64_t camera_stride;
} dwg_raster_frames;

/* Byte sizes of the three caller-owned workspaces.
 *  geom : per-Gaussian splat records + per-tile counters (kept from forward to backward)
 *  pairs: (Gaussian,tile) pair keys + depth-sorted id lists, for `pair_capacity` pairs
 *  image: per-pixel final transmittance + contributor count (kept from forward to backward) */
int dwg_raster_workspace_sizes(int32_t num_gaussians, int32_t image_height, int32_t image_width,
                               int64_t pair_capacity, size_t* geom_bytes, size_t* pairs_bytes,
                               size_t* image_bytes);

/* Device address (inside the geometry workspace) of the int32 header written by stage A / B:
 *   [0] K     pairs (Gaussian, 8x8 block) after exact culling = what the pair workspace must hold
 *   [1] overflow flag (stage B found pair_capacity < K: the frame is truncated and must be redone)
 *   [2] K_ref  sum over Gaussians of the 16x16 reference tiles their 3-sigma square touches (the K of SURVEY 8d's byte formula)
 *   [3] number of backward segments
 *   [9] (Gaussian, 32x32 supertile) pairs = keys sorted this frame. */
const int32_t* dwg_raster_num_pairs_ptr(const void* ws_geom);

/* viewmatrix = extrinsic^T, projmatrix = viewmatrix @ projection^T, campos = c2w[:3,3] exactly as
 * GaussianRenderer.build_gaussian_rasterizer forms them (gaussian_renderer.py:38-41), in one launch:
 * out35 = [viewmatrix 16 | projmatrix 16 | campos 3].  All pointers device, row-major 4x4. */
int dwg_raster_camera_setup(const float* extrinsic, const float* projection, const float* c2w, float* out35,
                            dwg_stream_t stream);

/* dwg_raster_camera_setup plus the field of view, all in device memory: out37 = [viewmatrix 16 | projmatrix 16 | campos 3 | tanfovx |
 * tanfovy] from device extrinsic / projection / c2w [4,4] and device tanfovy [1] (+ tanfovx [1] or NULL: = tanfovy, the square image of
 * gaussian_renderer.py:28-29).  out37 + 35 is what dwg_raster_settings::tanfov points at. */
int dwg_raster_camera_block(const float* extrinsic, const float* projection, const float* c2w, const float* tanfovy,
                            const float* tanfovx, float* out37, dwg_stream_t stream);

/* Stage A: project, build splat records, count the exact-culled (Gaussian, 8x8 block) pairs per Gaussian and the Gaussians per 32x32
 * supertile, scan. */
int dwg_raster_forward_bin(const dwg_raster_settings* cfg, int32_t num_gaussians,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp, int32_t* radii, void* ws_geom,
                           dwg_stream_t stream);

/* Stage B: one (depth, id) key per (Gaussian, supertile), one sort per 32x32-pixel supertile, stable split into the supertile's sixteen
 * 8x8-block lists, front-to-back compositing per block. */
int dwg_raster_forward_render(const dwg_raster_settings* cfg, int32_t num_gaussians, void* ws_geom,
                              void* ws_pairs, int64_t pair_capacity, void* ws_image,
                              float* out_color, float* out_depth, float* out_alpha,
                              dwg_stream_t stream);

/* The two forward stages for `frames->num_frames` frames per launch (see dwg_raster_frames; frames == NULL: one frame). */
int dwg_raster_forward_bin_frames(const dwg_raster_settings* cfg, const dwg_raster_frames* frames, int32_t num_gaussians,
                                  const float* means3D, const float* shs, const float* colors_precomp,
                                  const float* opacities, const float* scales, const float* rotations,
                                  const float* cov3D_precomp, int32_t* radii, void* ws_geom, dwg_stream_t stream);
int dwg_raster_forward_render_frames(const dwg_raster_settings* cfg, const dwg_raster_frames* frames, int32_t num_gaussians,
                                     void* ws_geom, void* ws_pairs, int64_t pair_capacity, void* ws_image,
                                     float* out_color, float* out_depth, float* out_alpha, dwg_stream_t stream);

/* Backward of the whole rasterizer. dL_dout_depth / dL_dout_alpha may be NULL (treated as zero).
 * No float atomics: every (Gaussian, block) pair's partial gradients go to the pair's own 48-byte row of a pair-ordered buffer inside
 * ws_pairs (a Gaussian's rows are contiguous), and ws_grad receives each Gaussian's rows summed in row order -- the same bits on every
 * run.  The rows' frame tag is drawn on the device by the forward of the same frame, so the three launches may be replayed from a
 * captured graph.  Gradient buffers are OVERWRITTEN (not accumulated). Any of dL_dshs / dL_dcolors / dL_dscales /
 * dL_drotations / dL_dcov3D may be NULL when the corresponding input was not supplied.
 * ws_geom / ws_pairs / ws_image must be the ones the forward of the same frame filled. */
int dwg_raster_backward(const dwg_raster_settings* cfg, int32_t num_gaussians,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, const float* rotations,
                        const float* cov3D_precomp, const void* ws_geom, const void* ws_pairs,
                        int64_t pair_capacity /* same value as in forward_render */,
                        const void* ws_image, void* ws_grad /* >= 12*G floats scratch */,
                        const float* dL_dout_color, const float* dL_dout_depth,
                        const float* dL_dout_alpha, float* dL_dmeans3D, float* dL_dmeans2D,
                        float* dL_dshs, float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                        float* dL_drotations, float* dL_dcov3D, dwg_stream_t stream);

/* The backward of `frames->num_frames` frames per launch (frames == NULL: one frame): the three launches run on (work, F) grids.
 * Workspaces are the frame-major ones the forward_*_frames calls filled (same pair_capacity); inputs and cameras lie at the strides of
 * `frames` exactly as in the forward; ws_grad >= F * 12 * G floats; image gradients are [F, 3, H, W] / [F, 1, H, W]; every gradient
 * output is frame-major [F, G, ...] (the caller sums over frames where the frames share their Gaussians).  Frame f's rows are bit for
 * bit those of a single-frame dwg_raster_backward on that frame (the call of gaussian_renderer.py:186-195 differentiated once per view). */
int dwg_raster_backward_frames(const dwg_raster_settings* cfg, const dwg_raster_frames* frames, int32_t num_gaussians,
                               const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, const float* rotations,
                               const float* cov3D_precomp, const void* ws_geom, const void* ws_pairs,
                               int64_t pair_capacity, const void* ws_image, void* ws_grad,
                               const float* dL_dout_color, const float* dL_dout_depth,
                               const float* dL_dout_alpha, float* dL_dmeans3D, float* dL_dmeans2D,
                               float* dL_dshs, float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                               float* dL_drotations, float* dL_dcov3D, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DWG_RASTER_H */
