/*
 * include/dwg_gridenc.h -- C-ABI of the multi-resolution grid encoder (boundary B2).
 *
 * Replaces the pybind backend `_gridencoder` of the reference:
 *   prototypes  /root/reference/core/nerf/gridencoder/src/gridencoder.h:12-14
 *   bindings    /root/reference/core/nerf/gridencoder/src/bindings.cpp:5-9
 *   caller      /root/reference/core/nerf/gridencoder/grid.py:51-58 (forward), :79-86 (backward)
 * Argument order and meaning follow grid_encode_forward / grid_encode_backward one for one; the extra trailing
 * `*_layout` selects the feature layout: 0 = [L,B,C] exactly as the reference backend writes it, 1 = [B,L*C]
 * (what grid.py produces after its permute+reshape, written directly with no extra copy).
 * Specialised for D = 3, C = 2, L <= 32 (the avatar's encoder is D=3, C=2, L=16: nerf_model.py:223-231).
 * Inputs outside [0,1] produce zero features and zero gradients (gridencoder.cu:110-135,272-278).
 * All pointers are device pointers; buffers are caller-allocated; grad_embeddings must be pre-zeroed (accumulated).
 */
#ifndef DWG_GRIDENC_H
#define DWG_GRIDENC_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

int dwg_grid_encode_forward(const float* inputs /*[B,D] in [0,1]*/, const float* embeddings /*[sO,C]*/,
                            const int32_t* offsets /*[L+1]*/, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                            uint32_t L, float S /*log2(per_level_scale)*/, uint32_t H /*base resolution*/,
                            float* dy_dx /*[B,L*D*C] or NULL*/, uint32_t gridtype /*0 hash, 1 tiled*/,
                            uint32_t align_corners, uint32_t interp /*0 linear, 1 smoothstep*/, uint32_t out_layout,
                            dwg_stream_t stream);

int dwg_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                             float* grad_embeddings /*[sO,C] accumulated, may be NULL*/, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, const float* dy_dx /*or NULL*/,
                             float* grad_inputs /*[B,D] or NULL (iff dy_dx NULL)*/, uint32_t gridtype,
                             uint32_t align_corners, uint32_t interp, uint32_t grad_layout,
                             const int32_t* host_offsets /*[L+1] HOST copy of `offsets` or NULL: enables the LDS-privatised
                                                           table-gradient path for the coarse levels*/,
                             dwg_stream_t stream);

/* Same, with XCD-private accumulation of the table gradient (MI355X: 8 XCDs, each with its own L2).  Device-scope float
 * atomics are executed at the memory side on this chip; here every XCD adds into its own copy of the table with L2-local
 * (workgroup-scope) atomics and one extra pass sums the 8 copies into grad_embeddings (+=) and clears them.
 * xcd_scratch: [8][offsets[L]*C] fp32, 16-byte aligned, ALL ZERO on entry (it is left all zero on return, so one allocation
 * serves every call); host_offsets is required. */
int dwg_grid_encode_backward_xcd(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                 float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                 uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, float* xcd_scratch,
                                 dwg_stream_t stream);

/* Same result through XCD-OWNED table slabs (the default for large batches): each 128-byte line of grad_embeddings belongs to
 * one XCD (line index mod 8); every chunk of 256 (point, level) lanes is visited once per XCD and a workgroup only adds to the
 * lines its own XCD (HW_REG_XCC_ID) owns, with L2-local (workgroup-scope) atomics.  No private copies, no reduce pass: the
 * traffic is the table lines actually touched.  grad_embeddings must be 128-byte aligned and is accumulated into;
 * xcd_counters: 16 uint32 of scratch (cleared inside: [0..7] work hand-out per XCD, [8..15] set to 1 by every XCD that took
 * part -- all eight must be 1 afterwards, which the Python binding verifies in its self-test); host_offsets is required. */
int dwg_grid_encode_backward_owner(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                   float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                   uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, uint32_t* xcd_counters,
                                   dwg_stream_t stream);

/* The same gradients with the table part BINNED instead of scattered with atomics (gridenc.hip "slab-binned"): the contributions are
 * sorted by 4096-entry slab of the table into `workspace` (dwg_grid_backward_slabs_workspace_bytes: B * L * 8 records of 16 bytes +
 * small tables), accumulated per slab in LDS and written with plain stores.  grad_embeddings must be ZERO on entry (it is overwritten
 * slab-wise, not accumulated into, except for the coarse levels and the few oversubscribed slabs); host_offsets is required. */
size_t dwg_grid_backward_slabs_workspace_bytes(uint32_t B, uint32_t L, uint32_t total_entries);
int dwg_grid_encode_backward_slabs(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                   float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                   uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, void* workspace,
                                   size_t workspace_bytes, dwg_stream_t stream);
/* The same pass ADDING into grad_embeddings instead of overwriting it: the caller's buffer already holds other contributions -- the table's
 * slice of the flat gradient buffer that a multi-view step accumulates several backward passes into (no zeroed temporary, no separate
 * add pass over the 50 MB table).  Slabs owned by one workgroup do a plain read-add-write of their 16-byte pieces. */
int dwg_grid_encode_backward_slabs_accumulate(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                   float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                   uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, void* workspace,
                                   size_t workspace_bytes, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
