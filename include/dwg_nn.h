/*
 * include/dwg_nn.h -- C-ABI of the non-GEMM layers of the SD-1.5 denoiser / VAE encoder (boundary B4):
 * GroupNorm(+SiLU) forward / input-gradient, LayerNorm, GEGLU, fused multi-head attention, row softmax fwd/bwd.
 * Reference seams they serve: ControlNetScoreDistillation._predict (/root/reference/core/guidance/controlnet.py:83-114)
 * and AutoEncoderSD.encode_images with autograd (/root/reference/core/guidance/vae.py:34-40, basic.py:368-372); the
 * reference reaches this math through diffusers -> torch.nn.functional, so the signatures below are this library's own.
 * Activations are NHWC / token-major bf16 (void* = __bf16 device pointers); affine parameters and statistics are fp32.
 */
#ifndef DWG_NN_H
#define DWG_NN_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* y = [silu]( (x - mean_g) * rstd_g * gamma + beta ), x/y [B, HW, C] bf16, G groups over channels (C % 8 == 0, C % G == 0,
 * G <= 64).  stats [B, G, 2] fp32 receives (sum x, sum x^2) per group -- keep it for dwg_groupnorm_backward. */
int dwg_groupnorm_forward(int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const float* gamma, const float* beta,
                          float eps, int32_t fuse_silu, void* y, float* stats, float* workspace, dwg_stream_t stream);
/* Floats of scratch `workspace` the GroupNorm entry points need (per-workgroup partial sums, reduced in a fixed order so
 * the statistics are bit-reproducible; one buffer may be shared by all calls on a stream). */
size_t dwg_groupnorm_workspace_floats(int32_t B, int32_t G);

/* dx of the above w.r.t. x (affine parameters are frozen: basic.py:347-352). dy/dx bf16 [B,HW,C]; scratch [B,G,2] fp32.
 * residual (bf16 [B,HW,C] or NULL) is added to dx: the skip-connection gradient of a ResNet block, saving a separate add pass. */
int dwg_groupnorm_backward(int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const void* dy, const float* stats,
                           const float* gamma, const float* beta, float eps, int32_t fuse_silu, void* dx, float* scratch,
                           float* workspace, const void* residual, dwg_stream_t stream);

/* LayerNorm over the last dimension: x/y [M, C] bf16, C % 8 == 0, C <= 2048. */
int dwg_layernorm_forward(int32_t M, int32_t C, const void* x, const float* gamma, const float* beta, float eps, void* y,
                          dwg_stream_t stream);

/* GEGLU: out[m, f] = x[m, f] * gelu(x[m, F + f]) (exact erf), x [M, 2F] bf16 -> out [M, F] bf16, F % 8 == 0. */
int dwg_geglu_forward(int64_t M, int32_t F, const void* x, void* out, dwg_stream_t stream);

/* O = softmax(scale * Q K^T) V per (image, head).  Q [B, Nq, H*d], K/V [B, Nk, H*d], O [B, Nq, H*d] bf16 addressed with
 * row strides ld* and per-image strides b* (elements, multiples of 8); head h occupies columns [h*d, (h+1)*d).
 * d % 8 == 0, d <= 160. */
int dwg_attention_forward(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                          const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                          int64_t bo, float scale, dwg_stream_t stream);

/* P = softmax(scale * S) row-wise: S fp32 [rows, n] (row stride lds) -> P bf16 (row stride ldp).  Used for the single-head
 * d = 512 attention of the VAE encoder mid block, whose backward needs P. */
int dwg_softmax_rows_forward(int32_t rows, int32_t n, float scale, const float* S, int64_t lds, void* P, int64_t ldp,
                             dwg_stream_t stream);
/* dS = scale * P * (dP - rowsum(dP * P)): P bf16, dP fp32 -> dS bf16. */
int dwg_softmax_rows_backward(int32_t rows, int32_t n, float scale, const void* P, int64_t ldp, const float* dP, int64_t lddp,
                              void* dS, int64_t ldds, dwg_stream_t stream);

/* The same layers for any activation element type: `dtype` = DWG_DTYPE_BF16 | DWG_DTYPE_F16 | DWG_DTYPE_F32 (dwg_types.h) of every
 * `void*` activation argument; affine parameters, statistics and scratch stay fp32.  The un-suffixed entry points above are the
 * DWG_DTYPE_BF16 case.  fp32 is the precision the reference runs the 3DGS stage in (/root/reference/configs/__init__.py:236,241 --
 * `--optim.fp16` is only passed to the NeRF stages), fp16 is its autocast storage type (configs/__init__.py:462, trainer.py:844,859). */
int dwg_groupnorm_forward_dt(int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const float* gamma,
                             const float* beta, float eps, int32_t fuse_silu, void* y, float* stats, float* workspace,
                             dwg_stream_t stream);
int dwg_groupnorm_backward_dt(int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const void* dy,
                              const float* stats, const float* gamma, const float* beta, float eps, int32_t fuse_silu, void* dx,
                              float* scratch, float* workspace, const void* residual, dwg_stream_t stream);
int dwg_layernorm_forward_dt(int32_t dtype, int32_t M, int32_t C, const void* x, const float* gamma, const float* beta, float eps,
                             void* y, dwg_stream_t stream);
int dwg_geglu_forward_dt(int32_t dtype, int64_t M, int32_t F, const void* x, void* out, dwg_stream_t stream);
/* fused attention: DWG_DTYPE_BF16, DWG_DTYPE_F16 or DWG_DTYPE_F32X (split-precision operands, csrc/attention_x.hip: three 16-bit MFMAs per
 * product, fp32-grade scores / probabilities / outputs); the exact-fp32 plans run attention as QK^T -> row softmax -> PV on dwg_gemm */
int dwg_attention_forward_dt(int32_t dtype, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq,
                             int64_t bq, const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O,
                             int64_t ldo, int64_t bo, float scale, dwg_stream_t stream);
/* The same with an optional workspace (dwg_attention_split_workspace_bytes: 0 when the launch would not use one).  With it, a launch whose
 * query blocks do not fill the chip (self-attention of the 32x32 / 16x16 latent levels: 128 / 32 workgroups) splits the KEYS over workgroups and a
 * second small launch merges the ranges in a fixed order -- run-to-run reproducible; the result differs from the unsplit launch's in rounding
 * order only.  Split-precision (F32X) operands only; other types ignore the workspace.  Nothing in the reference to mirror: diffusers'
 * scaled_dot_product_attention (core/guidance/controlnet.py:98-114 through the UNet's attention processors). */
size_t dwg_attention_split_workspace_bytes(int32_t dtype, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d);
int dwg_attention_forward_ws(int32_t dtype, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq,
                             int64_t bq, const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O,
                             int64_t ldo, int64_t bo, float scale, void* workspace, size_t workspace_bytes, dwg_stream_t stream);
/* P / dS have element type `dtype`; S / dP stay fp32 */
int dwg_softmax_rows_forward_dt(int32_t dtype, int32_t rows, int32_t n, float scale, const float* S, int64_t lds, void* P,
                                int64_t ldp, dwg_stream_t stream);
int dwg_softmax_rows_backward_dt(int32_t dtype, int32_t rows, int32_t n, float scale, const void* P, int64_t ldp, const float* dP,
                                 int64_t lddp, void* dS, int64_t ldds, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
