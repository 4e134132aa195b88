/*
 * include/dwg_elementwise.h -- C-ABI of the small bandwidth-bound helpers of the hot path.
 *   dwg_act_backward_colsum : activation backward + bias gradient of one Linear layer of the per-Gaussian MLPs
 *                             (autograd of /root/reference/core/nerf/nerf_model.py:28-33 and
 *                              /root/reference/core/deformation/deform_model.py:119-122)
 *   dwg_adam_step           : one fused Adam update over a flat fp32 buffer -- the optimizer step that closes the SDS
 *                             step (/root/reference/core/trainer.py:888-890; torch.optim.Adam groups built at
 *                             /root/reference/core/gaussian/gaussian_optimizer.py:67-93 and core/system/avatar.py:1590-1635)
 * All pointers are device pointers to fp32 buffers.
 */
#ifndef DWG_ELEMENTWISE_H
#define DWG_ELEMENTWISE_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* dz[m][n] = dy[m][n] * act'(y[m][n]) (act = DWG_ACT_* of dwg_gemm.h, derivative taken from the OUTPUT y; y == NULL means
 * identity); colsum[n] += sum_m dz[m][n] (colsum pre-zeroed by the caller, may be NULL; dz may be NULL). N <= 256. */
int dwg_act_backward_colsum(int32_t M, int32_t N, int32_t act, const float* dy, const float* y, float* dz, float* colsum,
                            dwg_stream_t stream);

/* Weight gradient of one Linear layer of the per-Gaussian MLPs: dw[n][k] = sum_m dz[m][n] * x[m][k], N, K <= 64, written to
 * dw with row stride lddw (only the [N,K] block is touched).  Deterministic two-pass reduction; workspace holds
 * dwg_mlp_wgrad_workspace_floats(M) floats. */
size_t dwg_mlp_wgrad_workspace_floats(int32_t M);
int dwg_mlp_wgrad(int32_t M, int32_t N, int32_t K, const float* dz, int32_t lddz, const float* x, int32_t ldx, float* dw,
                  int32_t lddw, float* workspace, dwg_stream_t stream);

/* A whole per-Gaussian MLP in one launch (nerf_model.py:28-33 / deform_model.py:111-143 forward): h_0 = x [M, Kin] (row stride ldx),
 * h_{l+1} = act_l(h_l W_l^T + b_l) for l < nlayers, W_l [widths[l], K_l] with row stride ldw[l] (only the first K_l columns are read:
 * `extra` [n_extra] (device; NULL / 0 = none) is a vector every row of the first layer's input is extended by -- the body pose that
 * deform_model.py:113-115 expands and concatenates: its product with W_0[:, Kin : Kin + n_extra] is the same for all rows and is added to
 * b_0 inside the launch), widths <= 64, Kin and the
 * hidden widths multiples of 8, acts in {NONE, RELU, LEAKY_RELU, SIGMOID}.  out [M, widths[last]] with row stride ldo.  hidden (may be
 * NULL) holds per hidden layer a [M, widths[l]] buffer that receives h_{l+1} (kept for the backward) -- a buffer for EVERY hidden layer or for
 * none (DWG_E_ARG otherwise: whether activations are kept is a compile-time property of the launch).  Inference passes NULL.
 * weights / ldw / biases / widths / acts / hidden are HOST arrays of length nlayers (<= 6). */
int dwg_mlp_chain_forward(int32_t M, int32_t Kin, const float* x, int32_t ldx, int32_t nlayers, const float* const* weights,
                          const int32_t* ldw, const float* const* biases, const int32_t* widths, const int32_t* acts,
                          float* const* hidden, float* out, int32_t ldo, const float* extra, int32_t n_extra, dwg_stream_t stream);

/* The backward of dwg_mlp_chain_forward in one launch + one reduce (nerf_model.py:28-33 / deform_model.py:111-143 under autograd): given
 * dy [M, widths[last]] (row stride lddy), the input x, the kept hidden activations (hidden[l] [M, widths[l]] contiguous, l < nlayers - 1)
 * and, when the last activation is not the identity, the output `out`: dx [M, Kin] (row stride lddx; NULL = not wanted), per layer
 * dw[l] [widths[l], >= K_l] with row stride lddw[l] (columns 0..K_l-1 written) and db[l] [widths[l]] (db or entries may be NULL).
 * `extra` [n_extra] (device; NULL / 0 = none): the vector the caller folded into the first layer's bias through the trailing columns of
 * W_0 -- their gradient dw[0][:, Kin + e] = db_0 * extra[e] is written too.  accumulate (HOST array, may be NULL = all 0): per layer bit 0
 * "dw[l] += " and bit 1 "db[l] += " instead of "=" (the caller's gradient slice already holds other contributions: the parameters' slices
 * of a flat gradient buffer).  Deterministic (fixed summation order, no atomics).
 * workspace: dwg_mlp_chain_backward_workspace_floats(M, nlayers) floats.  Array arguments are HOST arrays of length nlayers (<= 6). */
size_t dwg_mlp_chain_backward_workspace_floats(int32_t M, int32_t nlayers);
int dwg_mlp_chain_backward(int32_t M, int32_t Kin, const float* x, int32_t ldx, int32_t nlayers, const float* const* weights,
                           const int32_t* ldw, const int32_t* widths, const int32_t* acts, const float* const* hidden, const float* out,
                           int32_t ldo, const float* dy, int32_t lddy, float* dx, int32_t lddx, float* const* dw, const int32_t* lddw,
                           float* const* db, const int32_t* accumulate, const float* extra, int32_t n_extra, float* workspace,
                           dwg_stream_t stream);

/* torch.optim.Adam update (amsgrad=False, weight_decay=0) on n contiguous floats, step >= 1 is the 1-based step count;
 * grad is multiplied by grad_scale first (1/world_size after a sum all-reduce).  Buffers must be 16-byte aligned. */
int dwg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, int32_t step, float grad_scale, dwg_stream_t stream);
/* The same update with the per-step scalars read from DEVICE memory: hyper[0] = lr / (1 - beta1^t), hyper[1] = sqrt(1 - beta2^t),
 * hyper[2] = grad_scale (three floats the host refreshes before each replay) -- the form a launch needs inside a captured graph of the
 * whole avatar-side step (step_graph.GraphedTrainStep), where kernel arguments are frozen at capture time but the learning-rate schedule
 * (gaussian_optimizer.py:130-141) and the bias correction move every step. */
int dwg_adam_step_dev(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* hyper, float beta1,
                      float beta2, float eps, dwg_stream_t stream);

/* dwg_adam_step_dev for up to DWG_ADAM_MAX_GROUPS parameter groups in ONE launch (the captured step steps every group of every named
 * optimizer at its end: eight launches of a few microseconds each before): group g updates its n floats with its own betas / eps and the
 * scalars of row `hyper_row` of the device table hyper[rows][4].  Same arithmetic per element as dwg_adam_step_dev. */
#define DWG_ADAM_MAX_GROUPS 16
typedef struct dwg_adam_group {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;
    float beta1, beta2, eps;
    int32_t hyper_row;
} dwg_adam_group;
int dwg_adam_step_groups_dev(int32_t count, const dwg_adam_group* groups, const float* hyper, dwg_stream_t stream);

/* NHWC channel concat (torch.cat([h, skip], 1) of the UNet up blocks): out[r] = [a[r] | b[r]], bf16, Ca % 8 == Cb % 8 == 0. */
int dwg_concat_channels(int64_t rows, int32_t Ca, int32_t Cb, const void* a, const void* b, void* out, dwg_stream_t stream);
/* out = a + b on bf16 buffers (n % 8 == 0): gradient joins of the VAE-encoder backward. */
int dwg_add_bf16(int64_t n, const void* a, const void* b, void* out, dwg_stream_t stream);
/* out[b, 2i+py, 2j+px, :] = s<py><px>[b, i, j, :] -- NHWC bf16, C % 8 == 0.  Assembles the input gradient of a stride-2
 * convolution from its four output-parity classes, each of which is an ordinary stride-1 convolution of the incoming gradient
 * with a 2x2 / 2x1 / 1x2 / 1x1 subset of the taps (no zero-dilated input, a quarter of the multiply-adds). */
int dwg_interleave2x2(int32_t B, int32_t Ho, int32_t Wo, int32_t C, const void* s00, const void* s01, const void* s10, const void* s11,
                      void* out, dwg_stream_t stream);
/* fp32 -> bf16 copy. */
int dwg_cast_f32_to_bf16(int64_t n, const float* src, void* dst, dwg_stream_t stream);
/* The typed passes for the other plan element types (dtype = DWG_DTYPE_BF16 | DWG_DTYPE_F16 | DWG_DTYPE_F32 | DWG_DTYPE_F32X, dwg_types.h).  The two byte
 * movers above (concat, interleave) serve them as they are: pass channel counts in units of 2-byte elements (2 C for fp32). */
int dwg_add_dt(int32_t dtype, int64_t n, const void* a, const void* b, void* out, dwg_stream_t stream);
int dwg_cast_f32_to_dt(int32_t dtype, int64_t n, const float* src, void* dst, dwg_stream_t stream);

/* out[b][c][r] = in[b][r][c] for 2-byte elements (bf16 / fp16): R, C and the strides (elements) multiples of 8, 16-byte aligned bases.
 * The VAE mid-block attention (N = 4096, d = 512) transposes P, dS and the [N, C] projections with it so that every product of its forward
 * and backward is a K-contiguous GEMM (the direct-to-LDS MFMA kernels) instead of a strided one. */
int dwg_transpose_2byte(int32_t batch, int32_t R, int32_t C, const void* in, int64_t ld_in, int64_t batch_stride_in, void* out, int64_t ld_out,
                        int64_t batch_stride_out, dwg_stream_t stream);

/* The split-precision f32x format (DWG_DTYPE_F32X, dwg_types.h; layout: dreamwaltz-g_amd/csrc/dwg_xfmt.h): fp32 <-> hi / lo fp16 planes of a
 * contiguous tensor whose rows are multiples of 8 elements (n % 8 == 0, 16-byte aligned).  Input / output converters of the f32x denoiser / VAE
 * plans (the reference hands fp32 latents, text embeddings and images across boundary B4: core/guidance/controlnet.py:83-114, vae.py:34-40). */
int dwg_xfmt_pack(int64_t n, const float* src, void* dst, dwg_stream_t stream);
int dwg_xfmt_unpack(int64_t n, const void* src, float* dst, dwg_stream_t stream);
/* The f32x VAE encoder's boundary converters, one launch each (AutoencoderKL.encode inside the SDS autograd graph:
 * /root/reference/core/guidance/vae.py:34-40, basic.py:368-372).
 *   dwg_vae_image_pack:          image [B,3,H,W] fp32 in [0,1] -> x [B,H,W,8] f32x, channels 0..2 = 2 v - 1 (VaeImageProcessor.normalize), 3..7 = 0.
 *   dwg_vae_grad_prescale_pack:  d moments [B,8,hw] fp32 -> [B,hw,8] f32x times 2^k, k = floor(log2(target / max|g|)) clamped to [-60, 100]
 *                                (0 when max|g| = 0 or target <= 0): the backward is linear and 16-bit halves need the gradient in their range;
 *                                inv_out[0] = 2 * 2^-k (device scalar; the factor 2 is d(2 v - 1) / dv).
 *   dwg_vae_dx_unpack:           dx [B,H,W,8] f32x, channels 0..2 -> d image [B,3,H,W] fp32, times inv[0]. */
int dwg_vae_image_pack(int32_t B, int32_t H, int32_t W, const float* image_nchw, void* x_xs, dwg_stream_t stream);
int dwg_vae_grad_prescale_pack(int32_t B, int32_t hw, const float* g_nchw, float target, void* dst_xs, float* inv_out, dwg_stream_t stream);
int dwg_vae_dx_unpack(int32_t B, int32_t H, int32_t W, const void* dx_xs, const float* inv, float* out_nchw, dwg_stream_t stream);
/* Range telemetry of a stored f32x tensor (cold path; the hot path saturates and goes subnormal silently): ADDS into counters5 (device,
 * zeroed by the caller)  [0] hi halves at +-65504 (saturated, or on the edge),  [1] non-zero values whose hi half is subnormal or zero
 * (|x| < 6.1e-5: fewer than 22 significand bits),  [2] non-finite values,  [3] max |x| as fp32 bits (atomic max),  [4] elements seen.
 * The reference runs this stage in fp32 (/root/reference/configs/__init__.py:236,241): a non-zero [0] says the f32x plans left its range. */
int dwg_xfmt_range_scan(int64_t n, const void* src, uint64_t* counters5, dwg_stream_t stream);
/* dwg_transpose_2byte for any 2-byte plan type, and for f32x tensors (groups of 8 along c on the way in, along r on the way out). */
int dwg_transpose_dt(int32_t dtype, int32_t batch, int32_t R, int32_t C, const void* in, int64_t ld_in, int64_t batch_stride_in, void* out,
                     int64_t ld_out, int64_t batch_stride_out, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
