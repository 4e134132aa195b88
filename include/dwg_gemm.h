/*
 * include/dwg_gemm.h -- C-ABI of the MFMA GEMM / implicit-GEMM convolution primitive.
 *
 * It is the contraction engine behind three reference seams:
 *   - the per-Gaussian MLPs:   MLP.forward /root/reference/core/nerf/nerf_model.py:28-33,
 *                              DeformNetwork.forward /root/reference/core/deformation/deform_model.py:102-143   (f32)
 *   - the denoiser (boundary B4): ControlNetScoreDistillation._predict /root/reference/core/guidance/controlnet.py:83-114
 *                              and AutoEncoderSD.encode_images /root/reference/core/guidance/vae.py:34-40 -- every
 *                              conv / linear / attention product of the SD-1.5 UNet, ControlNet and VAE encoder (bf16)
 * The reference reaches these through torch.nn.functional (cuDNN / cuBLAS); there is no reference C interface to
 * mirror, so the descriptor below is this library's own.
 *
 *   C[m][n] = epilogue( alpha * sum_k A(m,k) * B(n,k) ),   A(m,k) = A[m*a_row_stride + k*a_k_stride],
 *                                                          B(n,k) = B[n*b_row_stride + k*b_k_stride]
 *   epilogue: + bias (f32, per column or per row) -> activation -> + residual -> store (f32 or bf16) | atomicAdd (split-K)
 * With conv_enabled, A is an NHWC tensor [img, Hin, Win, Cin] read through an im2col view:
 *   m = (img, oy, ox), k = (ky, kx, ci); iy = oy*stride - pad_t + ky (divided by conv_in_dilation when > 1, rows/cols that
 *   do not divide are zero: this is the gather form of a transposed convolution, used for input gradients).
 * Strides are in ELEMENTS.  batch = batch1 * batch2 with independent strides (image x head).
 */
#ifndef DWG_GEMM_H
#define DWG_GEMM_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* DWG_DTYPE_F32 / _BF16 / _F16 / _F32X: dwg_types.h */

#define DWG_GEMM_WS_COUNTERS 4096                              /* tile counters in a workspace header (dwg_gemm_desc::workspace_counters) */
#define DWG_GEMM_WS_HEADER_BYTES (DWG_GEMM_WS_COUNTERS * 4)
#define DWG_ACT_NONE 0
#define DWG_ACT_RELU 1
#define DWG_ACT_LEAKY_RELU 2 /* slope 0.01 (F.leaky_relu default, deform_model.py:121) */
#define DWG_ACT_SILU 3
#define DWG_ACT_GELU 4       /* exact erf form */
#define DWG_ACT_SIGMOID 5
#define DWG_ACT_GEGLU_PAIR 6  /* C[m][f] = (acc[f] + b[f]) * gelu(acc[f+32] + b[f+32]) on a 32-interleaved [hidden|gate] projection;
                                 C has N/2 columns (ldc counts those), N % 64 == 0, no residual / split-K */

typedef struct dwg_gemm_desc {
    const void* A; const void* B; void* C;
    const float* bias;        /* may be NULL */
    const void* residual;     /* may be NULL; same indexing as C with ldr */
    int32_t M, N, K;
    int64_t a_row_stride, a_k_stride, b_row_stride, b_k_stride, ldc, ldr;
    int32_t batch1, batch2;
    int64_t a_batch1_stride, a_batch2_stride, b_batch1_stride, b_batch2_stride, c_batch1_stride, c_batch2_stride,
        r_batch1_stride, r_batch2_stride;
    int32_t dtype;            /* DWG_DTYPE_F32 | DWG_DTYPE_BF16 | DWG_DTYPE_F16 | DWG_DTYPE_F32X of A and B (F16: the fp16-operand unit,
                                 csrc/gemm_f16.hip; F32X: split-precision fp32 operands -- csrc/dwg_xfmt.h, csrc/gemm_x.hip: M, N, K, strides and
                                 conv_cin count LOGICAL elements, every 8 of them 32 bytes; operands must be K-contiguous (k stride 1) with
                                 K, row strides and Cin multiples of 8, else DWG_E_ARG) */
    int32_t out_dtype;        /* DWG_DTYPE_F32 or the operand type */
    int32_t residual_dtype;
    int32_t act;              /* DWG_ACT_* */
    float alpha;
    int32_t bias_per_row;
    int32_t splitk;           /* > 1: split the contraction.  Without `workspace`: atomicAdd into a pre-zeroed f32 C (no
                                 bias/act/residual).  With `workspace` (batch 1): fp32 slabs + a reduce pass that applies the
                                 full epilogue; splitk == 0 then means "let the library choose" (small-M layers). */
    int32_t accumulate;       /* C += result (f32 C only) */
    int32_t conv_enabled, conv_cin, conv_hin, conv_win, conv_hout, conv_wout, conv_kh, conv_kw, conv_stride, conv_pad_t,
        conv_pad_l, conv_in_dilation;
    int32_t conv_in_upsample; /* 2: the conv reads a virtual nearest-neighbour 2x upsampled input (Upsample2D + conv fused) */
    const void* A2;           /* optional second NHWC source: channels [conv_cin1, conv_cin) (torch.cat([h, skip], 1) fused) */
    int32_t conv_cin1;
    int32_t bias_row_div;     /* > 0: row m uses bias row m / bias_row_div (per-image channel bias) */
    int64_t bias_ld;          /* row stride of that bias matrix in elements (0 -> N) */
    void* workspace;          /* optional split-K slab workspace (device), see dwg_gemm_workspace_bytes */
    size_t workspace_bytes;
    int32_t force_register_staging; /* != 0: use the register-staged kernel even where the direct-to-LDS path applies (A/B testing) */
    int32_t workspace_counters; /* != 0: the first DWG_GEMM_WS_HEADER_BYTES of `workspace` are tile-arrival counters that belong to the
                                 library: ZERO when the workspace is handed over, zero again after every dwg_gemm that used it (calls
                                 that share a workspace must not overlap in time).  The slabs follow the header, and the slice that
                                 arrives last at a tile sums the slabs (in slice order: the same bits as the reduce pass) inside the
                                 GEMM kernel -- no reduce launch.  0: the whole workspace is slabs, reduce pass as before. */
    const char* name;         /* optional label for dwg_prof */
} dwg_gemm_desc;

int dwg_gemm(const dwg_gemm_desc* desc, dwg_stream_t stream);

/* Bytes of split-K workspace the library would like for this descriptor (0: no split would be used): the slabs plus
   DWG_GEMM_WS_HEADER_BYTES for the counter header of `workspace_counters`. */
size_t dwg_gemm_workspace_bytes(const dwg_gemm_desc* desc);

#ifdef __cplusplus
}
#endif
#endif
