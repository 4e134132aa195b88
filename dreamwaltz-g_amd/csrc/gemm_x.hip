// gemm_x.hip -- the split-precision ("f32x") unit of the MFMA GEMM / implicit-GEMM convolution: gemm.hip compiled a third time for operands in
// the hi / lo fp16 format of dwg_xfmt.h.  fp32-grade products at the 16-bit MFMA rate: a b ~= ah bh + 2^-11 (al bh + ah bl), three
// v_mfma_f32_32x32x16_f16 per product into two fp32 accumulator sets, combined once per tile before the epilogue.  Same tiles, direct-to-LDS
// loaders (the format is MFMA-ready in HBM: no conversion in the k-loop), LDS-patch convolution, split-K slabs and fused epilogues as the
// bf16 unit; the epilogue splits what it stores.  Exports dwg_gemm_x / dwg_gemm_workspace_bytes_x, which dwg_gemm / dwg_gemm_workspace_bytes
// forward to for dtype == DWG_DTYPE_F32X (include/dwg_gemm.h).  Serves the "f32x" denoiser / VAE plans: the precision the reference runs the
// guidance stage in (/root/reference/configs/__init__.py:236,241) at a rate the exact-f32 MFMA (157 TFLOP/s) cannot reach.
#define DWG_GEMM_X_TU 1
#include "gemm.hip"
