// gemm_f16.hip -- the fp16-operand unit of the MFMA GEMM / implicit-GEMM convolution: gemm.hip compiled again with _Float16 operands
// (v_mfma_f32_32x32x16_f16; same tiles, loaders, direct-to-LDS staging, epilogues).  Exports dwg_gemm_f16 / dwg_gemm_workspace_bytes_f16,
// which dwg_gemm / dwg_gemm_workspace_bytes forward to for dtype == DWG_DTYPE_F16 (include/dwg_gemm.h).  Serves the fp16-storage
// denoiser / VAE plans: the reference's autocast storage type (/root/reference/configs/__init__.py:462, core/trainer.py:844,859).
#define DWG_GEMM_F16_TU 1
#include "gemm.hip"
