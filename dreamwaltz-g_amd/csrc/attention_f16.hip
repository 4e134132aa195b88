// attention_f16.hip -- the fp16-operand unit of the fused attention forward: attention.hip compiled again with _Float16 operands
// (v_mfma_f32_32x32x16_f16).  Exports dwg_attention_forward_f16, reached through dwg_attention_forward_dt(DWG_DTYPE_F16, ...)
// (include/dwg_nn.h).  Serves the fp16-storage plans (the reference's autocast storage type, configs/__init__.py:462).
#define DWG_ATTN_F16_TU 1
#include "attention.hip"
