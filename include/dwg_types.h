/* include/dwg_types.h -- error codes and opaque handle types shared by every dwg_*.h C-ABI header. */
#ifndef DWG_TYPES_H
#define DWG_TYPES_H
#include <stddef.h>
#include <stdint.h>

#define DWG_OK 0
#define DWG_E_ARG (-1)      /* bad argument (NULL pointer, size out of range, inconsistent options) */
#define DWG_E_LAUNCH (-2)   /* HIP runtime reported an error for a launch / memset */
#define DWG_E_CAPACITY (-3) /* caller-provided workspace too small */

typedef void* dwg_stream_t; /* hipStream_t */

/* element types of activation / weight tensors (the `dtype` argument of the *_dt entry points and of dwg_gemm_desc) */
#define DWG_DTYPE_F32 0  /* the reference's GS-stage precision (configs/__init__.py:236,241): exact-f32 MFMA */
#define DWG_DTYPE_BF16 1 /* default plans: bf16 storage, fp32 accumulation */
#define DWG_DTYPE_F16 2  /* the reference's --optim.fp16 storage (configs/__init__.py:462): fp16 storage, fp32 accumulation */
#endif
