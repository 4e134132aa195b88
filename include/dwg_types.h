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
#define DWG_DTYPE_F32X 3 /* split precision: an fp32 value kept as hi + 2^-11 lo fp16 halves, 32 bytes per 8 channels (16 B of hi, 16 B of lo:
                            dreamwaltz-g_amd/csrc/dwg_xfmt.h); products from three 16-bit MFMAs, fp32 accumulation -- fp32-grade results
                            (the reference's GS-stage precision) at the 16-bit MFMA rate.  Strides / indices of such tensors count LOGICAL
                            (4-byte) elements and must be multiples of 8 */
#endif
