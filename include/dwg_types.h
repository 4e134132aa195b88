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
#endif
