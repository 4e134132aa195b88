/*
 * include/dwg_prof.h -- optional per-kernel timing with HIP events recorded on the launch stream.
 * bench.py uses it to report the dominant kernel's average launch duration (roofline.achieved) from inside the timed
 * run; it has no reference counterpart (the reference has no profiler).
 * Disabled by default: zero overhead (no events are created or recorded).
 */
#ifndef DWG_PROF_H
#define DWG_PROF_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif
/* enable != 0 starts recording; enabling also clears previously recorded samples. */
int dwg_prof_enable(int32_t enable);
/* Synchronises the recorded events and returns, for the kernel registered under `name`, the number of launches and the
 * summed duration in milliseconds.  Unknown name -> count 0. */
int dwg_prof_query(const char* name, int64_t* count, double* total_ms);
/* Writes up to `cap` bytes of a newline-separated "name count total_ms" table; returns bytes needed. */
int64_t dwg_prof_dump(char* buf, int64_t cap);
/* Same samples aggregated by KERNEL SYMBOL (the name rocprofv3 --kernel-trace lists, e.g. "k_conv3x3_patch<128>"):
 * newline-separated "symbol<TAB>count<TAB>total_ms<TAB>work", where work is the summed algorithmic flops of those
 * launches (2*M*N*K for the GEMM / conv kernels, 0 where the caller computes bytes itself).  Returns bytes needed. */
int64_t dwg_prof_dump_symbols(char* buf, int64_t cap);
#ifdef __cplusplus
}
#endif
#endif
