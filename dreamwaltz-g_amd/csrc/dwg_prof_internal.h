// dwg_prof_internal.h -- launch wrapper used by every kernel file: times the launch when profiling is enabled.
#pragma once
#include <hip/hip_runtime.h>

bool dwg_prof_on();
// `symbol` = the kernel's own name (what rocprofv3 lists), `work` = algorithmic flops (or bytes) of this launch.
void dwg_prof_begin(const char* name, const char* symbol, double work, hipStream_t stream, void** token);
void dwg_prof_end(const char* name, hipStream_t stream, void* token);

// Launch failures are tracked per thread and per launch: hipGetLastError() is sticky across libraries (PyTorch leaves benign
// errors behind), so it is cleared right before and read right after every launch of ours.
int& dwg_launch_failed_flag();

#define DWG_LAUNCH_W(NAME, SYMBOL, WORK, KERNEL, GRID, BLOCK, LDS, STREAM, ...)        \
    do {                                                                              \
        void* tok__ = nullptr;                                                        \
        if (dwg_prof_on()) dwg_prof_begin(NAME, SYMBOL, WORK, (STREAM), &tok__);      \
        (void)hipGetLastError();                                                      \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, (STREAM), __VA_ARGS__);          \
        if (hipGetLastError() != hipSuccess) dwg_launch_failed_flag() = 1;            \
        if (tok__) dwg_prof_end(NAME, (STREAM), tok__);                               \
    } while (0)

#define DWG_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...) \
    DWG_LAUNCH_W(NAME, #KERNEL, 0.0, KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__)
