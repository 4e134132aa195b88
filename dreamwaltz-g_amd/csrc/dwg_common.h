// dwg_common.h -- shared helpers for the gfx950 (CDNA4, wave64) kernels of libdwg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dwg_types.h"

#define DWG_TILE 16
#define DWG_WAVE 64

int& dwg_launch_failed_flag();
#define DWG_RETURN_IF_LAUNCH_FAILED()                    \
    do {                                                 \
        int& f__ = dwg_launch_failed_flag();             \
        if (f__) { f__ = 0; return DWG_E_LAUNCH; }       \
    } while (0)

static inline size_t dwg_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int dwg_cdiv(int a, int b) { return (a + b - 1) / b; }

#if defined(__HIPCC__)
// ---- wave64 reductions through DPP (no LDS traffic). Total ends up in lane 63. ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dwg_dpp_add(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(r);
}
__device__ __forceinline__ float dwg_wave_sum_to_lane63(float v) {
    v = dwg_dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dwg_dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dwg_dpp_add<0x114, 0xF>(v);  // row_shr:4
    v = dwg_dpp_add<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
    v = dwg_dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
    v = dwg_dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = wave total
    return v;
}
__device__ __forceinline__ float dwg_wave_sum_all(float v) {
    v = dwg_wave_sum_to_lane63(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int dwg_lane() { return (int)(threadIdx.x & 63); }
// erf for the GELU epilogues of the bf16 layers: Abramowitz-Stegun 7.1.26, branch-free, one exp + one rcp + 6 fma.  Absolute
// error 6e-7 in fp32 (libm's erff costs ~3x the instructions with both branches executed under divergence); the results are
// rounded to bf16 (relative 4e-3) right after, and the fp32 MLPs of the avatar do not use GELU.
__device__ __forceinline__ float dwg_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));      // v_rcp_f32 (HIP's __frcp_rn is an out-of-line call: stack + clobbers)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float y = 1.f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
#endif
