/* Host build of dreamwaltz-g_amd/csrc/assemble_math.h for CPU-side derivative checks (test infrastructure only). */
#include "../../dreamwaltz-g_amd/csrc/assemble_math.h"

void host_assemble_forward(int n, float init_offset, float init_scale, int fix_opacity, const float* p, const float* off,
                           const float* ls, const float* ms, const float* q, const float* h, float* pos, float* scl, float* qn,
                           float* col, float* op) {
    for (int i = 0; i < n; i++) {
        dwg_assemble_geom(p + 3 * i, off + 3 * i, init_offset, ls + 3 * i, ms + 3 * i, init_scale, q + 4 * i, pos + 3 * i, scl + 3 * i,
                          qn + 4 * i);
        dwg_assemble_color(h + 4 * i, fix_opacity, col + 3 * i, op + i);
    }
}
void host_assemble_backward(int n, float init_offset, float init_scale, int fix_opacity, const float* ls, const float* q,
                            const float* h, const float* gpos, const float* gscl, const float* gqn, const float* gcol,
                            const float* gop, float* dp, float* doff, float* dls, float* dms, float* dq, float* dh) {
    for (int i = 0; i < n; i++) {
        dwg_assemble_geom_bwd(ls + 3 * i, q + 4 * i, init_offset, init_scale, gpos + 3 * i, gscl + 3 * i, gqn + 4 * i, dp + 3 * i,
                              doff + 3 * i, dls + 3 * i, dms + 3 * i, dq + 4 * i);
        dwg_assemble_color_bwd(h + 4 * i, fix_opacity, gcol + 3 * i, gop[i], dh + 4 * i);
    }
}
