/* Host build of dreamwaltz-g_amd/csrc/lbs_math.h for CPU-side derivative checks (test infrastructure only). */
#include "../../dreamwaltz-g_amd/csrc/lbs_math.h"

void host_lbs_apply(int n, const float* T12, const float* p, const float* q, float* pout, float* qout) {
    for (int i = 0; i < n; i++) dwg_lbs_apply(T12 + 12 * i, p + 3 * i, q + 4 * i, pout + 3 * i, qout + 4 * i);
}
void host_lbs_apply_bwd(int n, const float* T12, const float* p, const float* q, const float* gpout, const float* gqout,
                        float* gp, float* gq, float* gT12) {
    for (int i = 0; i < n; i++)
        dwg_lbs_apply_bwd(T12 + 12 * i, p + 3 * i, q + 4 * i, gpout + 3 * i, gqout + 4 * i, gp + 3 * i, gq + 4 * i, gT12 + 12 * i);
}

/* rest-joint backward of the kinematic chain (learn_*_betas) */
void host_joint_chain_rest_joint_bwd(int J, const float* pose, const int* parents, const float* g_t, float* dJ) {
    float Rg[64 * 9], Gp[64 * 3];
    dwg_joint_chain_rest_joint_bwd(J, pose, parents, g_t, Rg, Gp, dJ);
}
