// lbs_math.h -- per-Gaussian arithmetic of the LBS stage, written once for the HIP kernels in lbs.hip.
// (tests/ also compiles this header with g++ to check the hand-derived backward against autograd on CPU --
//  that host build is test infrastructure, the product only runs it inside the gfx950 kernels.)
//
// Follows, line by line, what the reference computes through PyTorch:
//   RigidTransform.transform_points      /root/reference/core/human/inverse_lbs.py:190-210
//   RigidTransform.transform_quaternions /root/reference/core/human/inverse_lbs.py:212-242 (flip_rotation_axis=True)
//   pytorch3d quaternion_to_matrix / matrix_to_quaternion (0.7.5)  [restated; see oracle/animate.py]
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DWG_HD __host__ __device__ __forceinline__
#else
#define DWG_HD static inline
#endif

// q = (r,i,j,k) -> R row-major, scaled by 2/|q|^2 so that non-unit q still yields a rotation
DWG_HD void dwg_quat_to_matrix(const float q[4], float R[9]) {
    float r = q[0], i = q[1], j = q[2], k = q[3];
    float ts = 2.f / (r * r + i * i + j * j + k * k);
    R[0] = 1.f - ts * (j * j + k * k); R[1] = ts * (i * j - k * r);       R[2] = ts * (i * k + j * r);
    R[3] = ts * (i * j + k * r);       R[4] = 1.f - ts * (i * i + k * k); R[5] = ts * (j * k - i * r);
    R[6] = ts * (i * k - j * r);       R[7] = ts * (j * k + i * r);       R[8] = 1.f - ts * (i * i + j * j);
}

DWG_HD void dwg_quat_to_matrix_bwd(const float q[4], const float gR[9], float gq[4]) {
    float r = q[0], i = q[1], j = q[2], k = q[3];
    float s = r * r + i * i + j * j + k * k;
    float ts = 2.f / s;
    float N[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                  i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float gts = 0.f;
    for (int a = 0; a < 9; a++) gts += gR[a] * N[a];
    float c = -ts * ts * gts;  // d ts / d q_c = -ts^2 * q_c
    gq[0] = ts * (-k * gR[1] + j * gR[2] + k * gR[3] - i * gR[5] - j * gR[6] + i * gR[7]) + c * r;
    gq[1] = ts * (j * gR[1] + k * gR[2] + j * gR[3] - 2.f * i * gR[4] - r * gR[5] + k * gR[6] + r * gR[7] - 2.f * i * gR[8]) + c * i;
    gq[2] = ts * (-2.f * j * gR[0] + i * gR[1] + r * gR[2] + i * gR[3] + k * gR[5] - r * gR[6] + k * gR[7] - 2.f * j * gR[8]) + c * j;
    gq[3] = ts * (-2.f * k * gR[0] - r * gR[1] + i * gR[2] + r * gR[3] - 2.f * k * gR[4] + j * gR[5] + i * gR[6] + j * gR[7]) + c * k;
}

// matrix_to_quaternion (best-conditioned of 4 candidates, floor 0.1, not standardised). Returns the branch.
DWG_HD int dwg_matrix_to_quat(const float m[9], float out[4]) {
    float x[4] = {1.f + m[0] + m[4] + m[8], 1.f + m[0] - m[4] - m[8], 1.f - m[0] + m[4] - m[8], 1.f - m[0] - m[4] + m[8]};
    float qa[4];
    int b = 0;
    for (int a = 0; a < 4; a++) qa[a] = x[a] > 0.f ? sqrtf(x[a]) : 0.f;
    for (int a = 1; a < 4; a++) if (qa[a] > qa[b]) b = a;  // first maximum, like torch.argmax
    float num[4];
    float sq = qa[b] * qa[b];
    if (b == 0) { num[0] = sq; num[1] = m[7] - m[5]; num[2] = m[2] - m[6]; num[3] = m[3] - m[1]; }
    else if (b == 1) { num[0] = m[7] - m[5]; num[1] = sq; num[2] = m[3] + m[1]; num[3] = m[2] + m[6]; }
    else if (b == 2) { num[0] = m[2] - m[6]; num[1] = m[3] + m[1]; num[2] = sq; num[3] = m[5] + m[7]; }
    else { num[0] = m[3] - m[1]; num[1] = m[6] + m[2]; num[2] = m[7] + m[5]; num[3] = sq; }
    float den = 2.f * fmaxf(qa[b], 0.1f);
    for (int a = 0; a < 4; a++) out[a] = num[a] / den;
    return b;
}

DWG_HD void dwg_matrix_to_quat_bwd(const float m[9], const float gout[4], float gm[9]) {
    float x[4] = {1.f + m[0] + m[4] + m[8], 1.f + m[0] - m[4] - m[8], 1.f - m[0] + m[4] - m[8], 1.f - m[0] - m[4] + m[8]};
    float qa[4];
    int b = 0;
    for (int a = 0; a < 4; a++) qa[a] = x[a] > 0.f ? sqrtf(x[a]) : 0.f;
    for (int a = 1; a < 4; a++) if (qa[a] > qa[b]) b = a;
    float num[4];
    float sq = qa[b] * qa[b];
    if (b == 0) { num[0] = sq; num[1] = m[7] - m[5]; num[2] = m[2] - m[6]; num[3] = m[3] - m[1]; }
    else if (b == 1) { num[0] = m[7] - m[5]; num[1] = sq; num[2] = m[3] + m[1]; num[3] = m[2] + m[6]; }
    else if (b == 2) { num[0] = m[2] - m[6]; num[1] = m[3] + m[1]; num[2] = sq; num[3] = m[5] + m[7]; }
    else { num[0] = m[3] - m[1]; num[1] = m[6] + m[2]; num[2] = m[7] + m[5]; num[3] = sq; }
    float den = 2.f * fmaxf(qa[b], 0.1f);
    float gn[4], gden = 0.f;
    for (int a = 0; a < 4; a++) { gn[a] = gout[a] / den; gden -= gout[a] * num[a] / (den * den); }
    float gqa = (qa[b] > 0.1f ? 2.f * gden : 0.f) + 2.f * qa[b] * gn[b];
    float gx = x[b] > 0.f ? gqa / (2.f * qa[b]) : 0.f;
    for (int a = 0; a < 9; a++) gm[a] = 0.f;
    const float s0[4] = {1.f, 1.f, -1.f, -1.f}, s1[4] = {1.f, -1.f, 1.f, -1.f}, s2[4] = {1.f, -1.f, -1.f, 1.f};
    gm[0] += s0[b] * gx; gm[4] += s1[b] * gx; gm[8] += s2[b] * gx;
    if (b == 0) { gm[7] += gn[1]; gm[5] -= gn[1]; gm[2] += gn[2]; gm[6] -= gn[2]; gm[3] += gn[3]; gm[1] -= gn[3]; }
    else if (b == 1) { gm[7] += gn[0]; gm[5] -= gn[0]; gm[3] += gn[2]; gm[1] += gn[2]; gm[2] += gn[3]; gm[6] += gn[3]; }
    else if (b == 2) { gm[2] += gn[0]; gm[6] -= gn[0]; gm[3] += gn[1]; gm[1] += gn[1]; gm[5] += gn[3]; gm[7] += gn[3]; }
    else { gm[3] += gn[0]; gm[1] -= gn[0]; gm[6] += gn[1]; gm[2] += gn[1]; gm[7] += gn[2]; gm[5] += gn[2]; }
}

// T12 = blended [R|T] rows (3x4 row-major). Point + quaternion transform with the row-1,2 flips (checklist Q3).
DWG_HD void dwg_lbs_apply(const float T12[12], const float p[3], const float* q /*4 or null*/, float pout[3], float* qout) {
    for (int r = 0; r < 3; r++) pout[r] = T12[4 * r] * p[0] + T12[4 * r + 1] * p[1] + T12[4 * r + 2] * p[2] + T12[4 * r + 3];
    if (!q) return;
    const float sg[3] = {1.f, -1.f, -1.f};
    float Rq[9], M[9];
    dwg_quat_to_matrix(q, Rq);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            float a = 0.f;
            for (int k = 0; k < 3; k++) a += (sg[r] * sg[k] * T12[4 * r + k]) * Rq[3 * k + c];
            M[3 * r + c] = a;
        }
    dwg_matrix_to_quat(M, qout);
}

// Backward of dwg_lbs_apply w.r.t. p and q (and, if gT12 != null, w.r.t. the blended transform).
DWG_HD void dwg_lbs_apply_bwd(const float T12[12], const float p[3], const float* q, const float gpout[3], const float* gqout,
                              float gp[3], float* gq, float* gT12) {
    for (int c = 0; c < 3; c++) gp[c] = T12[c] * gpout[0] + T12[4 + c] * gpout[1] + T12[8 + c] * gpout[2];
    if (gT12) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) gT12[4 * r + c] = gpout[r] * p[c];
            gT12[4 * r + 3] = gpout[r];
        }
    }
    if (!q) return;
    const float sg[3] = {1.f, -1.f, -1.f};
    float Rq[9], M[9], gM[9], gRq[9];
    dwg_quat_to_matrix(q, Rq);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            float a = 0.f;
            for (int k = 0; k < 3; k++) a += (sg[r] * sg[k] * T12[4 * r + k]) * Rq[3 * k + c];
            M[3 * r + c] = a;
        }
    dwg_matrix_to_quat_bwd(M, gqout, gM);
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) {
            float a = 0.f;
            for (int r = 0; r < 3; r++) a += (sg[r] * sg[k] * T12[4 * r + k]) * gM[3 * r + c];
            gRq[3 * k + c] = a;
        }
    dwg_quat_to_matrix_bwd(q, gRq, gq);
    if (gT12) {
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                float a = 0.f;
                for (int c = 0; c < 3; c++) a += gM[3 * r + c] * Rq[3 * k + c];
                gT12[4 * r + k] += sg[r] * sg[k] * a;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Skeleton backward w.r.t. the REST joints (needed for `learn_hand_betas` / `learn_face_betas`: avatar.py:1551-1553; the joint
// rotations do not depend on the shape coefficients, the rest joints do: J = J_template + (J_regressor . shapedirs) . betas).
//   smplx.lbs.batch_rigid_transform:  rel_k = J_k - J_parent(k) (root: J_0);  Rg_k = Rg_parent R_k;  p_k = p_parent + Rg_parent rel_k;
//                                     A_k = [Rg_k | p_k - Rg_k J_k]
// Given g_t[k] = d loss / d A_k[:3, 3] this writes dJ[k] = d loss / d J_k (everything linear in J for fixed rotations).
// ---------------------------------------------------------------------------------------------------------------------
// Rodrigues with angle = |r + 1e-8| (smplx.lbs.batch_rodrigues)
DWG_HD void dwg_rodrigues(const float r[3], float R[9]) {
    float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
    float angle = sqrtf(ax * ax + ay * ay + az * az);
    float dx = r[0] / angle, dy = r[1] / angle, dz = r[2] / angle;
    float s = sinf(angle), c = cosf(angle);
    float K[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            float kk = K[3 * a] * K[b] + K[3 * a + 1] * K[3 + b] + K[3 * a + 2] * K[6 + b];
            R[3 * a + b] = (a == b ? 1.f : 0.f) + s * K[3 * a + b] + (1.f - c) * kk;
        }
}

// Rg [J][9] and Gp [J][3] are scratch.  parents[k] < k for k > 0.
DWG_HD void dwg_joint_chain_rest_joint_bwd(int J, const float* pose /*[J,3]*/, const int* parents, const float* g_t /*[J,3]*/,
                                           float* Rg, float* Gp, float* dJ /*[J,3]*/) {
    for (int k = 0; k < J; k++) {
        float R[9];
        dwg_rodrigues(pose + 3 * k, R);
        if (k == 0) { for (int e = 0; e < 9; e++) Rg[e] = R[e]; }
        else {
            const float* P = Rg + 9 * parents[k];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) Rg[9 * k + 3 * a + b] = P[3 * a] * R[b] + P[3 * a + 1] * R[3 + b] + P[3 * a + 2] * R[6 + b];
        }
        for (int c = 0; c < 3; c++) { Gp[3 * k + c] = g_t[3 * k + c]; dJ[3 * k + c] = 0.f; }
    }
    for (int k = J - 1; k > 0; k--)                      // d loss / d p_k: own term + every descendant's
        for (int c = 0; c < 3; c++) Gp[3 * parents[k] + c] += Gp[3 * k + c];
    for (int k = 0; k < J; k++) {
        const float* Rk = Rg + 9 * k;
        for (int c = 0; c < 3; c++)                      // t_k = p_k - Rg_k J_k
            dJ[3 * k + c] -= Rk[c] * g_t[3 * k] + Rk[3 + c] * g_t[3 * k + 1] + Rk[6 + c] * g_t[3 * k + 2];
        if (k == 0) {
            for (int c = 0; c < 3; c++) dJ[c] += Gp[c];                      // p_0 = rel_0 = J_0
        } else {
            const float* P = Rg + 9 * parents[k];
            for (int c = 0; c < 3; c++) {                // p_k = p_parent + Rg_parent (J_k - J_parent)
                float g = P[c] * Gp[3 * k] + P[3 + c] * Gp[3 * k + 1] + P[6 + c] * Gp[3 * k + 2];
                dJ[3 * k + c] += g; dJ[3 * parents[k] + c] -= g;
            }
        }
    }
}
