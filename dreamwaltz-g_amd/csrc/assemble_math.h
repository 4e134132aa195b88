// assemble_math.h -- per-Gaussian activations / non-rigid composition of DreamWaltzG.animate, written once for assemble.hip.
// (tests/ also compiles this header with gcc to check the hand-derived backward against autograd on CPU.)
//
// Follows what the reference computes through PyTorch:
//   DreamWaltzG.non_rigid_transform, default flags        /root/reference/core/system/avatar.py:1464-1498
//       positions += offsets * init_offset ; scales = exp(_scales) + mlp_scales * init_scale  (additive branch, checklist Q4)
//       quaternions = F.normalize(_quaternions)            (use_non_rigid_rotations = False)
//   DreamWaltzG.static_mlp_forward                         /root/reference/core/system/avatar.py:1283-1290
//       colors = sigmoid(h[:, 1:4]) ; opacities = sigmoid(h[:, 0:1]) or ones when fix_opacities
//   GaussianModel activations (exp / sigmoid / normalize)  /root/reference/core/gaussian/gaussian_model.py:25-56
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DWG_AHD __host__ __device__ __forceinline__
#else
#define DWG_AHD static inline
#endif

DWG_AHD float dwg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// geometry of one free Gaussian
DWG_AHD void dwg_assemble_geom(const float p[3], const float off[3], float init_offset, const float ls[3], const float ms[3],
                               float init_scale, const float q[4], float pos[3], float scl[3], float qn[4]) {
    for (int c = 0; c < 3; c++) { pos[c] = p[c] + off[c] * init_offset; scl[c] = expf(ls[c]) + ms[c] * init_scale; }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float d = fmaxf(n, 1e-12f);                       // F.normalize: x / max(|x|, eps)
    for (int c = 0; c < 4; c++) qn[c] = q[c] / d;
}

DWG_AHD void dwg_assemble_geom_bwd(const float ls[3], const float q[4], float init_offset, float init_scale, const float gpos[3],
                                   const float gscl[3], const float gqn[4], float dp[3], float doff[3], float dls[3], float dms[3],
                                   float dq[4]) {
    for (int c = 0; c < 3; c++) {
        dp[c] = gpos[c]; doff[c] = gpos[c] * init_offset;
        dls[c] = gscl[c] * expf(ls[c]); dms[c] = gscl[c] * init_scale;
    }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 1e-12f) {                                       // d (q/n) = (g - qn (qn . g)) / n
        float qn[4], dot = 0.f;
        for (int c = 0; c < 4; c++) { qn[c] = q[c] / n; dot += qn[c] * gqn[c]; }
        for (int c = 0; c < 4; c++) dq[c] = (gqn[c] - qn[c] * dot) / n;
    } else {                                                // clamped branch: q / eps
        for (int c = 0; c < 4; c++) dq[c] = gqn[c] / 1e-12f;
    }
}

// appearance of one Gaussian from the static MLP's raw output h = (opacity logit, r, g, b logits)
DWG_AHD void dwg_assemble_color(const float h[4], int fix_opacity, float col[3], float* opac) {
    col[0] = dwg_sigmoid(h[1]); col[1] = dwg_sigmoid(h[2]); col[2] = dwg_sigmoid(h[3]);
    *opac = fix_opacity ? 1.f : dwg_sigmoid(h[0]);
}

DWG_AHD void dwg_assemble_color_bwd(const float h[4], int fix_opacity, const float gcol[3], float gopac, float dh[4]) {
    for (int c = 0; c < 3; c++) { const float s = dwg_sigmoid(h[c + 1]); dh[c + 1] = gcol[c] * s * (1.f - s); }
    if (fix_opacity) dh[0] = 0.f;
    else { const float s = dwg_sigmoid(h[0]); dh[0] = gopac * s * (1.f - s); }
}
