// meshbind_math.h -- per-point arithmetic of the mesh-bound Gaussians (hands / face), written once for meshbind.hip.
// (tests/ also compiles this header with gcc to check the hand-derived backward against autograd on CPU -- that host
//  build is test infrastructure, the product only runs it inside the gfx950 kernels.)
//
// Follows what the reference computes through PyTorch autograd:
//   MeshBindingGaussianModel.get_positions               /root/reference/core/system/avatar.py:1016-1025
//   MeshBindingGaussianModel.get_scales_and_quaternions  /root/reference/core/system/avatar.py:1027-1079
// including its quirks (SURVEY.md checklist): normals interpolated with the RAW barycentric coordinates while positions
// use the sum-normalised ones, tangent extents divided by n_points_per_triangle, s0 = 0 exactly (Q5); the tangent frame
// v1 = normalize(v0 x (1,0,0)) with eps 1e-9 added to the norm (Q6); rows 1,2 of the frame negated once (Q3).
#pragma once
#include "lbs_math.h"

#define DWG_MB_EPS 1e-9f

DWG_HD void dwg_mb_cross(const float a[3], const float b[3], float c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
DWG_HD float dwg_mb_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// u = c / (|c| + eps); returns |c|
DWG_HD float dwg_mb_unit(const float c[3], float u[3]) {
    float n = sqrtf(dwg_mb_dot(c, c));
    float inv = 1.f / (n + DWG_MB_EPS);
    u[0] = c[0] * inv; u[1] = c[1] * inv; u[2] = c[2] * inv;
    return n;
}
// gradient of dwg_mb_unit: gc = gu/(n+eps) - (c/n) (c . gu)/(n+eps)^2   (norm subgradient 0 at c = 0, like torch)
DWG_HD void dwg_mb_unit_bwd(const float c[3], float n, const float gu[3], float gc[3]) {
    float inv = 1.f / (n + DWG_MB_EPS);
    float k = n > 0.f ? dwg_mb_dot(c, gu) * inv * inv / n : 0.f;
    for (int i = 0; i < 3; i++) gc[i] = gu[i] * inv - c[i] * k;
}

typedef struct DwgMeshFrame {
    float v0[3], v1[3], v2[3];      // normal, tangent, bitangent
    float pn[3], c1[3], c2[3];      // pre-normalisation vectors
    float n0, n1, n2;               // their norms
} DwgMeshFrame;

DWG_HD void dwg_mb_frame(const float b[3], const float Nv[3][3], DwgMeshFrame* f) {
    const float ref[3] = {1.f, 0.f, 0.f};
    for (int i = 0; i < 3; i++) f->pn[i] = b[0] * Nv[0][i] + b[1] * Nv[1][i] + b[2] * Nv[2][i];
    f->n0 = dwg_mb_unit(f->pn, f->v0);
    dwg_mb_cross(f->v0, ref, f->c1);
    f->n1 = dwg_mb_unit(f->c1, f->v1);
    dwg_mb_cross(f->v0, f->v1, f->c2);
    f->n2 = dwg_mb_unit(f->c2, f->v2);
}

// One mesh-bound Gaussian: b = raw barycentric coordinates, sc = raw scale parameters (only [1],[2] are used),
// P = the triangle's posed vertices, Nv = their vertex normals.
DWG_HD void dwg_meshbind_point(const float b[3], const float sc[3], const float P[3][3], const float Nv[3][3], float n_per_tri,
                               float pos[3], float scl[3], float quat[4]) {
    const float S = b[0] + b[1] + b[2];
    for (int i = 0; i < 3; i++) pos[i] = (b[0] / S) * P[0][i] + (b[1] / S) * P[1][i] + (b[2] / S) * P[2][i];
    DwgMeshFrame f;
    dwg_mb_frame(b, Nv, &f);
    const float sg[3] = {1.f, -1.f, -1.f};
    float m[9];
    for (int i = 0; i < 3; i++) { m[3 * i] = sg[i] * f.v0[i]; m[3 * i + 1] = sg[i] * f.v1[i]; m[3 * i + 2] = sg[i] * f.v2[i]; }
    dwg_matrix_to_quat(m, quat);
    if (quat[0] < 0.f) for (int a = 0; a < 4; a++) quat[a] = -quat[a];          // standardize_quaternion
    float e1 = 0.f, e2 = 0.f;
    for (int k = 0; k < 3; k++) {
        float d[3] = {P[k][0] - pos[0], P[k][1] - pos[1], P[k][2] - pos[2]};
        e1 += fabsf(dwg_mb_dot(d, f.v1)); e2 += fabsf(dwg_mb_dot(d, f.v2));
    }
    scl[0] = 0.f;
    scl[1] = e1 / n_per_tri * fminf(fmaxf(sc[1], 0.5f), 2.f);
    scl[2] = e2 / n_per_tri * fminf(fmaxf(sc[2], 0.5f), 2.f);
}

// positions only (the canonical pass feeds the grid encoder: avatar.py:1328-1355)
DWG_HD void dwg_meshbind_position(const float b[3], const float P[3][3], float pos[3]) {
    const float S = b[0] + b[1] + b[2];
    for (int i = 0; i < 3; i++) pos[i] = (b[0] / S) * P[0][i] + (b[1] / S) * P[1][i] + (b[2] / S) * P[2][i];
}
// adds the gradient of dwg_meshbind_position w.r.t. b into gb and, when gP is given, w.r.t. the three vertices into gP
DWG_HD void dwg_meshbind_position_bwd(const float b[3], const float P[3][3], const float gpos[3], float gb[3], float (*gP)[3]) {
    const float S = b[0] + b[1] + b[2];
    float gbn[3], mix = 0.f;
    for (int v = 0; v < 3; v++) { gbn[v] = dwg_mb_dot(gpos, P[v]); mix += gbn[v] * (b[v] / S); }
    for (int v = 0; v < 3; v++) gb[v] += (gbn[v] - mix) / S;
    if (gP) for (int v = 0; v < 3; v++) for (int i = 0; i < 3; i++) gP[v][i] += (b[v] / S) * gpos[i];
}

DWG_HD float dwg_mb_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// Backward of dwg_meshbind_point w.r.t. b and sc and -- when gP / gN are given (learn_*_betas: the posed vertices then depend
// on a learnable parameter, avatar.py:1551-1577) -- w.r.t. the triangle's vertices and vertex normals.
// gb, gP, gN are ACCUMULATED into (the canonical-position gradient shares gb); gsc is written.
DWG_HD void dwg_meshbind_point_bwd(const float b[3], const float sc[3], const float P[3][3], const float Nv[3][3], float n_per_tri,
                                   const float gpos_in[3], const float gscl[3], const float gquat[4], float gb[3], float gsc[3],
                                   float (*gP)[3], float (*gN)[3]) {
    const float ref[3] = {1.f, 0.f, 0.f};
    const float sg[3] = {1.f, -1.f, -1.f};
    const float S = b[0] + b[1] + b[2];
    float pos[3];
    for (int i = 0; i < 3; i++) pos[i] = (b[0] / S) * P[0][i] + (b[1] / S) * P[1][i] + (b[2] / S) * P[2][i];
    DwgMeshFrame f;
    dwg_mb_frame(b, Nv, &f);
    float m[9], q[4], gq[4], gm[9];
    for (int i = 0; i < 3; i++) { m[3 * i] = sg[i] * f.v0[i]; m[3 * i + 1] = sg[i] * f.v1[i]; m[3 * i + 2] = sg[i] * f.v2[i]; }
    dwg_matrix_to_quat(m, q);
    const float qs = q[0] < 0.f ? -1.f : 1.f;
    for (int a = 0; a < 4; a++) gq[a] = qs * gquat[a];
    dwg_matrix_to_quat_bwd(m, gq, gm);
    float gv0[3], gv1[3], gv2[3], gpos[3] = {gpos_in[0], gpos_in[1], gpos_in[2]};
    for (int i = 0; i < 3; i++) { gv0[i] = sg[i] * gm[3 * i]; gv1[i] = sg[i] * gm[3 * i + 1]; gv2[i] = sg[i] * gm[3 * i + 2]; }
    // tangent extents
    float e1 = 0.f, e2 = 0.f;
    const float c1c = fminf(fmaxf(sc[1], 0.5f), 2.f), c2c = fminf(fmaxf(sc[2], 0.5f), 2.f);
    const float w1 = gscl[1] * c1c / n_per_tri, w2 = gscl[2] * c2c / n_per_tri;
    for (int k = 0; k < 3; k++) {
        float d[3] = {P[k][0] - pos[0], P[k][1] - pos[1], P[k][2] - pos[2]};
        float d1 = dwg_mb_dot(d, f.v1), d2 = dwg_mb_dot(d, f.v2);
        e1 += fabsf(d1); e2 += fabsf(d2);
        float s1 = w1 * dwg_mb_sign(d1), s2 = w2 * dwg_mb_sign(d2);
        for (int i = 0; i < 3; i++) {
            gv1[i] += s1 * d[i]; gv2[i] += s2 * d[i];
            gpos[i] -= s1 * f.v1[i] + s2 * f.v2[i];
            if (gP) gP[k][i] += s1 * f.v1[i] + s2 * f.v2[i];
        }
    }
    gsc[0] = 0.f;
    gsc[1] = (sc[1] >= 0.5f && sc[1] <= 2.f) ? gscl[1] * e1 / n_per_tri : 0.f;
    gsc[2] = (sc[2] >= 0.5f && sc[2] <= 2.f) ? gscl[2] * e2 / n_per_tri : 0.f;
    // frame chain: v2 <- c2 = v0 x v1 ; v1 <- c1 = v0 x ref ; v0 <- pn
    float gc2[3], gc1[3], gpn[3], t[3];
    dwg_mb_unit_bwd(f.c2, f.n2, gv2, gc2);
    dwg_mb_cross(f.v1, gc2, t); for (int i = 0; i < 3; i++) gv0[i] += t[i];      // d/da (a x b) = b x g
    dwg_mb_cross(gc2, f.v0, t); for (int i = 0; i < 3; i++) gv1[i] += t[i];      // d/db (a x b) = g x a
    dwg_mb_unit_bwd(f.c1, f.n1, gv1, gc1);
    dwg_mb_cross(ref, gc1, t); for (int i = 0; i < 3; i++) gv0[i] += t[i];
    dwg_mb_unit_bwd(f.pn, f.n0, gv0, gpn);
    for (int v = 0; v < 3; v++) gb[v] += dwg_mb_dot(gpn, Nv[v]);
    if (gN) for (int v = 0; v < 3; v++) for (int i = 0; i < 3; i++) gN[v][i] += b[v] * gpn[i];
    dwg_meshbind_position_bwd(b, P, gpos, gb, gP);
}

// ---- vertex normals (utils/mesh.py:34-94) backward pieces ----
// y = x / sqrt(max(x.x, 1e-20)) (safe_normalize): gx from gy
DWG_HD void dwg_mb_safe_normalize_bwd(const float x[3], const float gy[3], float gx[3]) {
    const float n2 = dwg_mb_dot(x, x);
    if (n2 > 1e-20f) {
        const float inv = 1.f / sqrtf(n2);
        const float y[3] = {x[0] * inv, x[1] * inv, x[2] * inv};
        const float d = dwg_mb_dot(y, gy);
        for (int i = 0; i < 3; i++) gx[i] = (gy[i] - y[i] * d) * inv;
    } else {
        for (int i = 0; i < 3; i++) gx[i] = gy[i] * 1e10f;          // clamp active: d/dx (x / 1e-10)
    }
}
// face normal fn = safe_normalize((b - a) x (c - a)): gradient w.r.t. the three corners from g_fn
DWG_HD void dwg_mb_face_normal_bwd(const float a[3], const float b[3], const float c[3], const float gfn[3], float ga[3], float gb_[3],
                                   float gc[3]) {
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    float n[3], gn[3], ge1[3], ge2[3];
    dwg_mb_cross(e1, e2, n);
    dwg_mb_safe_normalize_bwd(n, gfn, gn);
    dwg_mb_cross(e2, gn, ge1);        // d/da (a x b) . g = b x g
    dwg_mb_cross(gn, e1, ge2);        // d/db (a x b) . g = g x a
    for (int i = 0; i < 3; i++) { ga[i] = -(ge1[i] + ge2[i]); gb_[i] = ge1[i]; gc[i] = ge2[i]; }
}
