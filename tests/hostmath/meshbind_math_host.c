/* Host build of dreamwaltz-g_amd/csrc/meshbind_math.h for CPU-side derivative checks (test infrastructure only). */
#include <stdlib.h>
#include "../../dreamwaltz-g_amd/csrc/meshbind_math.h"

typedef const float (*cmat3)[3];
typedef float (*mat3)[3];

/* per point i: b[i,3], sc[i,3], P[i,3,3], Nv[i,3,3] (already gathered) */
void host_meshbind_forward(int n, float n_per_tri, const float* b, const float* sc, const float* P, const float* Nv, float* pos,
                           float* scl, float* quat) {
    for (int i = 0; i < n; i++)
        dwg_meshbind_point(b + 3 * i, sc + 3 * i, (cmat3)(P + 9 * i), (cmat3)(Nv + 9 * i), n_per_tri, pos + 3 * i, scl + 3 * i,
                           quat + 4 * i);
}
/* gP / gN ([n,3,3], may be NULL): gradients w.r.t. the gathered vertices / normals of each point */
void host_meshbind_backward(int n, float n_per_tri, const float* b, const float* sc, const float* P, const float* Nv,
                            const float* gpos, const float* gscl, const float* gquat, float* gb, float* gsc, float* gP, float* gN) {
    for (int i = 0; i < n; i++) {
        gb[3 * i] = gb[3 * i + 1] = gb[3 * i + 2] = 0.f;
        if (gP) for (int k = 0; k < 9; k++) { gP[9 * i + k] = 0.f; gN[9 * i + k] = 0.f; }
        dwg_meshbind_point_bwd(b + 3 * i, sc + 3 * i, (cmat3)(P + 9 * i), (cmat3)(Nv + 9 * i), n_per_tri, gpos + 3 * i, gscl + 3 * i,
                               gquat + 4 * i, gb + 3 * i, gsc + 3 * i, gP ? (mat3)(gP + 9 * i) : 0, gP ? (mat3)(gN + 9 * i) : 0);
    }
}
/* vertex normals backward on the host: the same two steps as k_vertex_normals_bwd_sum / k_face_normals_bwd */
void host_vertex_normals_backward(int Vp, int Fp, const float* verts, const int* tri, const float* g_vn, float* g_verts) {
    float* fn = (float*)malloc(sizeof(float) * 3 * (size_t)Fp);
    float* s = (float*)calloc(3 * (size_t)Vp, sizeof(float));
    float* gs = (float*)calloc(3 * (size_t)Vp, sizeof(float));
    for (int f = 0; f < Fp; f++) {
        const float* a = verts + 3 * tri[3 * f]; const float* b = verts + 3 * tri[3 * f + 1]; const float* c = verts + 3 * tri[3 * f + 2];
        float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, n[3];
        dwg_mb_cross(e1, e2, n);
        float inv = 1.f / sqrtf(fmaxf(dwg_mb_dot(n, n), 1e-20f));
        for (int k = 0; k < 3; k++) { fn[3 * f + k] = n[k] * inv; for (int v = 0; v < 3; v++) s[3 * tri[3 * f + v] + k] += n[k] * inv; }
    }
    for (int v = 0; v < Vp; v++) {
        if (dwg_mb_dot(s + 3 * v, s + 3 * v) > 1e-20f) dwg_mb_safe_normalize_bwd(s + 3 * v, g_vn + 3 * v, gs + 3 * v);
    }
    for (int f = 0; f < Fp; f++) {
        int ia = tri[3 * f], ib = tri[3 * f + 1], ic = tri[3 * f + 2];
        float gfn[3], ga[3], gb[3], gc[3];
        for (int k = 0; k < 3; k++) gfn[k] = gs[3 * ia + k] + gs[3 * ib + k] + gs[3 * ic + k];
        dwg_mb_face_normal_bwd(verts + 3 * ia, verts + 3 * ib, verts + 3 * ic, gfn, ga, gb, gc);
        for (int k = 0; k < 3; k++) { g_verts[3 * ia + k] += ga[k]; g_verts[3 * ib + k] += gb[k]; g_verts[3 * ic + k] += gc[k]; }
    }
    free(fn); free(s); free(gs);
}
