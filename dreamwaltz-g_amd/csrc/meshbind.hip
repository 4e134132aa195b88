// meshbind.hip -- mesh-bound Gaussians (hands / face): vertex normals + per-point position / tangent frame / extents,
// forward and backward (include/dwg_meshbind.h; reference avatar.py:1016-1079, utils/mesh.py:34-94).
// ~10^4 points: the work is tiny and latency-bound, the win is ONE launch instead of the reference's ~150 op kernels
// per direction.  One lane per point / vertex / face; all per-point arithmetic lives in meshbind_math.h.
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "meshbind_math.h"
#include "../../include/dwg_meshbind.h"

namespace {

__global__ __launch_bounds__(256) void k_face_normals(int Fp, const float* __restrict__ verts, const int* __restrict__ tri,
                                                      float* __restrict__ fn) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= Fp) return;
    const float* a = verts + 3 * (size_t)tri[3 * f];
    const float* b = verts + 3 * (size_t)tri[3 * f + 1];
    const float* c = verts + 3 * (size_t)tri[3 * f + 2];
    float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, n[3];
    dwg_mb_cross(e1, e2, n);
    const float inv = 1.f / sqrtf(fmaxf(dwg_mb_dot(n, n), 1e-20f));      // safe_normalize
    fn[3 * f] = n[0] * inv; fn[3 * f + 1] = n[1] * inv; fn[3 * f + 2] = n[2] * inv;
}

__global__ __launch_bounds__(256) void k_vertex_normals(int Vp, const float* __restrict__ fn, const int* __restrict__ vf_off,
                                                        const int* __restrict__ vf_faces, float* __restrict__ vn) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= Vp) return;
    float n[3] = {0.f, 0.f, 0.f};
    for (int e = vf_off[v]; e < vf_off[v + 1]; e++) {
        const float* f = fn + 3 * (size_t)vf_faces[e];
        n[0] += f[0]; n[1] += f[1]; n[2] += f[2];
    }
    if (!(dwg_mb_dot(n, n) > 1e-20f)) { n[0] = 0.f; n[1] = 0.f; n[2] = 1.f; }
    const float inv = 1.f / sqrtf(fmaxf(dwg_mb_dot(n, n), 1e-20f));
    vn[3 * v] = n[0] * inv; vn[3 * v + 1] = n[1] * inv; vn[3 * v + 2] = n[2] * inv;
}

__device__ __forceinline__ void gather3(const float* __restrict__ src, const int* __restrict__ t, float out[3][3]) {
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const float* p = src + 3 * (size_t)t[v];
        out[v][0] = p[0]; out[v][1] = p[1]; out[v][2] = p[2];
    }
}

__global__ __launch_bounds__(256) void k_meshbind_fwd(int M, int n_per, const float* __restrict__ bary, const float* __restrict__ sc,
                                                      const float* __restrict__ vc, const float* __restrict__ vo,
                                                      const float* __restrict__ vn, const int* __restrict__ tri,
                                                      float* __restrict__ pos_c, float* __restrict__ pos, float* __restrict__ scl,
                                                      float* __restrict__ quat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int* t = tri + 3 * (size_t)(i / n_per);
    const float b[3] = {bary[3 * (size_t)i], bary[3 * (size_t)i + 1], bary[3 * (size_t)i + 2]};
    const float s[3] = {sc[3 * (size_t)i], sc[3 * (size_t)i + 1], sc[3 * (size_t)i + 2]};
    float P[3][3], N[3][3], po[3], so[3], qo[4];
    gather3(vo, t, P); gather3(vn, t, N);
    dwg_meshbind_point(b, s, P, N, (float)n_per, po, so, qo);
#pragma unroll
    for (int c = 0; c < 3; c++) { pos[3 * (size_t)i + c] = po[c]; scl[3 * (size_t)i + c] = so[c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) quat[4 * (size_t)i + c] = qo[c];
    if (vc && pos_c) {
        gather3(vc, t, P);
        dwg_meshbind_position(b, P, po);
#pragma unroll
        for (int c = 0; c < 3; c++) pos_c[3 * (size_t)i + c] = po[c];
    }
}

__global__ __launch_bounds__(256) void k_meshbind_bwd(int M, int n_per, const float* __restrict__ bary, const float* __restrict__ sc,
                                                      const float* __restrict__ vc, const float* __restrict__ vo,
                                                      const float* __restrict__ vn, const int* __restrict__ tri,
                                                      const float* __restrict__ g_pos_c, const float* __restrict__ g_pos,
                                                      const float* __restrict__ g_scl, const float* __restrict__ g_quat,
                                                      float* __restrict__ g_bary, float* __restrict__ g_sc,
                                                      float* __restrict__ g_vc /*[Vp,3] accumulate | null*/,
                                                      float* __restrict__ g_vo, float* __restrict__ g_vn,
                                                      float* __restrict__ corner /*[M][3][9] | null: per-(Gaussian, corner) rows {gP, gN, gPc} instead of the atomics*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int* t = tri + 3 * (size_t)(i / n_per);
    const float b[3] = {bary[3 * (size_t)i], bary[3 * (size_t)i + 1], bary[3 * (size_t)i + 2]};
    const float s[3] = {sc[3 * (size_t)i], sc[3 * (size_t)i + 1], sc[3 * (size_t)i + 2]};
    float P[3][3], N[3][3];
    gather3(vo, t, P); gather3(vn, t, N);
    float gp[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    if (g_pos) { gp[0] = g_pos[3 * (size_t)i]; gp[1] = g_pos[3 * (size_t)i + 1]; gp[2] = g_pos[3 * (size_t)i + 2]; }
    if (g_scl) { gs[0] = g_scl[3 * (size_t)i]; gs[1] = g_scl[3 * (size_t)i + 1]; gs[2] = g_scl[3 * (size_t)i + 2]; }
    if (g_quat) { gq[0] = g_quat[4 * (size_t)i]; gq[1] = g_quat[4 * (size_t)i + 1]; gq[2] = g_quat[4 * (size_t)i + 2]; gq[3] = g_quat[4 * (size_t)i + 3]; }
    float gb[3] = {0.f, 0.f, 0.f}, gsc[3];
    float gP[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, gN[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const bool want_v = g_vo != nullptr || corner != nullptr;
    dwg_meshbind_point_bwd(b, s, P, N, (float)n_per, gp, gs, gq, gb, gsc, want_v ? gP : nullptr, want_v ? gN : nullptr);
    if (corner) {
#pragma unroll
        for (int v = 0; v < 3; v++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                corner[((size_t)i * 3 + v) * 9 + c] = gP[v][c];
                corner[((size_t)i * 3 + v) * 9 + 3 + c] = gN[v][c];
                corner[((size_t)i * 3 + v) * 9 + 6 + c] = 0.f;
            }
    } else if (want_v) {
#pragma unroll
        for (int v = 0; v < 3; v++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                atomicAdd(g_vo + 3 * (size_t)t[v] + c, gP[v][c]);
                atomicAdd(g_vn + 3 * (size_t)t[v] + c, gN[v][c]);
            }
    }
    if (vc && g_pos_c) {
        gather3(vc, t, P);
        const float gc[3] = {g_pos_c[3 * (size_t)i], g_pos_c[3 * (size_t)i + 1], g_pos_c[3 * (size_t)i + 2]};
        float gPc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        dwg_meshbind_position_bwd(b, P, gc, gb, (g_vc || corner) ? gPc : nullptr);
        if (corner) {
#pragma unroll
            for (int v = 0; v < 3; v++)
#pragma unroll
                for (int c = 0; c < 3; c++) corner[((size_t)i * 3 + v) * 9 + 6 + c] = gPc[v][c];
        } else if (g_vc) {
#pragma unroll
            for (int v = 0; v < 3; v++)
#pragma unroll
                for (int c = 0; c < 3; c++) atomicAdd(g_vc + 3 * (size_t)t[v] + c, gPc[v][c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { g_bary[3 * (size_t)i + c] = gb[c]; g_sc[3 * (size_t)i + c] = gsc[c]; }
}

// Backward of k_vertex_normals / k_face_normals: g_vn [Vp,3] -> gradient of the summed face normals per vertex (g_s), then per
// face the sum over its corners and the face-normal chain, gathered per vertex over the incident-face table (k_face_normals_bwd_gather below).
__global__ __launch_bounds__(256) void k_vertex_normals_bwd_sum(int Vp, const float* __restrict__ fn, const int* __restrict__ vf_off,
                                                                const int* __restrict__ vf_faces, const float* __restrict__ g_vn,
                                                                float* __restrict__ g_s) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= Vp) return;
    float n[3] = {0.f, 0.f, 0.f};
    for (int e = vf_off[v]; e < vf_off[v + 1]; e++) {
        const float* f = fn + 3 * (size_t)vf_faces[e];
        n[0] += f[0]; n[1] += f[1]; n[2] += f[2];
    }
    float g[3] = {0.f, 0.f, 0.f};
    if (dwg_mb_dot(n, n) > 1e-20f) {          // otherwise the normal is the constant (0,0,1): no gradient
        const float gy[3] = {g_vn[3 * v], g_vn[3 * v + 1], g_vn[3 * v + 2]};
        dwg_mb_safe_normalize_bwd(n, gy, g);
    }
    g_s[3 * v] = g[0]; g_s[3 * v + 1] = g[1]; g_s[3 * v + 2] = g[2];
}

// True iff entry e of vertex v's incident-face list names a face that an EARLIER entry of the list already named (a face listing a vertex
// twice): its corners were added then.
__device__ __forceinline__ bool mb_seen_before(const int* __restrict__ vf_faces, int e0, int e) {
    for (int q = e0; q < e; q++) if (vf_faces[q] == vf_faces[e]) return true;
    return false;
}

// g_vo / g_vn / g_vc of vertex v = the rows of the Gaussians on its incident faces at the corner that IS v, added in list order (faces as the
// incident-face table lists them, Gaussians of a face in index order): one thread per vertex, no atomics, the same bits on every run.
__global__ __launch_bounds__(256) void k_meshbind_gather_verts(int Vp, int n_per, const int* __restrict__ tri, const int* __restrict__ vf_off,
                                                               const int* __restrict__ vf_faces, const float* __restrict__ corner,
                                                               float* __restrict__ g_vc, float* __restrict__ g_vo, float* __restrict__ g_vn) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= Vp) return;
    float a[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int e0 = vf_off[v], e1 = vf_off[v + 1];
    for (int e = e0; e < e1; e++) {
        if (mb_seen_before(vf_faces, e0, e)) continue;
        const int f = vf_faces[e];
        for (int c = 0; c < 3; c++) {
            if (tri[3 * (size_t)f + c] != v) continue;
            for (int k = 0; k < n_per; k++) {
                const float* r = corner + (((size_t)f * n_per + k) * 3 + c) * 9;
#pragma unroll
                for (int q = 0; q < 9; q++) a[q] += r[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
        g_vo[3 * (size_t)v + q] = a[q]; g_vn[3 * (size_t)v + q] = a[3 + q];
        if (g_vc) g_vc[3 * (size_t)v + q] = a[6 + q];
    }
}

// k_face_normals_bwd as a gather: vertex v adds, in list order, the corner gradients of its incident faces (a face's three corner gradients
// are recomputed by each of its corners' threads: a few thousand faces).  g_verts is ACCUMULATED into (one thread per vertex: no race).
__global__ __launch_bounds__(256) void k_face_normals_bwd_gather(int Vp, const float* __restrict__ verts, const int* __restrict__ tri,
                                                                 const int* __restrict__ vf_off, const int* __restrict__ vf_faces,
                                                                 const float* __restrict__ g_s, float* __restrict__ g_verts) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= Vp) return;
    float acc[3] = {0.f, 0.f, 0.f};
    const int e0 = vf_off[v], e1 = vf_off[v + 1];
    for (int e = e0; e < e1; e++) {
        if (mb_seen_before(vf_faces, e0, e)) continue;
        const int f = vf_faces[e];
        const int idx[3] = {tri[3 * (size_t)f], tri[3 * (size_t)f + 1], tri[3 * (size_t)f + 2]};
        float gfn[3], p3[3][3], gcorner[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) gfn[k] = g_s[3 * (size_t)idx[0] + k] + g_s[3 * (size_t)idx[1] + k] + g_s[3 * (size_t)idx[2] + k];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int k = 0; k < 3; k++) p3[c][k] = verts[3 * (size_t)idx[c] + k];
        dwg_mb_face_normal_bwd(p3[0], p3[1], p3[2], gfn, gcorner[0], gcorner[1], gcorner[2]);
#pragma unroll
        for (int c = 0; c < 3; c++)
            if (idx[c] == v) { acc[0] += gcorner[c][0]; acc[1] += gcorner[c][1]; acc[2] += gcorner[c][2]; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) g_verts[3 * (size_t)v + k] += acc[k];
}

}  // namespace

extern "C" {

int dwg_mesh_vertex_normals(int32_t Vp, int32_t Fp, const float* verts, const int32_t* triangles, const int32_t* vf_offsets,
                            const int32_t* vf_faces, float* face_normals, float* vertex_normals, dwg_stream_t stream_) {
    if (Vp < 0 || Fp < 0) return DWG_E_ARG;
    if (Vp == 0) return DWG_OK;
    if (!verts || !vf_offsets || !vertex_normals || (Fp > 0 && (!triangles || !vf_faces || !face_normals))) return DWG_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (Fp > 0)
        DWG_LAUNCH("mesh_face_normals", k_face_normals, dim3(dwg_cdiv(Fp, 256)), dim3(256), 0, stream, Fp, verts, triangles, face_normals);
    DWG_LAUNCH("mesh_vertex_normals", k_vertex_normals, dim3(dwg_cdiv(Vp, 256)), dim3(256), 0, stream, Vp, (const float*)face_normals,
               vf_offsets, vf_faces, vertex_normals);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_meshbind_forward(int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params, const float* verts_cnl,
                         const float* verts_obs, const float* vnormals_obs, const int32_t* triangles, float* pos_cnl_out,
                         float* pos_out, float* scales_out, float* quats_out, dwg_stream_t stream_) {
    if (Fp < 0 || n_per_tri <= 0) return DWG_E_ARG;
    if (Fp == 0) return DWG_OK;
    if (!bary || !scale_params || !verts_obs || !vnormals_obs || !triangles || !pos_out || !scales_out || !quats_out) return DWG_E_ARG;
    if ((verts_cnl == nullptr) != (pos_cnl_out == nullptr)) return DWG_E_ARG;
    const int M = Fp * n_per_tri;
    DWG_LAUNCH("meshbind_fwd", k_meshbind_fwd, dim3(dwg_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream_, M, n_per_tri, bary, scale_params,
               verts_cnl, verts_obs, vnormals_obs, triangles, pos_cnl_out, pos_out, scales_out, quats_out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_meshbind_backward(int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params, const float* verts_cnl,
                          const float* verts_obs, const float* vnormals_obs, const int32_t* triangles, const float* g_pos_cnl,
                          const float* g_pos, const float* g_scales, const float* g_quats, float* g_bary, float* g_scale_params,
                          dwg_stream_t stream_) {
    if (Fp < 0 || n_per_tri <= 0) return DWG_E_ARG;
    if (Fp == 0) return DWG_OK;
    if (!bary || !scale_params || !verts_obs || !vnormals_obs || !triangles || !g_bary || !g_scale_params) return DWG_E_ARG;
    if (g_pos_cnl && !verts_cnl) return DWG_E_ARG;
    const int M = Fp * n_per_tri;
    DWG_LAUNCH("meshbind_bwd", k_meshbind_bwd, dim3(dwg_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream_, M, n_per_tri, bary, scale_params,
               verts_cnl, verts_obs, vnormals_obs, triangles, g_pos_cnl, g_pos, g_scales, g_quats, g_bary, g_scale_params,
               (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_meshbind_backward_verts(int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params, const float* verts_cnl,
                                const float* verts_obs, const float* vnormals_obs, const int32_t* triangles, const float* g_pos_cnl,
                                const float* g_pos, const float* g_scales, const float* g_quats, float* g_bary, float* g_scale_params,
                                float* g_verts_cnl, float* g_verts_obs, float* g_vnormals_obs, dwg_stream_t stream_) {
    if (Fp < 0 || n_per_tri <= 0) return DWG_E_ARG;
    if (Fp == 0) return DWG_OK;
    if (!bary || !scale_params || !verts_obs || !vnormals_obs || !triangles || !g_bary || !g_scale_params) return DWG_E_ARG;
    if (!g_verts_obs || !g_vnormals_obs) return DWG_E_ARG;
    if (g_pos_cnl && !verts_cnl) return DWG_E_ARG;
    const int M = Fp * n_per_tri;
    DWG_LAUNCH("meshbind_bwd", k_meshbind_bwd, dim3(dwg_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream_, M, n_per_tri, bary, scale_params,
               verts_cnl, verts_obs, vnormals_obs, triangles, g_pos_cnl, g_pos, g_scales, g_quats, g_bary, g_scale_params, g_verts_cnl,
               g_verts_obs, g_vnormals_obs, (float*)nullptr);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_meshbind_backward_verts_gather(int32_t Vp, int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params,
                                       const float* verts_cnl, const float* verts_obs, const float* vnormals_obs, const int32_t* triangles,
                                       const int32_t* vf_offsets, const int32_t* vf_faces, const float* g_pos_cnl, const float* g_pos,
                                       const float* g_scales, const float* g_quats, float* g_bary, float* g_scale_params, float* corner_rows,
                                       float* g_verts_cnl, float* g_verts_obs, float* g_vnormals_obs, dwg_stream_t stream_) {
    if (Vp < 0 || Fp < 0 || n_per_tri <= 0) return DWG_E_ARG;
    if (Fp == 0 || Vp == 0) return DWG_OK;
    if (!bary || !scale_params || !verts_obs || !vnormals_obs || !triangles || !g_bary || !g_scale_params) return DWG_E_ARG;
    if (!vf_offsets || !vf_faces || !corner_rows || !g_verts_obs || !g_vnormals_obs) return DWG_E_ARG;
    if (g_pos_cnl && !verts_cnl) return DWG_E_ARG;
    if (g_verts_cnl && !(verts_cnl && g_pos_cnl)) return DWG_E_ARG;
    const int M = Fp * n_per_tri;
    hipStream_t stream = (hipStream_t)stream_;
    DWG_LAUNCH("meshbind_bwd", k_meshbind_bwd, dim3(dwg_cdiv(M, 256)), dim3(256), 0, stream, M, n_per_tri, bary, scale_params,
               verts_cnl, verts_obs, vnormals_obs, triangles, g_pos_cnl, g_pos, g_scales, g_quats, g_bary, g_scale_params, (float*)nullptr,
               (float*)nullptr, (float*)nullptr, corner_rows);
    DWG_LAUNCH("meshbind_gather_verts", k_meshbind_gather_verts, dim3(dwg_cdiv(Vp, 256)), dim3(256), 0, stream, Vp, n_per_tri, triangles,
               vf_offsets, vf_faces, (const float*)corner_rows, g_verts_cnl, g_verts_obs, g_vnormals_obs);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_mesh_vertex_normals_backward(int32_t Vp, int32_t Fp, const float* verts, const int32_t* triangles, const int32_t* vf_offsets,
                                     const int32_t* vf_faces, const float* g_vertex_normals, float* face_normals_scratch,
                                     float* g_sum_scratch, float* g_verts, dwg_stream_t stream_) {
    if (Vp < 0 || Fp < 0) return DWG_E_ARG;
    if (Vp == 0 || Fp == 0) return DWG_OK;
    if (!verts || !triangles || !vf_offsets || !vf_faces || !g_vertex_normals || !face_normals_scratch || !g_sum_scratch || !g_verts)
        return DWG_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    DWG_LAUNCH("mesh_face_normals", k_face_normals, dim3(dwg_cdiv(Fp, 256)), dim3(256), 0, stream, Fp, verts, triangles, face_normals_scratch);
    DWG_LAUNCH("mesh_vertex_normals_bwd", k_vertex_normals_bwd_sum, dim3(dwg_cdiv(Vp, 256)), dim3(256), 0, stream, Vp,
               (const float*)face_normals_scratch, vf_offsets, vf_faces, g_vertex_normals, g_sum_scratch);
    // (round 6: a per-vertex gather over the incident-face table instead of per-face float atomics: the same bits on every run)
    DWG_LAUNCH("mesh_face_normals_bwd", k_face_normals_bwd_gather, dim3(dwg_cdiv(Vp, 256)), dim3(256), 0, stream, Vp, verts, triangles, vf_offsets,
               vf_faces, (const float*)g_sum_scratch, g_verts);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
