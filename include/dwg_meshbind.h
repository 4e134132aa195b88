/*
 * include/dwg_meshbind.h -- C-ABI of the mesh-bound Gaussians (hands / face), SURVEY.md section 8a rows L14-L15.
 *
 * Replaces the PyTorch op chains of
 *   MeshBindingGaussianModel.get_positions               /root/reference/core/system/avatar.py:1016-1025
 *   MeshBindingGaussianModel.get_scales_and_quaternions  /root/reference/core/system/avatar.py:1027-1079
 *   compute_normal                                       /root/reference/core/utils/mesh.py:34-94
 * with one forward and one backward launch (the reference spends ~150 small kernels + their autograd here).
 * All pointers are device pointers to dense row-major tensors: fp32 data, int32 indices.  M = Fp * n_per_tri points,
 * point i belongs to triangle i / n_per_tri (the `_points_to_vertices` expansion of avatar.py:995-1003).
 */
#ifndef DWG_MESHBIND_H
#define DWG_MESHBIND_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Vertex normals of the posed sub-mesh: face normals safe_normalize((v1-v0) x (v2-v0)), summed over the faces incident to
 * each vertex, (0,0,1) where the sum vanishes, safe-normalised (eps 1e-20 on the squared norm).  The reference scatters with
 * three index_add_ calls; here each vertex GATHERS its faces through a CSR adjacency (vf_offsets[Vp+1], vf_faces[3*Fp],
 * built once on the host: the topology is static), which makes the sum order fixed.  face_normals [Fp,3] is scratch. */
int dwg_mesh_vertex_normals(int32_t Vp, int32_t Fp, const float* verts /*[Vp,3]*/, const int32_t* triangles /*[Fp,3]*/,
                            const int32_t* vf_offsets, const int32_t* vf_faces, float* face_normals, float* vertex_normals /*[Vp,3]*/,
                            dwg_stream_t stream);

/* positions (sum-normalised barycentric blend), scales (0, tangent extents / n_per_tri * clamp(scale_param, 0.5, 2)) and
 * standardised frame quaternions of all M points; optionally also the canonical-pose positions (verts_cnl / pos_cnl_out may
 * both be NULL).  bary [Fp,n_per_tri,3] RAW learnable coordinates, scale_params [M,3]. */
int dwg_meshbind_forward(int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params, const float* verts_cnl,
                         const float* verts_obs /*[Vp,3]*/, const float* vnormals_obs /*[Vp,3]*/, const int32_t* triangles,
                         float* pos_cnl_out /*[M,3]*/, float* pos_out /*[M,3]*/, float* scales_out /*[M,3]*/,
                         float* quats_out /*[M,4]*/, dwg_stream_t stream);

/* Gradients w.r.t. bary and scale_params (vertices and normals are produced under no_grad: avatar.py:1570-1577).
 * Any of the incoming gradients may be NULL (treated as zero).  Both outputs are fully written. */
int dwg_meshbind_backward(int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params, const float* verts_cnl,
                          const float* verts_obs, const float* vnormals_obs, const int32_t* triangles, const float* g_pos_cnl,
                          const float* g_pos, const float* g_scales, const float* g_quats, float* g_bary /*[Fp,n,3]*/,
                          float* g_scale_params /*[M,3]*/, dwg_stream_t stream);

/* Same as dwg_meshbind_backward and additionally ACCUMULATES (buffers zeroed by the caller) the gradients w.r.t. the posed
 * vertices, their vertex normals and the canonical vertices -- needed when the vertices depend on a learnable parameter
 * (`learn_hand_betas` / `learn_face_betas`: avatar.py:1551-1577, scripts/train_w_expr.sh:66).  g_verts_cnl may be NULL.  (This form scatters with float atomics -- sums in arrival order --
 * and is kept for C callers of the round-2 signature; the Python path calls dwg_meshbind_backward_verts_gather below.) */
int dwg_meshbind_backward_verts(int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params, const float* verts_cnl,
                                const float* verts_obs, const float* vnormals_obs, const int32_t* triangles, const float* g_pos_cnl,
                                const float* g_pos, const float* g_scales, const float* g_quats, float* g_bary, float* g_scale_params,
                                float* g_verts_cnl /*[Vp,3]*/, float* g_verts_obs /*[Vp,3]*/, float* g_vnormals_obs /*[Vp,3]*/,
                                dwg_stream_t stream);

/* dwg_meshbind_backward_verts WITHOUT float atomics (round 6; `learn_hand_betas` is on in sub-stage 2.1 of the shipped recipe,
 * scripts/train_w_expr.sh:66): the per-Gaussian kernel writes one row {g position, g normal, g canonical position} per (Gaussian, corner)
 * into `corner_rows` (Fp * n_per_tri * 27 floats) and a per-vertex pass adds the rows of the vertex's incident faces (vf_offsets /
 * vf_faces: the table dwg_mesh_vertex_normals takes) in table order.  g_verts_* [Vp,3] are OVERWRITTEN (g_verts_cnl may be NULL);
 * dwg_mesh_vertex_normals_backward then accumulates into g_verts_obs, also by a per-vertex gather.  The same bits on every run. */
int dwg_meshbind_backward_verts_gather(int32_t Vp, int32_t Fp, int32_t n_per_tri, const float* bary, const float* scale_params,
                                       const float* verts_cnl, const float* verts_obs, const float* vnormals_obs, const int32_t* triangles,
                                       const int32_t* vf_offsets, const int32_t* vf_faces, const float* g_pos_cnl, const float* g_pos,
                                       const float* g_scales, const float* g_quats, float* g_bary, float* g_scale_params, float* corner_rows,
                                       float* g_verts_cnl, float* g_verts_obs, float* g_vnormals_obs, dwg_stream_t stream);

/* Backward of dwg_mesh_vertex_normals (autograd through compute_normal, utils/mesh.py:34-94): adds d loss / d verts into g_verts
 * [Vp,3] given d loss / d vertex_normals.  face_normals_scratch [Fp,3] and g_sum_scratch [Vp,3] are overwritten. */
int dwg_mesh_vertex_normals_backward(int32_t Vp, int32_t Fp, const float* verts, const int32_t* triangles, const int32_t* vf_offsets,
                                     const int32_t* vf_faces, const float* g_vertex_normals, float* face_normals_scratch,
                                     float* g_sum_scratch, float* g_verts, dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
