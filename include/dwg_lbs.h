/*
 * include/dwg_lbs.h -- C-ABI of the SMPL-X linear-blend-skinning stage (boundary B3).
 *
 * The reference keeps this stage in Python objects:
 *   GeneralLinearBlendSkinning.forward / get_full_transform   /root/reference/core/human/inverse_lbs.py:652-784
 *   RigidTransform.transform_points / transform_quaternions   /root/reference/core/human/inverse_lbs.py:190-251
 *   DreamWaltzG.lbs_transform                                 /root/reference/core/system/avatar.py:1426-1462
 * The Python mirror (dreamwaltz-g_amd/lbs.py) keeps those call signatures and routes the arithmetic here.
 * All pointers are device pointers to dense fp32 row-major tensors (int32 for indices).
 */
#ifndef DWG_LBS_H
#define DWG_LBS_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Rodrigues + kinematic chain + rest-pose removal (+ global translation):
 *   A_out[j] = compose(J_pose_rigid, G_transl_offset)[j]  -- the `joint_pose_transform` of avatar.py:1441-1444.
 * pose [J,3] axis-angle (the 165-vector of inverse_lbs.py:611-624 after `+= pose_mean`), joints [J,3] (rest joints of
 * the shaped template, inverse_lbs.py:681), parents [J] with parents[0] = -1, transl [3] or NULL.
 * When joint_shape_dirs is given, `joints` is the TEMPLATE joint set and the shaped joints are joints + dirs . shape
 * (vertices2joints(J_regressor, v_template + blend_shapes(...)) of inverse_lbs.py:676-681, re-associated).
 * rot_mats_out [J,9] (optional) are the per-joint rotation matrices needed for the pose blend-shape feature. J <= 64. */
int dwg_lbs_joint_chain(int32_t J, const float* pose, const float* joints, const int32_t* parents, const float* transl,
                        const float* joint_shape_dirs /*[J,3,S] = J_regressor . [shapedirs|expr_dirs], or NULL*/,
                        const float* shape_coeffs /*[S] = cat(betas, expression)*/, int32_t n_shape,
                        float* A_out /*[J,16]*/, float* rot_mats_out /*[J,9] or NULL*/, dwg_stream_t stream);

/* Per-Gaussian blend + transforms:  T_i = sum_j w_ij A_j ;  p' = R_i p + t_i ;
 * q' = matrix_to_quaternion(F R_i F quaternion_to_matrix(q)), F = diag(1,-1,-1)  (flip_rotation_axis=True).
 * weights [N,J]; when normalize_weights != 0 each row is divided by its sum first (LBSUtils.lbs_weight_activation,
 * avatar.py:913-918).  quats / quats_out may be NULL (points only, the canonical pass avatar.py:1517-1522).
 * T12_save [N,12] (optional) keeps the blended [R|t] rows for the backward. */
int dwg_lbs_blend_forward(int32_t N, int32_t J, int32_t normalize_weights, const float* A /*[J,16]*/,
                          const float* weights, const float* points, const float* quats, float* points_out,
                          float* quats_out, float* T12_save, dwg_stream_t stream);

/* Gradients w.r.t. points and quats (skinning weights and the skeleton are frozen by default: configs/__init__.py:197). */
int dwg_lbs_blend_backward(int32_t N, const float* T12 /*[N,12] from forward*/, const float* points, const float* quats,
                           const float* g_points_out, const float* g_quats_out, float* g_points, float* g_quats,
                           dwg_stream_t stream);

/* transform_V applied to a FIXED vertex subset (mesh-bound Gaussians, avatar.py:1570-1576):
 *   out_t = T_rigid(t) * (x_t + shapedirs_sub[t] . shape + posedirs_sub[t] . (rot_mats[1:] - I))
 * with T_rigid(t) = sum_j lbs_weights_sub[t,j] A_j (A already carries the translation).  The *_sub arrays are the rows of the
 * body model's shapedirs [V,3,S] / posedirs [F,3V] / lbs_weights [V,J] gathered once for the subset into vertex-major
 * [Vp,3,S] / [Vp,3,F] / [Vp,J]; either direction array may be NULL to skip that offset. */
int dwg_lbs_vertex_transform(int32_t Vp, int32_t J, int32_t n_shape, int32_t n_posefeat, const float* vertex_coords /*[Vp,3]*/,
                             const float* A, const float* lbs_weights_sub, const float* shapedirs_sub, const float* shape_coeffs,
                             const float* posedirs_sub, const float* rot_mats /*[J,9]*/, float* out /*[Vp,3]*/,
                             dwg_stream_t stream);

/* Gradient of dwg_lbs_vertex_transform's output w.r.t. the shape coefficients (`learn_hand_betas` / `learn_face_betas`,
 * avatar.py:1551-1553; scripts/train_w_expr.sh:66): both dependency paths -- the per-vertex blend-shape offset and the rest
 * joints J(beta) that enter every A_j's translation through the kinematic chain (smplx batch_rigid_transform).  The joint
 * rotations do not depend on the coefficients, so for a fixed pose the map is linear and this is exact.
 * g_shape [n_shape] is OVERWRITTEN; g_A_transl_scratch [J,3] is scratch; pose [J,3] / parents / joint_shape_dirs [J,3,S] are the
 * inputs dwg_lbs_joint_chain was called with. */
int dwg_lbs_vertex_transform_backward_shape(int32_t Vp, int32_t J, int32_t n_shape, const float* A, const float* lbs_weights_sub,
                                            const float* shapedirs_sub, const float* g_out /*[Vp,3]*/, const float* pose,
                                            const int32_t* parents, const float* joint_shape_dirs, float* g_A_transl_scratch,
                                            float* g_shape, dwg_stream_t stream);

/* The same gradient with a caller-owned workspace of dwg_lbs_vertex_transform_backward_shape_workspace_floats(Vp) floats (one row of
 * partial sums per workgroup): what the Python path calls.  No float atomics in either form (round 6): the sums are formed in a fixed order,
 * the same bits on every run; the form above keeps one library-owned workspace per process and serialises its callers on it. */
size_t dwg_lbs_vertex_transform_backward_shape_workspace_floats(int32_t Vp);
int dwg_lbs_vertex_transform_backward_shape_ws(int32_t Vp, int32_t J, int32_t n_shape, const float* A, const float* lbs_weights_sub,
                                               const float* shapedirs_sub, const float* g_out, const float* pose, const int32_t* parents,
                                               const float* joint_shape_dirs, float* g_A_transl_scratch, float* g_shape, float* workspace,
                                               dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
