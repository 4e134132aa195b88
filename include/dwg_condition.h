/* dwg_condition.h -- OpenPose-style condition image of a posed SMPL-X body, on the GPU.
 *
 * Replaces, for condition_type 'pose' / 'openpose', what the reference's data layer does per step on the CPU:
 *   /root/reference/core/human/smpl_condition.py:191-235  SMPL2Condition.export_pose   (projection, invisibility, draw)
 *   /root/reference/core/human/smpl_condition.py:82-143   OcclusionCulling.__call__    (one ray per keypoint against the body mesh;
 *                                                          the reference builds an open3d BVH over 20 908 triangles every step)
 *   /root/reference/core/human/smpl_condition.py:20-79    to_controlnet_pose           (x / W, y / H, dist rows; None = missing)
 *   /root/reference/core/human/open_pose.py:48-333        draw_bodypose / draw_handpose / draw_facepose / adaptive_draw_poses
 *   /root/reference/utils/open3d.py:8-19                  build_ray_casting_scene      (here: the mesh is used as it lies in HBM)
 * and the PIL -> LANCZOS(identity at equal size) -> float / 255 -> NCHW conversion of core/guidance/controlnet.py:33-55 when the
 * float output is requested.
 *
 * Keypoint order (smpl_condition.py:22): body 18, left hand 21, right hand 21, face landmarks 51, face contour 17 = 128.
 * All pointers are device pointers; nothing is allocated inside; launches go to `stream`.
 */
#ifndef DWG_CONDITION_H
#define DWG_CONDITION_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif

#define DWG_COND_KEYPOINTS 128
#define DWG_COND_DRAW_BODY 1
#define DWG_COND_DRAW_HAND 2
#define DWG_COND_DRAW_FACE 4
#define DWG_COND_FLIP_LR 8

/* export_pose up to the drawing call.  keypoints [K,3] world (fp32), extrinsic [4,4] row-major world->camera, intrinsics [3,3]
 * already adjusted to the condition size (adjust_intrinsics_size), mesh vertices [V,3] fp32 + triangles [F,3] int32,
 * groups [K] (0 body, 1 hand, 2 face -> threshold index).  rows [K,4] fp64 = (x / W, y / H, distance to the camera centre, valid);
 * (fp64 because the drawing code truncates x * W to a pixel: an fp32 row would move ~0.3 % of the keypoints by one pixel)
 * valid = 0 where the reference hands None to the drawing code (behind the camera, occluded, NaN).  Arithmetic in fp64 as numpy's;
 * the rays are rounded to fp32 as the reference rounds them for open3d. */
int dwg_condition_keypoints(int32_t K, const float* keypoints, const float* extrinsic, const float* intrinsics, int32_t V,
                            const float* vertices, int32_t F, const int32_t* triangles, const uint8_t* groups, float thres_body,
                            float thres_hand, float thres_face, int32_t use_occlusion_culling, double* rows, dwg_stream_t stream);

/* bytes of scratch dwg_condition_draw needs for an H x W image (limb spans) */
size_t dwg_condition_workspace_bytes(int32_t H, int32_t W);

/* adaptive_draw_poses for one person (hand_dist_thres = None) from the 128 rows above.  flags: DWG_COND_*.  Either output may be
 * NULL: out_u8 [H,W,3] RGB uint8 (the reference's wire format), out_chw [3,H,W] fp32 in [0,1] (what ControlNet consumes). */
int dwg_condition_draw(int32_t H, int32_t W, const double* rows, int32_t flags, uint8_t* out_u8, float* out_chw, void* workspace,
                       dwg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
