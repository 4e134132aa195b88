"""OpenPose-style condition image of the posed body ON THE GPU -- host-side mirror of the reference's
`core/human/smpl_condition.py` (`SMPL2Condition`, `OcclusionCulling`) and `utils/open3d.py:8-19` (`build_ray_casting_scene`) for
condition types 'pose' / 'openpose', over `csrc/condition.hip` (include/dwg_condition.h).

Same names, arguments and error behaviour as the reference seam; what differs, deliberately:
  * the "ray casting scene" is the mesh itself as it lies in HBM (no BVH is built for a mesh that moves every step);
  * `export_pose` returns a CUDA uint8 tensor [H, W, 3] (RGB, the reference's wire format) instead of a PIL image --
    `ConditionImage.to_pil()` gives the PIL image where one is wanted, `export_pose_chw` gives the float [1, 3, H, W] in [0, 1]
    that `ControlNetScoreDistillation.prepare_image` would make of it (controlnet.py:33-55 at equal size is the identity resize);
  * nothing here leaves the device: no `.cpu()`, no host synchronisation.
There is no CPU fallback: CPU tensors raise."""
import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib

N_KEYPOINTS = 128                 # body 18 + hands 2 x 21 + face 51 + 17 (smpl_condition.py:22)
DRAW_BODY, DRAW_HAND, DRAW_FACE, FLIP_LR = 1, 2, 4, 8


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class RayCastingScene:
    """Stands where open3d's RaycastingScene stands: vertices [V,3] fp32 + triangles [F,3] int32 on the device."""

    def __init__(self, vertices: torch.Tensor, triangles: torch.Tensor):
        if not vertices.is_cuda:
            raise RuntimeError("dreamwaltz_g_amd.condition runs on the GPU only (HIP kernels); got a CPU tensor")
        self.vertices = vertices.detach().reshape(-1, 3).float().contiguous()
        self.triangles = triangles.to(device=vertices.device, dtype=torch.int32).reshape(-1, 3).contiguous()


def build_ray_casting_scene(vertices, triangles) -> RayCastingScene:
    """utils/open3d.py:8-19 (one mesh; [1,V,3] or [V,3] vertices; triangles as a tensor or numpy array)."""
    if not torch.is_tensor(triangles):
        triangles = torch.as_tensor(np.asarray(triangles).astype(np.int32))
    if vertices.dim() == 3:
        if vertices.shape[0] != 1:
            raise NotImplementedError("one person per condition image")
        vertices = vertices[0]
    return RayCastingScene(vertices, triangles)


def adjust_intrinsics_size(intrinsics: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """data/camera/utils.py:233-242 on a copy ([..., 3, 3])."""
    k = intrinsics.clone()
    width_raw, height_raw = k[..., 0, 2] * 2, k[..., 1, 2] * 2
    k[..., 0, 0] = k[..., 0, 0] * (width / width_raw)
    k[..., 1, 1] = k[..., 1, 1] * (height / height_raw)
    k[..., 0, 2] = width / 2
    k[..., 1, 2] = height / 2
    return k


class OcclusionCulling:
    """smpl_condition.py:82-143: the keypoint groups and their thresholds; the rays are cast inside dwg_condition_keypoints."""

    def __init__(self, smpl_type: str, ignore_body_self_occlusion: bool = False) -> None:
        if smpl_type == 'smpl':
            self.face_indices = [0, 14, 15, 16, 17]
            self.hand_indices = []
            self.body_indices = [i for i in range(18) if i not in self.face_indices]
        elif smpl_type == 'smplx':
            self.face_indices = [0, 14, 15, 16, 17] + [i for i in range(18 + 21 * 2, 128)]
            self.hand_indices = [i for i in range(18, 18 + 21 * 2)]
            self.body_indices = [i for i in range(128) if (i not in self.face_indices) and (i not in self.hand_indices)]
        else:
            raise NotImplementedError
        # One person per image: every ray that hits anything hits that person's own mesh, so ignoring body SELF occlusion
        # (:132-135, the shipped default configs/__init__.py:445) means the body group is never culled.
        self.ignore_body_self_occlusion = ignore_body_self_occlusion
        self.thres_body, self.thres_face, self.thres_hand = 0.2, 0.02, 0.2          # __call__ defaults, :100-103
        self._groups = {}

    def groups(self, K: int, device) -> torch.Tensor:
        key = (K, str(device))
        if key not in self._groups:
            g = torch.zeros(K, dtype=torch.uint8)
            g[[i for i in self.hand_indices if i < K]] = 1
            g[[i for i in self.face_indices if i < K]] = 2
            self._groups[key] = g.to(device)
        return self._groups[key]


class ConditionImage:
    """uint8 [H, W, 3] RGB on the device, with the conversions the consumers of the reference's PIL image use."""

    def __init__(self, u8: torch.Tensor):
        self.u8 = u8

    def to_pil(self):
        from PIL import Image
        return Image.fromarray(self.u8.cpu().numpy())

    def to_chw(self) -> torch.Tensor:
        return (self.u8.permute(2, 0, 1).float() / 255.0).unsqueeze(0)


class SMPL2Condition:
    """smpl_condition.py:145-320 for condition_type 'pose' / 'openpose'.  `cfg` carries draw_body_keypoints, draw_hand_keypoints,
    draw_face_landmarks, openpose_left_right_flip, use_occlusion_culling, smpl_type, ignore_body_self_occlusion (PromptConfig)."""

    def __init__(self, cfg) -> None:
        self.draw_body = cfg.draw_body_keypoints
        self.draw_hand = cfg.draw_hand_keypoints
        self.draw_face = cfg.draw_face_landmarks
        self.openpose_left_right_flip = cfg.openpose_left_right_flip
        if cfg.use_occlusion_culling:
            self.occlusion_culling = OcclusionCulling(cfg.smpl_type, cfg.ignore_body_self_occlusion)
        else:
            self.occlusion_culling = None
        self._ws = {}

    # -- the two launches ------------------------------------------------------------------------------------------------
    def pose_rows(self, keypoints: torch.Tensor, ray_casting_scene: Optional[RayCastingScene], extrinsic: torch.Tensor,
                  intrinsics: torch.Tensor) -> torch.Tensor:
        """export_pose up to the drawing call (smpl_condition.py:191-224): fp64 rows [K,4] = (x / W, y / H, dist, valid)."""
        if not keypoints.is_cuda:
            raise RuntimeError("dreamwaltz_g_amd.condition runs on the GPU only (HIP kernels); got a CPU tensor")
        if keypoints.dim() == 3:
            if keypoints.shape[0] != 1:
                raise NotImplementedError("one person per condition image")
            keypoints = keypoints[0]
        dev = keypoints.device
        kp = keypoints.detach().float().contiguous()
        K = int(kp.shape[0])
        ext = extrinsic.detach().to(dev).float().reshape(4, 4).contiguous()
        intr = intrinsics.detach().to(dev).float().reshape(3, 3).contiguous()
        rows = torch.empty(K, 4, dtype=torch.float64, device=dev)
        cull = self.occlusion_culling is not None
        oc = self.occlusion_culling
        if cull and ray_casting_scene is None:
            raise ValueError("occlusion culling needs the body mesh (build_ray_casting_scene)")
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_condition_keypoints(
            K, p(kp), p(ext), p(intr), int(ray_casting_scene.vertices.shape[0]) if cull else 0,
            p(ray_casting_scene.vertices) if cull else None, int(ray_casting_scene.triangles.shape[0]) if cull else 0,
            p(ray_casting_scene.triangles) if cull else None, p(oc.groups(K, dev)) if cull else None,
            (float('inf') if oc.ignore_body_self_occlusion else oc.thres_body) if cull else 0.0, oc.thres_hand if cull else 0.0, oc.thres_face if cull else 0.0, 1 if cull else 0,
            p(rows), _stream(dev)), "dwg_condition_keypoints")
        return rows

    def draw(self, rows: torch.Tensor, height: int, width: int, out_u8: bool = True, out_chw: bool = False):
        """adaptive_draw_poses (open_pose.py:279-333) from the rows -> (uint8 [H,W,3] | None, float [1,3,H,W] | None)."""
        if rows.shape[0] != N_KEYPOINTS:
            raise ValueError("the OpenPose layout has 128 keypoints (body 18, hands 2 x 21, face 68); got %d" % rows.shape[0])
        dev = rows.device
        rows = rows.double().contiguous()
        key = (str(dev), height, width)
        if key not in self._ws:
            self._ws[key] = torch.empty(max(int(_lib.lib().dwg_condition_workspace_bytes(height, width)), 16), dtype=torch.uint8, device=dev)
        flags = ((DRAW_BODY if self.draw_body else 0) | (DRAW_HAND if self.draw_hand else 0) | (DRAW_FACE if self.draw_face else 0)
                 | (FLIP_LR if self.openpose_left_right_flip else 0))
        u8 = torch.empty(height, width, 3, dtype=torch.uint8, device=dev) if out_u8 else None
        chw = torch.empty(1, 3, height, width, dtype=torch.float32, device=dev) if out_chw else None
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_condition_draw(height, width, p(rows), flags, p(u8) if out_u8 else None, p(chw) if out_chw else None,
                                                 p(self._ws[key]), _stream(dev)), "dwg_condition_draw")
        return u8, chw

    # -- the reference's methods -----------------------------------------------------------------------------------------
    def export_pose(self, keypoints, ray_casting_scene, **camera_params) -> ConditionImage:
        """smpl_condition.py:191-235; camera_params: extrinsic [4,4], intrinsics [3,3] (size-adjusted), width, height."""
        rows = self.pose_rows(keypoints, ray_casting_scene, camera_params['extrinsic'], camera_params['intrinsics'])
        u8, _ = self.draw(rows, camera_params['height'], camera_params['width'])
        return ConditionImage(u8)

    def export_pose_chw(self, keypoints, ray_casting_scene, **camera_params) -> torch.Tensor:
        """The same image as the float [1,3,H,W] in [0,1] ControlNet consumes (no uint8 / PIL round trip)."""
        rows = self.pose_rows(keypoints, ray_casting_scene, camera_params['extrinsic'], camera_params['intrinsics'])
        return self.draw(rows, camera_params['height'], camera_params['width'], out_u8=False, out_chw=True)[1]

    def __call__(self, smpl_outputs, triangles, camera_params: dict, condition_type: str, condition_height: int,
                 condition_width: int) -> ConditionImage:
        """smpl_condition.py:271-320 (numpy-style branch, 'pose' / 'openpose')."""
        if condition_type not in ('pose', 'openpose'):
            raise NotImplementedError("condition_type %r: only the OpenPose skeleton image is on the SDS hot path" % (condition_type,))
        extrinsic = camera_params['extrinsic'][0]
        intrinsics = camera_params['intrinsics'][0]
        assert extrinsic.dim() == 2 and extrinsic.numel() == 16
        assert intrinsics.dim() == 2 and intrinsics.numel() == 9
        intrinsics = adjust_intrinsics_size(intrinsics, width=condition_width, height=condition_height)
        scene = build_ray_casting_scene(smpl_outputs.vertices.detach(), triangles) if self.occlusion_culling is not None else None
        return self.export_pose(smpl_outputs.joints.detach(), scene, intrinsics=intrinsics, extrinsic=extrinsic, width=condition_width,
                                height=condition_height)
