"""Captures golden vectors for the OpenPose condition-image path from the IMPORTED reference (runs in the build container only;
/root/reference does not exist on the GPU box).  Output: tests/golden/reference_golden_r2_condition.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_golden_condition.py

What runs is the reference's own code: SMPL2Condition.export_pose, OcclusionCulling.__call__, to_controlnet_pose,
adjust_intrinsics_size, transform_keypoints_to_novelview, project_camera3d_to_2d, SE3_Mat2RT.  Two third-party pieces are absent
here and replaced at their call boundary (keys they influence are still the reference's arithmetic on the stand-in's output):
  * open3d RaycastingScene.cast_rays -> oracle.condition.ray_cast (its t_hit is saved as an INPUT of the fixture: `*.t_hit`);
  * cv2 drawing -> not run: `draw_poses` is replaced by a recorder, the fixture holds the PoseResult rows it was called with.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import _ref_stubs  # noqa: E402
from oracle import animate as oa, condition as oc  # noqa: E402

_ref_stubs.install(oa)
sys.path.insert(0, "/root/reference")
import core.human.smpl_condition as sc  # noqa: E402
from data.camera.utils import adjust_intrinsics_size  # noqa: E402


class _T:
    def __init__(self, a):
        self.a = a

    def numpy(self):
        return self.a


class FakeScene:
    """stands where the open3d RaycastingScene stands; remembers what it answered"""

    def __init__(self, vertices, triangles):
        self.v, self.t = vertices, triangles
        self.t_hit = None

    def cast_rays(self, rays):
        rays = np.asarray(rays, dtype=np.float32)
        N, K, _ = rays.shape
        out = np.zeros((N, K), dtype=np.float32)
        for n in range(N):
            out[n] = oc.ray_cast(rays[n, 0, :3], rays[n, :, 3:], self.v, self.t).astype(np.float32)
        self.t_hit = out
        return {"t_hit": _T(out), "geometry_ids": _T(np.where(np.isinf(out), 2 ** 32 - 1, 0).astype(np.int64))}    # one mesh: id 0, miss: INVALID_ID


def ellipsoid_mesh(nu=24, nv=16, radii=(0.22, 0.45, 0.14), centre=(0.0, 0.0, 0.0)):
    us = np.linspace(0, 2 * np.pi, nu, endpoint=False); vs = np.linspace(0, np.pi, nv + 1)
    verts = np.array([[radii[0] * np.sin(v) * np.cos(u) + centre[0], radii[1] * np.cos(v) + centre[1], radii[2] * np.sin(v) * np.sin(u) + centre[2]]
                      for v in vs for u in us])
    tris = []
    for i in range(nv):
        for j in range(nu):
            a, b = i * nu + j, i * nu + (j + 1) % nu
            c, d = a + nu, b + nu
            tris += [(a, c, b), (b, c, d)]
    return verts.astype(np.float64), np.array(tris, dtype=np.int64)


def look_at_extrinsic(azimuth_deg, elevation_deg, radius):
    """world -> camera, OpenCV axes (x right, y down, z forward), camera on a sphere looking at the origin."""
    az, el = np.radians(azimuth_deg), np.radians(elevation_deg)
    pos = radius * np.array([np.sin(el) * np.sin(az), np.cos(el), np.sin(el) * np.cos(az)])
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0])); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)
    E = np.eye(4); E[:3, :3] = R; E[:3, 3] = -R @ pos
    return E


def main():
    out = {}
    rng = np.random.default_rng(7)
    verts, tris = ellipsoid_mesh()
    # 128 keypoints: on / slightly outside the surface all around (so roughly half are hidden from any view), a few far off
    idx = rng.integers(0, verts.shape[0], 128)
    kp = verts[idx] * (1.0 + rng.uniform(0.0, 0.25, (128, 1)))
    kp[5] = [0.0, 0.0, 6.0]; kp[40] = [0.3, -0.2, -7.0]            # behind one of the cameras
    cases = [("front", 0.0, 80.0, 2.5, 512, 512, 1.2, False), ("side", 100.0, 70.0, 2.0, 512, 512, 1.0, False),
             ("wide", 215.0, 95.0, 3.0, 384, 256, 1.5, False), ("side_ignore_body", 100.0, 70.0, 2.0, 512, 512, 1.0, True)]
    cond = object.__new__(sc.SMPL2Condition)
    cond.draw_body = cond.draw_hand = cond.draw_face = True
    cond.openpose_left_right_flip = False
    for name, az, el, rad, W, H, f, ignore in cases:
        cond.occlusion_culling = sc.OcclusionCulling("smplx", ignore)
        E = look_at_extrinsic(az, el, rad)
        K_raw = np.array([[f * 512, 0.0, 256.0], [0.0, f * 512, 256.0], [0.0, 0.0, 1.0]])
        K = adjust_intrinsics_size(K_raw.copy(), width=W, height=H)
        scene = FakeScene(verts, tris)
        rec = {}

        def recorder(poses, H, W, **kw):
            rec["poses"], rec["kw"], rec["HW"] = poses, kw, (H, W)
            return np.zeros((H, W, 3), dtype=np.uint8)
        sc.draw_poses = recorder
        img = cond.export_pose(kp[None].copy(), scene, extrinsic=E.copy(), intrinsics=K.copy(), width=W, height=H)
        assert img.size == (W, H)
        pose = rec["poses"][0]
        rows = np.full((128, 3), np.nan)
        allk = list(pose.body.keypoints) + list(pose.left_hand) + list(pose.right_hand) + list(pose.face)
        assert len(allk) == 128
        for i, k in enumerate(allk):
            if k is not None:
                rows[i] = [k.x, k.y, k.dist]
        p = "cond.%s." % name
        out[p + "extrinsic"], out[p + "intrinsics_raw"], out[p + "intrinsics"] = E, K_raw, K
        out[p + "size"] = np.array([W, H])
        out[p + "t_hit"] = scene.t_hit[0]
        out[p + "rows"] = rows
        out[p + "draw_kwargs"] = np.array(sorted("%s=%s" % kv for kv in rec["kw"].items()))
        # the occlusion rule on its own
        R, T = E[:3, :3], E[:3, 3:4]
        center = np.dot(np.linalg.inv(R), -T)
        occ, tfar = cond.occlusion_culling(center=center, keypoints=kp[None].copy(), ray_casting_scene=FakeScene(verts, tris))
        out[p + "occluded"], out[p + "t_far"] = occ[0], tfar[0]
    out["cond.keypoints"], out["cond.vertices"], out["cond.triangles"] = kp, verts, tris
    oc_ = sc.OcclusionCulling("smplx")
    out["cond.face_indices"], out["cond.hand_indices"], out["cond.body_indices"] = (np.array(oc_.face_indices), np.array(oc_.hand_indices),
                                                                                    np.array(oc_.body_indices))
    path = os.path.join(HERE, "reference_golden_r2_condition.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays;", {c[0]: int(np.isnan(out["cond.%s.rows" % c[0]][:, 0]).sum()) for c in cases}, "missing keypoints")


if __name__ == "__main__":
    main()
