"""Generates the golden fixtures under tests/golden/ by importing the parts of the reference that import in this
container.  Run here only (needs /root/reference); the fixtures it writes are data
(inputs + expected outputs) and are what travels to the GPU box.

    python tests/golden/capture_golden.py

Inert / arithmetic stand-ins used at capture time (never shipped as product code):
  loguru                      -> logger stub (no arithmetic)
  smplx.lbs, pytorch3d.transforms -> oracle.animate restatements  ==> fixtures tagged "stub_dependent" only pin the
                                 reference's IN-REPO algebra (compose order, flip path, transform assembly), not the
                                 third-party functions themselves.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import animate as oa  # noqa: E402


def _stub_modules():
    lg = types.ModuleType("loguru")
    class _L:
        def __getattr__(self, k):
            return lambda *a, **k2: None
    lg.logger = _L()
    sys.modules["loguru"] = lg
    smplx = types.ModuleType("smplx")
    class SMPL:  # noqa
        pass
    class SMPLX(SMPL):  # noqa
        pass
    smplx.SMPL, smplx.SMPLX = SMPL, SMPLX
    lbs = types.ModuleType("smplx.lbs")
    lbs.blend_shapes = oa.blend_shapes
    lbs.vertices2joints = oa.vertices2joints
    lbs.batch_rodrigues = lambda r, dtype=None: oa.batch_rodrigues(r)
    lbs.batch_rigid_transform = lambda rm, j, parents, dtype=None: oa.batch_rigid_transform(rm, j, parents)
    smplx.lbs = lbs
    sys.modules["smplx"] = smplx; sys.modules["smplx.lbs"] = lbs
    p3 = types.ModuleType("pytorch3d"); p3t = types.ModuleType("pytorch3d.transforms")
    p3t.quaternion_to_matrix = oa.quaternion_to_matrix
    p3t.matrix_to_quaternion = oa.matrix_to_quaternion
    p3t.quaternion_multiply = oa.quaternion_multiply
    p3t.standardize_quaternion = oa.standardize_quaternion
    p3.transforms = p3t
    sys.modules["pytorch3d"] = p3; sys.modules["pytorch3d.transforms"] = p3t
    return SMPLX


def main():
    SMPLX = _stub_modules()
    out = {}
    g = torch.Generator().manual_seed(0)
    # ---- 1. SH colours (core/gaussian/gaussian_utils.py:12-17, spherical_harmonics.py:117-172): direct import
    from core.gaussian.gaussian_utils import get_colors
    sh = torch.randn(64, 16, 3, generator=g) * 0.4
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    for lv in (1, 2, 3, 4):
        out["sh_colors_l%d" % lv] = get_colors(sh, dirs, lv).numpy()
    out["sh_in"] = sh.numpy(); out["sh_dirs"] = dirs.numpy()
    # ---- 2. DeformNetwork (core/deformation/deform_model.py:61-143): import with loguru stub
    from core.deformation.deform_model import DeformNetwork
    torch.manual_seed(0)
    net = DeformNetwork(xyz_input_ch=32, D=4, W=64)
    assert sum(p.numel() for p in net.parameters()) == 19274
    x = torch.randn(40, 32, generator=g) * 0.1
    pose = torch.randn(1, 63, generator=g) * 0.3
    d_xyz, d_scale, d_rot = net(x, pose)
    for k, v in net.state_dict().items():
        out["deform." + k] = v.numpy()
    out["deform_x"] = x.numpy(); out["deform_pose"] = pose.numpy()
    out["deform_warp"] = d_xyz.detach().numpy(); out["deform_scaling"] = d_scale.detach().numpy()
    out["deform_rotation"] = d_rot.detach().numpy()
    # ---- 3. exponential LR schedule (core/optim/optim_utils.py:4-38)
    from core.optim.optim_utils import get_expon_lr_func
    f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=10000)
    steps = np.array([0, 1, 10, 100, 1000, 5000, 9999, 10000])
    out["lr_steps"] = steps; out["lr_values"] = np.array([f(int(s)) for s in steps], dtype=np.float64)
    # ---- 4. exp_se3 (core/deformation/rigid_utils.py:60-83)
    from core.deformation.rigid_utils import exp_se3
    S = torch.randn(9, 6, generator=g); th = torch.rand(9, 1, generator=g) + 0.1
    out["se3_S"] = S.numpy(); out["se3_theta"] = th.numpy(); out["se3_out"] = exp_se3(S, th).numpy()
    # ---- 5. RigidTransform algebra + GeneralLinearBlendSkinning (stub-dependent)
    from core.human.inverse_lbs import RigidTransform, GeneralLinearBlendSkinning
    body = oa.SyntheticBody(V=300, F_=500, seed=3)
    fake = SMPLX()
    fake.NUM_JOINTS = 54; fake.NUM_BODY_JOINTS = 21
    fake.faces = body.faces.numpy(); fake.parents = torch.from_numpy(body.parents)
    for k in ("betas", "v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "pose_mean", "expr_dirs",
              "expression", "jaw_pose", "leye_pose", "reye_pose"):
        setattr(fake, k, getattr(body, k))
    fake.body_pose = torch.zeros(1, 63); fake.global_orient = torch.zeros(1, 3)
    fake.left_hand_pose = torch.zeros(1, 45); fake.right_hand_pose = torch.zeros(1, 45)
    fake.use_pca = False; fake.left_hand_components = torch.zeros(1); fake.right_hand_components = torch.zeros(1)
    glbs = GeneralLinearBlendSkinning(fake)
    inp = oa.random_smpl_inputs(seed=5)
    jaw = torch.randn(1, 3, generator=g)  # must be ignored (checklist Q1)
    tJ, tV, tr = glbs.forward(**inp, jaw_pose=jaw)
    for k, v in inp.items():
        out["glbs_in." + k] = v.numpy()
    out["glbs_tJ"] = tJ.SE3.detach().numpy(); out["glbs_tV"] = tV.SE3.detach().numpy()
    for k, v in tr.items():
        out["glbs_tr." + k] = v.SE3.detach().numpy()
    N = 50
    w = torch.softmax(torch.randn(N, 55, generator=g), -1)
    pts = torch.randn(N, 3, generator=g) * 0.3
    q = torch.randn(N, 4, generator=g)
    jt = RigidTransform.compose(tr["J_pose_rigid"], tr["G_transl_offset"]).squeeze(0)
    out["rt_w"] = w.numpy(); out["rt_pts"] = pts.numpy(); out["rt_q"] = q.numpy()
    out["rt_points_weighted"] = jt.transform_points(pts, weights=w).detach().numpy()
    out["rt_quats_flip"] = jt.transform_quaternions(q, weights=w, flip_rotation_axis=True).detach().numpy()
    idx = torch.randint(0, 300, (N,), generator=g)
    out["rt_idx"] = idx.numpy()
    out["rt_points_indexed"] = tV.squeeze(0).transform_points(pts, indices=idx).detach().numpy()
    out["rt_inverse"] = RigidTransform(SE3=tr["J_pose_rigid"].SE3.clone()).inverse().SE3.detach().numpy()
    out["body_seed"] = np.array([3]); out["body_V"] = np.array([300]); out["body_F"] = np.array([500])
    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_golden.npz"), "keys:", len(out))


if __name__ == "__main__":
    main()
