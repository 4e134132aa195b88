"""Pins the CPU oracle against golden vectors captured from the importable parts of the reference
(tests/golden/capture_golden.py)."""
import os

import numpy as np
import torch

from oracle import animate as oa
from oracle import raster as oraster

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz"))


def test_sh_colors_match_reference_eval_sh():
    sh, dirs = G["sh_in"], G["sh_dirs"]
    for lv in (1, 2, 3, 4):
        ref = G["sh_colors_l%d" % lv]
        got = oraster.sh_colors(sh[:, :lv * lv].copy(), dirs, np.zeros(3, np.float32), lv - 1)
        assert np.abs(got - ref).max() < 2e-6, lv


def test_deform_network_matches_reference():
    p = {k[len("deform."):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("deform.")}
    warp, scaling, rotation = oa.deform_forward(torch.from_numpy(G["deform_x"]), torch.from_numpy(G["deform_pose"]), p)
    assert torch.allclose(warp, torch.from_numpy(G["deform_warp"]), atol=1e-6)
    assert torch.allclose(scaling, torch.from_numpy(G["deform_scaling"]), atol=1e-6)
    assert torch.allclose(rotation, torch.from_numpy(G["deform_rotation"]), atol=1e-6)


def test_glbs_and_rigid_transform_algebra_match_reference():
    """stub-dependent fixture: pins compose order, jaw-pose-ignored bug, flip path, transform assembly."""
    body = oa.SyntheticBody(V=int(G["body_V"][0]), F_=int(G["body_F"][0]), seed=int(G["body_seed"][0]))
    inp = {k[len("glbs_in."):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("glbs_in.")}
    tJ, tV, tr = oa.glbs_forward(body, **inp)
    assert torch.allclose(tJ, torch.from_numpy(G["glbs_tJ"]), atol=1e-6)
    assert torch.allclose(tV, torch.from_numpy(G["glbs_tV"]), atol=1e-6)
    for k in ("V_shape_offset", "V_pose_offset", "V_pose_rigid", "J_shape_offset", "J_pose_rigid", "G_transl_offset"):
        assert torch.allclose(tr[k], torch.from_numpy(G["glbs_tr." + k]), atol=1e-6), k
    w, pts, q = (torch.from_numpy(G[k]) for k in ("rt_w", "rt_pts", "rt_q"))
    jt = oa.se3_compose(tr["J_pose_rigid"], tr["G_transl_offset"])[0]
    assert torch.allclose(oa.transform_points(jt, pts, weights=w), torch.from_numpy(G["rt_points_weighted"]), atol=1e-6)
    assert torch.allclose(oa.transform_quaternions_flip(jt, q, w), torch.from_numpy(G["rt_quats_flip"]), atol=1e-6)
    idx = torch.from_numpy(G["rt_idx"])
    assert torch.allclose(oa.transform_points(tV[0], pts, indices=idx), torch.from_numpy(G["rt_points_indexed"]), atol=1e-6)


def test_quaternion_helpers_are_consistent():
    g = torch.Generator().manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(100, 4, generator=g, dtype=torch.float64), dim=-1)
    R = oa.quaternion_to_matrix(q)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(100, 3, 3), atol=1e-12)
    q2 = oa.matrix_to_quaternion(R)
    assert torch.allclose(oa.standardize_quaternion(q2), oa.standardize_quaternion(q), atol=1e-10)
    # batch_rodrigues against the matrix exponential
    r = torch.randn(20, 3, generator=g, dtype=torch.float64)
    K = torch.zeros(20, 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -r[:, 2], r[:, 1], r[:, 2], -r[:, 0], -r[:, 1], r[:, 0]
    assert torch.allclose(oa.batch_rodrigues(r), torch.linalg.matrix_exp(K), atol=1e-7)


def test_grid_offsets_match_survey_table():
    off, pls = oa.grid_offsets()
    sizes = np.diff(off)
    assert list(sizes[:5]) == [4920, 15632, 42880, 125000, 373248]
    assert all(s == 524288 for s in sizes[5:]) and off[-1] == 6328848


def test_lr_schedule_matches_reference_get_expon_lr_func():
    """The optimizer's position schedule (dreamwaltz_g_amd.sds_step.get_expon_lr_func) against values produced by the
    reference's core/optim/optim_utils.get_expon_lr_func (golden: lr_init 1.6e-4, lr_final 1.6e-6, delay_mult 0.01, 10000 steps)."""
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd.sds_step import get_expon_lr_func
    f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=10000)
    got = np.array([f(int(s)) for s in G["lr_steps"]])
    assert np.allclose(got, G["lr_values"], rtol=1e-12, atol=0.0)
