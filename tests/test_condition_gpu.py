"""-m gpu parity of the GPU condition-image generator (dreamwaltz_g_amd/condition.py over csrc/condition.hip)
against oracle/condition.py -- whose numpy half is pinned on the reference's golden rows by tests/test_oracle_condition.py.

Bars: keypoint rows -- validity identical, x / W, y / H within 1e-9 (fp64 on both sides, fp32-representable inputs), distance within
1e-9; image from GIVEN rows -- bit-exact uint8 (integer rasterisation rules, fp32 blend with round-half-even); end to end -- bit-exact
except where an fp64 last-bit difference moves a keypoint across a pixel boundary (none observed; bar: 99.9 % of pixels)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import condition as oc

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_r2_condition.npz"))
CASES = ["front", "side", "wide", "side_ignore_body"]


def _cfg(**kw):
    from dreamwaltz_g_amd import configs
    c = configs.PromptConfig()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _inputs(name):
    """fp32-representable copies of the golden inputs (the C-ABI takes fp32 geometry and camera matrices)."""
    p = "cond.%s." % name
    f = lambda a: np.asarray(a, dtype=np.float32)          # noqa: E731
    W, H = [int(v) for v in G[p + "size"]]
    return f(G["cond.keypoints"]), f(G["cond.vertices"]), G["cond.triangles"].astype(np.int32), f(G[p + "extrinsic"]), f(G[p + "intrinsics"]), W, H


@pytest.mark.parametrize("name", CASES)
def test_keypoint_rows_match_the_oracle(name):
    from dreamwaltz_g_amd import condition as cd
    kp, v, t, E, K, W, H = _inputs(name)
    ignore = name.endswith("ignore_body")
    ref = oc.pose_keypoints(kp, v, t, E, K, W, H, ignore_body_self_occlusion=ignore)
    c = cd.SMPL2Condition(_cfg(ignore_body_self_occlusion=ignore))
    scene = cd.build_ray_casting_scene(torch.from_numpy(v).cuda()[None], t)
    rows = c.pose_rows(torch.from_numpy(kp).cuda()[None], scene, torch.from_numpy(E).cuda(), torch.from_numpy(K).cuda()).cpu().numpy()
    valid = rows[:, 3] > 0.5
    assert np.array_equal(valid, ~np.isnan(ref[:, 0]))
    assert np.abs(rows[valid, :2] - ref[valid, :2]).max() < 1e-9
    assert np.abs(rows[:, 2] - np.where(np.isnan(ref[:, 2]), rows[:, 2], ref[:, 2])).max() < 1e-9
    # and the golden rows themselves (captured from the reference with fp64 inputs): same validity, same place to 1e-5
    g = G["cond.%s.rows" % name]
    assert np.array_equal(valid, ~np.isnan(g[:, 0])) and np.abs(rows[valid, :2] - g[valid, :2]).max() < 1e-5


@pytest.mark.parametrize("name,flags", [("front", {}), ("side", dict(draw_face_landmarks=True)), ("wide", dict(draw_face_landmarks=True, openpose_left_right_flip=True)),
                                        ("front", dict(draw_body_keypoints=False, draw_face_landmarks=True)), ("side", dict(draw_hand_keypoints=False))])
def test_image_from_the_reference_rows_is_bit_exact(name, flags):
    from dreamwaltz_g_amd import condition as cd
    g = G["cond.%s.rows" % name]
    W, H = [int(v) for v in G["cond.%s.size" % name]]
    cfg = _cfg(**flags)
    ref = oc.draw_poses(g, H, W, draw_body=cfg.draw_body_keypoints, draw_hand=cfg.draw_hand_keypoints, draw_face=cfg.draw_face_landmarks,
                        flip_LR=cfg.openpose_left_right_flip)
    rows = np.concatenate([np.nan_to_num(g, nan=0.0), (~np.isnan(g[:, :1])).astype(np.float64)], axis=1)
    u8, chw = cd.SMPL2Condition(cfg).draw(torch.from_numpy(rows).cuda(), H, W, out_u8=True, out_chw=True)
    assert u8.shape == (H, W, 3) and u8.dtype == torch.uint8
    assert (ref.sum(2) > 0).mean() > 0.01
    assert np.array_equal(u8.cpu().numpy(), ref)
    assert (chw[0] - u8.permute(2, 0, 1).float() / 255.0).abs().max() < 1e-7          # the same bytes / 255 (torch multiplies by 1/255)


def test_export_pose_and_call_mirror_end_to_end():
    from dreamwaltz_g_amd import condition as cd
    kp, v, t, E, K, W, H = _inputs("side")
    ref = oc.draw_poses(oc.pose_keypoints(kp, v, t, E, K, W, H, ignore_body_self_occlusion=True), H, W, draw_face=False)
    c = cd.SMPL2Condition(_cfg())                                            # shipped defaults: no face, body never culled
    K_raw = G["cond.side.intrinsics_raw"].astype(np.float32)
    out = c(types.SimpleNamespace(vertices=torch.from_numpy(v).cuda()[None], joints=torch.from_numpy(kp).cuda()[None]), t,
            dict(extrinsic=torch.from_numpy(E).cuda()[None], intrinsics=torch.from_numpy(K_raw).cuda()[None]), "pose", H, W)
    img = out.u8.cpu().numpy()
    assert (img == ref).all(2).mean() >= 0.999
    assert out.to_pil().size == (W, H) and out.to_chw().shape == (1, 3, H, W)
    chw = c.export_pose_chw(torch.from_numpy(kp).cuda(), cd.build_ray_casting_scene(torch.from_numpy(v).cuda(), t),
                            extrinsic=torch.from_numpy(E).cuda(), intrinsics=torch.from_numpy(K).cuda(), width=W, height=H)
    assert (chw - out.to_chw()).abs().max() < 1e-7
    with pytest.raises(NotImplementedError):
        c(types.SimpleNamespace(vertices=None, joints=None), t, {}, "depth", H, W)
    with pytest.raises(RuntimeError):
        c.pose_rows(torch.from_numpy(kp), None, torch.from_numpy(E), torch.from_numpy(K))


def _ellipsoid(nu, nv, radii=(0.25, 0.8, 0.15)):
    us = np.linspace(0, 2 * np.pi, nu, endpoint=False); vs = np.linspace(0, np.pi, nv + 1)
    vv, uu = np.meshgrid(vs, us, indexing="ij")
    verts = np.stack([radii[0] * np.sin(vv) * np.cos(uu), radii[1] * np.cos(vv), radii[2] * np.sin(vv) * np.sin(uu)], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * nu + j).reshape(-1); b = (i * nu + (j + 1) % nu).reshape(-1)
    tris = np.concatenate([np.stack([a, a + nu, b], 1), np.stack([b, a + nu, b + nu], 1)])
    return verts.astype(np.float32), tris.astype(np.int32)


def test_body_sized_mesh():
    """A closed mesh of the reference's size (SMPL-X: 10 475 vertices, 20 908 triangles; here 10 658 / 21 024): rows against the
    oracle's brute-force ray cast, image against the oracle's drawing."""
    from dreamwaltz_g_amd import condition as cd
    v, t = _ellipsoid(146, 72)
    g = np.random.default_rng(0)
    kp = (v[g.integers(0, v.shape[0], 128)] * (1.0 + g.uniform(0, 0.2, (128, 1)))).astype(np.float32)
    _, _, _, E, K, W, H = _inputs("front")
    ref = oc.pose_keypoints(kp, v, t, E, K, W, H)
    c = cd.SMPL2Condition(_cfg(ignore_body_self_occlusion=False, draw_face_landmarks=True))
    scene = cd.build_ray_casting_scene(torch.from_numpy(v).cuda(), t)
    rows = c.pose_rows(torch.from_numpy(kp).cuda(), scene, torch.from_numpy(E).cuda(), torch.from_numpy(K).cuda()).cpu().numpy()
    valid = rows[:, 3] > 0.5
    assert 20 < valid.sum() < 110
    assert (valid != ~np.isnan(ref[:, 0])).sum() <= 1           # a hit within 1e-9 of a threshold may flip
    both = valid & ~np.isnan(ref[:, 0])
    assert np.abs(rows[both, :2] - ref[both, :2]).max() < 1e-9
    img = c.export_pose(torch.from_numpy(kp).cuda(), scene, extrinsic=torch.from_numpy(E).cuda(), intrinsics=torch.from_numpy(K).cuda(),
                        width=W, height=H).u8.cpu().numpy()
    assert (img == oc.draw_poses(ref, H, W)).all(2).mean() >= 0.999
