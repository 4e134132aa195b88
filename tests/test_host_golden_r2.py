"""Host-side logic of the PRODUCT (dreamwaltz-g_amd/: camera matrices, rasterizer settings, Python-side colour / covariance
options, gradient hooks, view-dependent prompt choice, optimizer groups and learning rates, condition-image preparation,
RigidTransform algebra) against golden vectors produced by the reference's own code (tests/golden/capture_golden_r2.py).
No GPU: none of this is kernel work."""
import os

import numpy as np
import torch

import dwg_import  # noqa: F401
from dreamwaltz_g_amd import avatar as av, camera, configs, guidance, optim, pgc, renderer, rigid, synth, text

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_r2.npz"))
G1 = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz"))


def T(key):
    return torch.from_numpy(np.asarray(G[key]))


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach() if torch.is_tensor(a) else torch.as_tensor(a)
    assert tuple(a.shape) == tuple(b.shape), (a.shape, b.shape)
    assert torch.allclose(a.double(), b.double(), atol=atol, rtol=rtol), float((a.double() - b.double()).abs().max())


def test_camera_matrices_match_reference():
    """data/camera/utils.py: to_extrinsic / to_projection / get_tan_half_fov."""
    E, C2W = camera.to_extrinsic(T("cam.radius"), T("cam.azimuth"), T("cam.elevation"))
    close(E, T("cam.extrinsic")); close(C2W, T("cam.c2w"))
    tanfov = camera.get_tan_half_fov(T("cam.fov"))
    close(tanfov, T("cam.tanfov"))
    close(camera.to_projection(tanfov, 0.01, 1000.0), T("cam.projection"))
    close(camera.to_projection(tanfov, 0.01, 1000.0, aspect_wh=640 / 480), T("cam.projection_wide"))
    close(camera.to_projection(tanfov, 0.01, 1000.0, tanfov_x=tanfov * 1.5), T("cam.projection_tanfov_x"))
    E2, C2 = camera.to_extrinsic(T("cam.radius"), T("cam.azimuth"), T("cam.elevation"), at_vector=((0.1, -0.2, 0.05),))
    close(E2, T("cam.extrinsic_at")); close(C2, T("cam.c2w_at"))
    cam = camera.make_camera(radius=2.0, azimuth=30.0, elevation=80.0, fovy=55.0, height=96, width=96)
    close(cam["extrinsic"][0], T("cam.extrinsic")[0]); close(cam["projection"][0], T("cam.projection")[0])


def test_rasterizer_settings_match_reference_build_gaussian_rasterizer():
    """gaussian_renderer.py:23-70, captured through a recording rasterizer at the import seam."""
    rend = renderer.GaussianRenderer(sh_levels=4, bg_color=(0.5, 0.5, 0.5))
    data = dict(extrinsic=T("cam.extrinsic")[:1], projection=T("cam.projection")[:1], c2w=T("cam.c2w")[:1], tanfov=T("cam.tanfov")[:1],
                image_height=96, image_width=96)
    rs = rend.build_gaussian_rasterizer(data).raster_settings
    close(rs.viewmatrix, T("rs.viewmatrix")); close(rs.projmatrix, T("rs.projmatrix")); close(rs.campos, T("rs.campos")); close(rs.bg, T("rs.bg"))
    sc = G["rs.scalars"]
    assert abs(rs.tanfovx - sc[0]) < 1e-7 and abs(rs.tanfovy - sc[1]) < 1e-7
    assert (rs.sh_degree, rs.scale_modifier, rs.image_height, rs.image_width) == (int(sc[2]), float(sc[3]), int(sc[4]), int(sc[5]))
    assert rs.prefiltered is False and rs.debug is False
    rsx = rend.build_gaussian_rasterizer(dict(data, tanfov_x=T("cam.tanfov")[:1] * 1.25, image_width=120)).raster_settings
    assert abs(rsx.tanfovx - G["rs.scalars_x"][0]) < 1e-7 and abs(rsx.tanfovy - G["rs.scalars_x"][1]) < 1e-7


def test_python_side_colour_and_covariance_options_match_reference():
    """gaussian_renderer.py:72-128 (compute_colors / compute_3d_covariance), gaussian_utils.get_colors, eval_sh."""
    close(renderer.GaussianRenderer.compute_3d_covariance(T("cov3d.scales"), T("cov3d.quats")), T("cov3d.out"), atol=1e-8)
    sh, dirs = torch.from_numpy(G1["sh_in"]), torch.from_numpy(G1["sh_dirs"])
    r = renderer.GaussianRenderer()
    for lv in (1, 2, 3, 4):
        close(renderer.get_colors(sh, dirs, lv), torch.from_numpy(G1["sh_colors_l%d" % lv]), atol=2e-6)
        close(r.compute_colors(sh, directions=dirs, sh_levels=lv), torch.from_numpy(G1["sh_colors_l%d" % lv]), atol=2e-6)


def test_gradient_hooks_match_reference_pgc():
    """core/guidance/pgc.py:15-43."""
    g = T("pgc.grad")
    close(pgc.build_grad_hook_func(True, False, 1.5)(g.clone()), T("pgc.clip"))
    close(pgc.build_grad_hook_func(True, False, 0.7, mask=T("pgc.mask"))(g.clone()), T("pgc.clip_mask"))
    close(pgc.build_grad_hook_func(False, True, 1.0)(g.clone()), T("pgc.norm"))
    close(pgc.build_grad_hook_func(True, True, 2.0)(g.clone()), T("pgc.clip_norm"))


def test_view_dependent_prompt_selection_matches_reference():
    """core/guidance/text.py:36-154 over a sweep of azimuths / elevations, and the prompt list itself."""
    cfg = configs.PromptConfig()
    assert [cfg.angle_front, cfg.angle_overhead] == list(G["text.cfg"])
    ta = text.TextAugmentation("a person", cfg)
    assert list(ta.azimuth_range) + list(ta.elevation_range) == list(G["text.ranges"])
    assert ta.texts == [str(s) for s in G["text.texts"]]
    for i, e in enumerate(G["text.elevations"]):
        for j, a in enumerate(G["text.azimuths"]):
            assert int(ta(torch.tensor([a]), torch.tensor([e]))[0]) == int(G["text.index"][i, j]), (a, e)


def _tiny_avatar(learn_hand_betas=False):
    body = synth.synthetic_body(V=64, J=55, seed=0)
    glbs = av.GeneralLinearBlendSkinning(body)
    n = 8
    g = torch.Generator().manual_seed(0)
    vi = torch.arange(12); tri = torch.tensor([[0, 1, 2], [3, 4, 5]])
    mesh = {"hands": av.MeshBindingGaussianModel(body["v_template"][vi], tri, vi)}
    return av.DreamWaltzG(glbs, torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g) + 0.1, torch.randn(n, 4, generator=g),
                          torch.rand(n, 55, generator=g), {}, mesh, learn_hand_betas=learn_hand_betas)


def test_get_optimizer_groups_and_learning_rates_match_reference():
    """avatar.get_optimizer(cfg) (avatar.py:1590-1635): optimizer names, group names, Adam hyper-parameters and the learning rates
    GaussianOptimizer.update_learning_rate produces (gaussian_optimizer.py:130-141), incl. the mesh groups' rates (avatar.py:1085-1090)."""
    cfg = configs.TrainConfig()
    c = G["opt.cfg"]
    assert [cfg.render.position_lr_init, cfg.render.position_lr_final, cfg.render.scaling_lr, cfg.render.rotation_lr, cfg.nerf.lr,
            cfg.render.betas_lr, cfg.render.lbs_lr] == list(c)
    cfg.optim.iters = 10000
    a = _tiny_avatar(learn_hand_betas=True)
    opts = a.get_optimizer(cfg)
    assert list(opts.keys()) == ['avatar', 'lbs', 'nerf', 'mesh_hands']
    go = opts['avatar']
    assert [pg['name'] for pg in go.param_groups] == [str(s) for s in G["opt.group_names"]]
    spatial = float(G["opt.spatial_scale"][0])
    for i, it in enumerate(G["opt.iterations"]):
        go.update_learning_rate(spatial, int(it))
        got = [pg['lr'] for pg in go.param_groups]
        assert np.allclose(got, G["opt.lrs"][i], rtol=1e-12, atol=0), (it, got, G["opt.lrs"][i])
    eps, b1, b2 = G["opt.adam"]
    assert all(pg['eps'] == eps and tuple(pg['betas']) == (b1, b2) for pg in go.param_groups)
    nerf = opts['nerf'].param_groups
    assert [pg['lr'] for pg in nerf] == [c[4] * 10, c[4], c[4]] and all(tuple(pg['betas']) == (0.9, 0.99) and pg['eps'] == 1e-15 for pg in nerf)
    assert [pg['lr'] for pg in opts['lbs'].param_groups] == [c[5]] and opts['lbs'].param_groups[0]['eps'] == 1e-8
    mesh = opts['mesh_hands'].param_groups
    assert [(pg['name'], pg['lr']) for pg in mesh] == [('bary_coords', c[0]), ('scales', c[2])] and all(pg['eps'] == 1e-15 for pg in mesh)
    assert not hasattr(opts['nerf'], 'update_learning_rate') and not hasattr(opts['mesh_hands'], 'update_learning_rate')
    # every parameter lives in the one flat buffer, gradients alias the flat gradient buffer
    buf = opts.buffers
    assert a._positions.data_ptr() == buf.flat.data_ptr() and a._positions.grad.data_ptr() == buf.grad.data_ptr()
    assert buf.total % 4 == 0 and opts['mesh_hands'].end == buf.total


def test_prepare_image_matches_reference_controlnet():
    """controlnet.py:33-55: PIL -> LANCZOS resize -> float/255 -> NCHW -> repeat to the CFG batch."""
    from PIL import Image
    pi = guidance.ControlNetScoreDistillation.prepare_image
    out = pi(None, [Image.fromarray(G["img.in"])], width=64, height=64, batch_size=2, num_images_per_prompt=1, device="cpu", dtype=torch.float32)
    close(out, T("img.out"), atol=0)
    out = pi(None, Image.fromarray(G["img.in_same"]), width=64, height=64, batch_size=2, num_images_per_prompt=1, device="cpu", dtype=torch.float32)
    close(out, T("img.out_same"), atol=0)
    t = T("img.out")[:1]
    close(pi(None, [t], 64, 64, 2, 1, "cpu", torch.float32), T("img.out"), atol=0)


def test_rigid_transform_class_matches_reference():
    """rigid.RigidTransform (the small-algebra paths, which run wherever their inputs live) vs inverse_lbs.py:15-260."""
    RT = rigid.RigidTransform
    src = T("rt.A").clone()
    rt = RT(SE3=src)
    close(rt.inverse().SE3, T("rt.inverse")); close(src, T("rt.source_after_inverse"))          # source mutated in place (Q7)
    src = T("rt.source_after_inverse")
    w = T("rt.w")
    close(RT(SE3=src.clone()).weight(w).SE3, T("rt.weighted"))
    close(RT(SE3=src.clone()).compose(RT(SE3=T("rt.B")), RT(SE3=src.clone())).SE3, T("rt.compose"))
    wt = RT(SE3=src.clone()).weight(w)
    close(RT._inverse_transform_points(T("rt.pts"), R=wt.R, T=wt.T), T("rt.inv_points"), atol=1e-5)
    close(RT(SE3=src.clone()).transform_quaternions(T("rt.q"), weights=w, rotation_mode='matrix'), T("sd.rt.quat_mode_matrix"))
    close(RT(SE3=src.clone()).transform_quaternions(T("rt.q"), weights=w, rotation_mode='quaternion'), T("sd.rt.quat_mode_quaternion"))
    close(RT(SE3=src.clone()).transform_quaternions(T("rt.q"), indices=T("rt.idx"), rotation_mode='matrix'), T("sd.rt.quat_indexed"))
    # constructor forms and squeeze (mutates, returns self)
    t = RT(T=torch.tensor([[1.0, 2.0, 3.0]]))
    assert t.SE3.shape == (1, 4, 4) and t.squeeze(0) is t and t.SE3.shape == (4, 4) and t.R.shape == (3, 3) and t.T.shape == (3,)
    # round-1 golden through the class: points by index / by weight, flip path
    pts, q, idx = (torch.from_numpy(G1[k]) for k in ("rt_pts", "rt_q", "rt_idx"))
    tV = RT(SE3=torch.from_numpy(G1["glbs_tV"])).squeeze(0)
    close(tV.transform_points(pts, indices=idx), torch.from_numpy(G1["rt_points_indexed"]))
    jt = RT.compose(RT(SE3=torch.from_numpy(G1["glbs_tr.J_pose_rigid"])), RT(SE3=torch.from_numpy(G1["glbs_tr.G_transl_offset"]))).squeeze(0)
    w1 = torch.from_numpy(G1["rt_w"])
    close(jt.transform_points(pts, weights=w1), torch.from_numpy(G1["rt_points_weighted"]))
    close(jt.transform_quaternions(q, weights=w1, flip_rotation_axis=True), torch.from_numpy(G1["rt_quats_flip"]))


def test_position_lr_schedule_is_the_reference_function():
    f = optim.get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=10000)
    assert np.allclose([f(int(s)) for s in G1["lr_steps"]], G1["lr_values"], rtol=1e-12, atol=0.0)
