"""Pins the CPU oracle against the round-2 golden vectors: outputs of the reference's OWN in-repo code run in the build
container (tests/golden/capture_golden_r2.py -> reference_golden_r2.npz).  Keys under "sd." went through the arithmetic
stand-ins for smplx.lbs / pytorch3d.transforms (they pin the reference's algebra and orchestration, not those libraries)."""
import os

import numpy as np
import torch

from oracle import animate as oa
from oracle import sd15 as osd

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_r2.npz"))


def T(key):
    return torch.from_numpy(np.asarray(G[key]))


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach() if torch.is_tensor(a) else torch.as_tensor(a)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a.double(), b.double(), atol=atol, rtol=rtol), float((a.double() - b.double()).abs().max())


def test_compute_normal_matches_reference():
    """utils/mesh.py:34-94."""
    vn, fn = oa.compute_normal(T("normal.verts"), T("normal.tri"))
    close(vn, T("normal.vn")); close(fn, T("normal.fn"))


def test_mesh_binding_matches_reference():
    """MeshBindingGaussianModel.get_positions / get_scales_and_quaternions (avatar.py:1016-1079): raw bary for normals (Q5),
    /6 tangent scales, clamp(0.5, 2), rows 1,2 negated (Q3), s0 == 0."""
    tri = T("normal.tri")
    pos = oa.mesh_positions(T("mesh.bary"), T("mesh.vobs"), tri)
    close(pos, T("mesh.positions"))
    sc, q = oa.mesh_scales_and_quaternions(T("mesh.bary"), T("mesh.scales_raw"), T("mesh.vobs"), tri, pos, 6)
    close(sc, T("mesh.scales")); close(q, T("mesh.quaternions"), atol=2e-6)
    assert float(sc[:, 0].abs().max()) == 0.0


def test_mlp_matches_reference():
    """nerf_model.py:12-33."""
    w = [T("mlp.sd.net.%d.weight" % l) for l in range(3)]; b = [T("mlp.sd.net.%d.bias" % l) for l in range(3)]
    close(oa.mlp_forward(T("mlp.x"), w, b), T("mlp.y"))


def test_activations_match_reference():
    """gaussian_model.py:25-56."""
    close(torch.exp(T("act.scales_raw")), T("act.scales"))
    close(torch.exp(T("act.scales_raw").mean(-1, keepdim=True).expand(-1, 3)), T("act.scales_mean"))
    close(torch.nn.functional.normalize(T("act.quats_raw"), dim=-1), T("act.quats"))
    close(torch.sigmoid(T("act.opac_raw")), T("act.opac"))
    p = T("act.opac")
    close(torch.log(p / (1 - p)), T("act.inv_sigmoid"))


def test_non_rigid_transform_matches_reference_in_every_branch():
    """avatar.py:1464-1498 (scale branch keyed by non_rigid_ROTATION_mode: checklist Q4)."""
    base = {k: T("nrt." + k) for k in ("_scales", "_quaternions", "positions", "offsets", "mlp_scales", "mlp_quats")}
    variants = {"default": {}, "mul": dict(non_rigid_rotation_mode='mul'), "no_learn_scale": dict(learn_scale=False),
                "no_nr_scales": dict(use_non_rigid_scales=False), "no_offsets": dict(use_non_rigid_offsets=False)}
    for name, fl in variants.items():
        p, s, q = oa.non_rigid_transform(base["positions"], base["offsets"], base["mlp_scales"], base["mlp_quats"], base["_scales"],
                                         base["_quaternions"], **fl)
        close(p, T("nrt.%s.positions" % name)); close(s, T("nrt.%s.scales" % name)); close(q, T("nrt.%s.quaternions" % name))
    for name, fl in {"rot_add": dict(use_non_rigid_rotations=True), "rot_mul": dict(use_non_rigid_rotations=True, non_rigid_rotation_mode='mul'),
                     "rot_nolearn": dict(use_non_rigid_rotations=True, learn_quaternions=False)}.items():
        _, _, q = oa.non_rigid_transform(base["positions"], base["offsets"], base["mlp_scales"], base["mlp_quats"], base["_scales"],
                                         base["_quaternions"], **fl)
        close(q, T("sd.nrt.%s.quaternions" % name))


def test_rigid_transform_algebra_matches_reference():
    """inverse_lbs.py:102-188: inverse (mutates its input, Q7), weight, compose order, general-inverse point transform (Q8)."""
    A = T("rt.A").clone()
    inv = oa.se3_inverse(A)
    close(inv, T("rt.inverse")); close(A, T("rt.source_after_inverse"))
    src = T("rt.source_after_inverse")
    w = T("rt.w")
    close(oa.se3_weight(src, w), T("rt.weighted"))
    close(oa.se3_compose(src, T("rt.B"), src), T("rt.compose"))
    wt = oa.se3_weight(src, w)
    close(oa.inverse_transform_points(T("rt.pts"), wt[..., :3, :3], wt[..., :3, 3]), T("rt.inv_points"), atol=1e-5)
    close(oa.transform_quaternions(src, T("rt.q"), weights=w, rotation_mode='matrix'), T("sd.rt.quat_mode_matrix"))
    close(oa.transform_quaternions(src, T("rt.q"), weights=w, rotation_mode='quaternion'), T("sd.rt.quat_mode_quaternion"))
    close(oa.transform_quaternions(src, T("rt.q"), indices=T("rt.idx"), rotation_mode='matrix'), T("sd.rt.quat_indexed"))
    # the two golden vectors round 1 captured but never read
    G1 = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz"))
    body = oa.SyntheticBody(V=int(G1["body_V"][0]), F_=int(G1["body_F"][0]), seed=int(G1["body_seed"][0]))
    inp = {k[len("glbs_in."):]: torch.from_numpy(G1[k]) for k in G1.files if k.startswith("glbs_in.")}
    _, _, tr = oa.glbs_forward(body, **inp)
    close(oa.se3_inverse(tr["J_pose_rigid"].clone()), torch.from_numpy(G1["rt_inverse"]), atol=2e-6)


def test_exp_se3_matches_reference():
    """core/deformation/rigid_utils.py:60-83 (captured in round 1 as se3_out; DeformNetwork's is_6dof branch, off by default)."""
    G1 = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz"))
    S, th = torch.from_numpy(G1["se3_S"]), torch.from_numpy(G1["se3_theta"])
    w, v = S[:, :3], S[:, 3:]
    K = torch.zeros(S.shape[0], 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    t = th[:, :, None]
    R = torch.eye(3) + torch.sin(t) * K + (1 - torch.cos(t)) * (K @ K)
    p = (t * torch.eye(3) + (1 - torch.cos(t)) * K + (t - torch.sin(t)) * (K @ K)) @ v[:, :, None]
    out = torch.eye(4).repeat(S.shape[0], 1, 1)
    out[:, :3, :3] = R; out[:, :3, 3:] = p
    close(out, torch.from_numpy(G1["se3_out"]), atol=1e-5)


def _animate_inputs():
    V, F_, seed = (int(x) for x in G["sd.animate.body"])
    body = oa.SyntheticBody(V=V, F_=F_, seed=seed)
    nets = oa.init_avatar_networks(seed=int(G["sd.animate.nets_seed"][0]), table_std=float(G["sd.animate.table_std"][0]))
    nets["deform"] = {k[len("sd.animate.deform."):]: T(k) for k in G.files if k.startswith("sd.animate.deform.")}
    params = {k: T("sd.animate.param." + k) for k in ("_positions", "_scales", "_quaternions", "_lbs_weights")}
    mesh = dict(vertex_indices=T("sd.animate.mesh.vertex_indices"), triangles=T("sd.animate.mesh.triangles"),
                vertex_coords=T("sd.animate.mesh.vertex_coords"), bary=T("sd.animate.mesh.bary"), scales=T("sd.animate.mesh.scales"))
    cnl = {k[len("sd.animate.cnl."):]: T(k) for k in G.files if k.startswith("sd.animate.cnl.")}
    obs = {k[len("sd.animate.obs."):]: T(k) for k in G.files if k.startswith("sd.animate.obs.")}
    return body, nets, params, mesh, cnl, obs


def _check_animate(tag, extra_betas):
    body, nets, params, mesh, cnl, obs = _animate_inputs()
    leaves = dict(_positions=params["_positions"].clone().requires_grad_(True), _scales=params["_scales"].clone().requires_grad_(True),
                  _quaternions=params["_quaternions"].clone().requires_grad_(True), _lbs_weights=params["_lbs_weights"])
    nets["table"] = nets["table"].clone().requires_grad_(True)
    mesh["bary"] = mesh["bary"].clone().requires_grad_(True); mesh["scales"] = mesh["scales"].clone().requires_grad_(True)
    eb = None if extra_betas is None else extra_betas.clone().requires_grad_(True)
    out = oa.animate(leaves, nets, body, obs, cnl, mesh=mesh, extra_betas=eb)
    pre = "sd.%s." % tag
    for f in ("positions", "opacities", "colors", "quaternions", "scales"):
        close(out[f], T(pre + "out." + f), atol=3e-6)
    loss = sum((out[f] * T(pre + "lossw." + f)).sum() for f in ("positions", "opacities", "colors", "quaternions", "scales"))
    loss.backward()

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    assert rel(leaves["_positions"].grad, T(pre + "grad._positions")) < 1e-4
    assert rel(leaves["_scales"].grad, T(pre + "grad._scales")) < 1e-5
    assert rel(leaves["_quaternions"].grad, T(pre + "grad._quaternions")) < 1e-4
    assert rel(nets["table"].grad, T(pre + "grad.table")) < 1e-4
    assert rel(mesh["bary"].grad, T(pre + "grad.bary")) < 1e-4
    assert rel(mesh["scales"].grad, T(pre + "grad.mesh_scales")) < 1e-5
    if eb is not None:
        assert rel(eb.grad, T(pre + "grad._betas")) < 1e-4


def test_animate_matches_the_reference_animate():
    """DreamWaltzG.animate as written (avatar.py:1500-1588), free + mesh-bound Gaussians, outputs and every parameter gradient."""
    _check_animate("animate", None)


def test_animate_with_learned_hand_betas_matches_the_reference():
    """Sub-stage 2.1 of the shipped recipe (train_w_expr.sh:66): mesh-bound vertices follow lbs_model.forward(extra_betas=_betas)
    (avatar.py:1551-1565); includes d loss / d _betas."""
    _check_animate("animate_betas", T("sd.animate.param._betas"))


def test_inverse_lbs_transform_matches_reference():
    body, nets, params, mesh, cnl, obs = _animate_inputs()
    _, _, ctr = oa.glbs_forward(body, **cnl)
    w = oa.lbs_weight_activation(params["_lbs_weights"])
    close(oa.inverse_lbs_transform(T("sd.invlbs.in"), ctr, w), T("sd.invlbs.out"), atol=1e-5)


def _sds_cfg():
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import sd15 as nsd
    c = [int(x) for x in G["sd.sds.cfg"]]
    ucfg = nsd.UNetConfig(block_out_channels=(c[0], c[1]), layers_per_block=c[2], heads=c[3], cross_dim=c[4], groups=c[5],
                          attn_blocks=(True, False), cond_channels=(c[6], c[7]))
    vcfg = nsd.VAEConfig(block_out_channels=(c[8], c[9]), layers_per_block=c[10], groups=c[11])
    usd = nsd.random_state_dict(nsd.unet_param_shapes(ucfg), seed=0)
    csd = nsd.random_state_dict(nsd.controlnet_param_shapes(ucfg), seed=1)
    vsd = nsd.random_state_dict(nsd.vae_encoder_param_shapes(vcfg), seed=2)
    return ucfg, vcfg, usd, csd, vsd


def test_sds_call_matches_reference_calc_gradients_and_call():
    """BasicScoreDistillation.calc_gradients / __call__ (basic.py:546-663, 778-917) run on the oracle's reduced-width networks:
    pins ('neg','text') ordering, CFG extrapolation at guidance 50, 'sjc' weight, targets = sources - gradients, SpecifyGradient."""
    ucfg, vcfg, usd, csd, vsd = _sds_cfg()
    t = T("sd.sds.timestep")
    text = torch.cat([T("sd.sds.text.neg"), T("sd.sds.text.text")], 0)
    close(text, T("sd.sds.text_embeddings"))
    with torch.no_grad():
        pred = osd.predict_noise(ucfg, usd, csd, torch.cat([T("sd.sds.latents_noisy")] * 2), t, text, T("sd.sds.cond"))
        u, c = pred.chunk(2)
        npred = u + float(G["sd.sds.guidance"][0]) * (c - u)
    close(npred, T("sd.sds.noise_pred"), atol=1e-4, rtol=1e-4)
    close(npred - T("sd.sds.noise"), T("sd.sds.gradients"), atol=1e-4, rtol=1e-4)
    # whole call through oracle.sd15.sds_step
    img = T("sd.sds.call.image").clone().requires_grad_(True)
    grads, g_img = osd.sds_step(ucfg, vcfg, usd, csd, vsd, img, text, T("sd.sds.cond"), t, T("sd.sds.call.noise"), T("sd.sds.call.vae_noise"),
                                guidance_scale=float(G["sd.sds.guidance"][0]))
    lat = osd.sample_latents(osd.vae_encode_moments(vcfg, vsd, img), T("sd.sds.call.vae_noise"), vcfg.scaling_factor)
    close(lat, T("sd.sds.call.latents"), atol=1e-5)
    close(grads, T("sd.sds.call.gradients"), atol=2e-4, rtol=1e-4)
    close(g_img, T("sd.sds.call.image_grad"), atol=1e-4, rtol=1e-3)
    assert list(G["sd.sds.call.keys"]) == ['diffusion_loss', 'gradients', 'latents', 'sources', 'targets', 'timestep']
    close(T("sd.sds.call.sources") - T("sd.sds.call.gradients"), T("sd.sds.call.targets"), atol=1e-6)
    assert float(G["sd.sds.call.diffusion_loss"][0]) == 1.0
    assert list(G["sd.sds.types"]) == ['sds', 'sjc', 'True', 'uniform']
