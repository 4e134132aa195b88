"""-m gpu: the ASSEMBLED step -- Scene.forward (animate -> render) -> guidance -> backward -> optimizers, as SDSTrainer.train_step
runs it (mirror of the loop body of /root/reference/core/trainer.py:859-890) -- against the oracle chain
oracle.animate -> raster oracle (C) -> oracle.sd15.sds_step, and the sync-free pair sizing against the exact one."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import animate as oa
from oracle import sd15 as osd
from tests import raster_cases as rc
from tests.test_animate_gpu import _avatar_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, r):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1); r = torch.as_tensor(r).detach().double().cpu().reshape(-1)
    return float((a - r).norm() / r.norm().clamp_min(1e-30))


def _cos(a, r):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1); r = torch.as_tensor(r).detach().double().cpu().reshape(-1)
    return float((a @ r) / (a.norm() * r.norm()).clamp_min(1e-30))


def _note(name, **kw):
    path = os.path.join(ROOT, "gpurun_out", "parity_sds_step.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[name] = kw
    json.dump(d, open(path, "w"), indent=1)
    print("[parity]", name, kw)


def _oracle_chain(params, nets, body, obs, cnl, mesh, cam, res, image_grad_fn):
    """float32 oracle animate (autograd) -> C raster oracle forward -> d loss / d image from `image_grad_fn(image[3,H,W])` -> C raster
    oracle backward (float64) -> autograd back to the parameters.  Returns (image, grads by name, reference pair count)."""
    from dreamwaltz_g_amd import camera
    leaves = dict(_positions=params["_positions"].clone().requires_grad_(True), _scales=params["_scales"].clone().requires_grad_(True),
                  _quaternions=params["_quaternions"].clone().requires_grad_(True), _lbs_weights=params["_lbs_weights"])
    n = dict(nets); n["table"] = nets["table"].clone().requires_grad_(True)
    m = None
    if mesh is not None:
        m = dict(mesh); m["bary"] = mesh["bary"].clone().requires_grad_(True); m["scales"] = mesh["scales"].clone().requires_grad_(True)
    out = oa.animate(leaves, n, body, obs, cnl, mesh=m)
    view, proj, campos, tfx, tfy = camera.raster_matrices({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in cam.items()})
    sc = dict(means3D=out["positions"].detach(), opacities=out["opacities"].detach(), colors=out["colors"].detach(), scales=out["scales"].detach(),
              rotations=out["quaternions"].detach(), viewmatrix=view, projmatrix=proj, campos=campos, tanfovx=tfx, tanfovy=tfy,
              bg=torch.tensor([0.5, 0.5, 0.5]), H=res, W=res)
    fwd = rc.oracle_forward(sc)
    g_img = image_grad_fn(torch.from_numpy(fwd["color"]))
    gb = rc.oracle_backward(sc, g_img.numpy().astype(np.float32), None, None, dtype=np.float64)
    torch.autograd.backward([out["positions"], out["opacities"], out["colors"], out["scales"], out["quaternions"]],
                            [torch.from_numpy(gb["means3D"]).float(), torch.from_numpy(gb["opacities"]).float().reshape(-1, 1),
                             torch.from_numpy(gb["colors"]).float(), torch.from_numpy(gb["scales"]).float(),
                             torch.from_numpy(gb["rotations"]).float()])
    grads = dict(_positions=leaves["_positions"].grad, _scales=leaves["_scales"].grad, _quaternions=leaves["_quaternions"].grad,
                 table=n["table"].grad)
    if m is not None:
        grads.update(bary=m["bary"].grad, mesh_scales=m["scales"].grad)
    return fwd, grads, int(fwd["num_pairs"])


@pytest.mark.parametrize("async_pairs", [False, True])
def test_step_without_guidance_matches_oracle_chain_and_first_adam_update(async_pairs):
    """3 000 free + ~1 100 mesh-bound Gaussians, 128x128, loss = sum(image * W): image, every parameter gradient, the pair count and
    the first Adam update (|delta| = lr of the parameter's group, sign = -sign(grad)) against the oracle chain."""
    from dreamwaltz_g_amd import sds_step, synth
    res = 128
    a, params, nets, body, _, cnl, mesh = _avatar_pair(with_mesh=True)
    step = sds_step.SDSStep(res=res, guidance=False, avatar=a, async_pair_count=async_pairs, iters=10000)
    obs = synth.random_smpl_inputs(seed=0)
    wimg = step.trainer.diffusion.wimg[0].permute(2, 0, 1).cpu()          # [3,H,W]
    fwd, gref, Kref = _oracle_chain(params, nets, body, obs, cnl, mesh, step.data, res, lambda img: wimg)
    before = {k: getattr(a, k).detach().clone() for k in ("_positions", "_scales", "_quaternions")}
    loss, render_outputs, _, _ = step.run()
    img = render_outputs["image"][0].permute(2, 0, 1).detach().cpu()
    err = (img - torch.from_numpy(fwd["color"])).abs()
    assert float(torch.quantile(err.reshape(-1), 0.999)) <= 1e-4, float(err.max())
    assert step.num_pairs[1] == Kref, (step.num_pairs, Kref)
    assert set(render_outputs.keys()) >= {"image", "depth", "alpha", "image_fg", "regularizations"}
    got = dict(_positions=a._positions.grad, _scales=a._scales.grad, _quaternions=a._quaternions.grad, table=a.nerf_encoder.embeddings.grad,
               bary=a.mesh_binding_gaussians["hands"]._bary_coords.grad, mesh_scales=a.mesh_binding_gaussians["hands"]._scales.grad)
    rep = {}
    for k, g in got.items():
        rep[k] = _rel(g, gref[k])
        assert rep[k] < 5e-3, (k, rep[k])
    _note("step_no_guidance_async%d" % int(async_pairs), **rep)
    # first Adam step: m_hat = g, v_hat = g^2  ->  delta = -lr g / (|g| + eps)
    spatial = float(step.data["radius"][0]) * float(step.data["tanfov"][0])
    lrs = dict(_positions=sds_step.get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=20000)(1) * spatial,
               _scales=2.5e-3 * spatial, _quaternions=1e-3)
    for k, lr in lrs.items():
        g = gref[k].reshape(-1)
        sel = g.abs() > 1e-3 * g.abs().max()
        delta = (getattr(a, k).detach().cpu() - before[k].cpu()).reshape(-1)
        expect = -lr * torch.sign(g)
        bad = ((delta[sel] - expect[sel]).abs() > 1e-3 * lr).float().mean()
        assert float(bad) < 1e-3, (k, float(bad))
    assert step.trainer.redone_frames == 0


# Bars of the end-to-end "image -> guidance -> raster backward -> parameters" chain per plan precision: (every parameter gradient rel-L2,
# cosine).  f32x / f32 are the reference's own precision for the guidance stage (configs/__init__.py:236,241): what is left is the avatar
# side's own agreement with the oracle chain (float atomics, <= 2e-3 without diffusion).  f16 / bf16: the stated tolerance of those plans for
# PARAMETER gradients at this reduced width under CFG 50 -- about 1.5x the values measured on an MI355X (profiles/r04_parity_sds_step.json,
# DESIGN.md section 2).
# measured: f32x / f32 2.6e-4 .. 2.0e-3; f16 1.5 .. 4.2 %; bf16 15 .. 28 %.  Round 5: the step is bit-reproducible now, so the f32x / f32 numbers no
# longer move from run to run -- but the worst parameter (the 798 mesh-bound scales, a near-cancelling sum: cosine 0.999998) answers a 2e-6
# relative change of the IMAGE gradient (the VAE backward with / without its power-of-two pre-scale: both 2.1e-6 off the fp32 oracle) with
# 1.43e-3 -> 2.01e-3, and the exact-f32 plans sit at 1.72e-3: the chain's own conditioning, not a precision of the plans.  Bar 3e-3.
_STEP_BARS = {"f32x": (3e-3, 0.99999), "f32": (3e-3, 0.99999), "f16": (6e-2, 0.998), "bf16": (0.42, 0.95)}


@pytest.mark.parametrize("dtype", ["f32x", "f32", "f16", "bf16"])
def test_step_with_reduced_width_guidance_matches_oracle_chain(dtype):
    """The same chain with the SDS gradient in the middle (reduced-width UNet / ControlNet / VAE, 128x128, forced timestep and
    noises): EVERY parameter gradient (free Gaussians' positions / scales / quaternions, grid table, mesh-bound barycentrics / scales)
    against oracle.animate -> raster oracle -> oracle.sd15.sds_step (all fp32 / fp64 on the CPU), per plan precision."""
    from dreamwaltz_g_amd import guidance, sd15, sds_step, synth
    res = 128
    dev = torch.device("cuda")
    ucfg = sd15.UNetConfig(block_out_channels=(64, 128, 128, 128), cross_dim=64, cond_channels=(16, 32, 32, 64))
    vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
    usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=1)
    csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=2)
    vsd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
    gd = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, usd, csd, vsd, image_hw=res, dtype=dtype)
    a, params, nets, body, _, cnl, mesh = _avatar_pair(with_mesh=True)
    step = sds_step.SDSStep(res=res, guidance=True, avatar=a, guidance_obj=gd, async_pair_count=False, gpu_condition=False)
    obs = synth.random_smpl_inputs(seed=0)
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(1, 4, res // 8, res // 8, generator=g); vnoise = torch.randn(1, 4, res // 8, res // 8, generator=g)
    t = torch.tensor([500])
    az, el = step.data["azimuth"], step.data["elevation"]
    vi = int(step.trainer.view_prompt(az, el)[0])
    text = torch.cat([step.text["neg"], step.text["viewed"][vi]], 0).cpu()
    cond = step.data["cond_images"].cpu()

    def image_grad(img):
        imgr = img[None].clone().requires_grad_(True)
        _, g_img = osd.sds_step(ucfg, vcfg, usd, csd, vsd, imgr, text, cond, 500, noise, vnoise)
        return g_img[0]
    fwd, gref, _ = _oracle_chain(params, nets, body, obs, cnl, mesh, step.data, res, image_grad)
    loss, render_outputs, sd_outputs, _ = step.run(timestep=t.to(dev), noise=noise.to(dev), posterior_noise=vnoise.to(dev))
    assert float(loss) == 1.0 and sd_outputs["gradients"].shape == (1, 4, res // 8, res // 8)
    assert sorted(sd_outputs.keys()) == ['diffusion_loss', 'gradients', 'latents', 'sources', 'targets', 'timestep']
    got = {"_positions": a._positions.grad, "_scales": a._scales.grad, "_quaternions": a._quaternions.grad, "table": a.nerf_encoder.embeddings.grad,
           "bary": a.mesh_binding_gaussians["hands"]._bary_coords.grad, "mesh_scales": a.mesh_binding_gaussians["hands"]._scales.grad}
    rep = {}
    for k, gv in got.items():
        rep[k + "_rel"], rep[k + "_cos"] = _rel(gv, gref[k]), _cos(gv, gref[k])
    _note("step_reduced_width_guidance_" + dtype, **rep)
    bar, cmin = _STEP_BARS[dtype]
    for k in got:
        assert rep[k + "_rel"] < bar and rep[k + "_cos"] > cmin, (dtype, k, rep)


def test_step_draws_its_condition_image_on_the_gpu():
    """SDSStep(gpu_condition=True): the per-step OpenPose image of the posed body (full-vertex skeleton pass, 128 keypoints, culling
    against 20 908 triangles, drawing) equals the oracle's export_pose of the same posed geometry, changes with the pose, and is
    what the guidance consumes."""
    from dreamwaltz_g_amd import guidance, sd15, sds_step, synth
    from oracle import condition as oc
    res = 128
    dev = torch.device("cuda")
    ucfg = sd15.UNetConfig(block_out_channels=(64, 128, 128, 128), cross_dim=64, cond_channels=(16, 32, 32, 64))
    vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
    gd = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, image_hw=res)
    step = sds_step.SDSStep(n_gaussians=3000, res=res, guidance=True, guidance_obj=gd, async_pair_count=False)
    assert step.condition is not None and step.condition["triangles"].shape == (20908, 3)
    pose = synth.random_smpl_inputs(seed=5, device=dev)
    img = step.condition_image(pose)
    assert img.shape == (1, 3, res, res) and float(img.max()) <= 1.0 and float((img.sum(1) > 0).float().mean()) > 0.005
    # the same posed geometry through the oracle
    lbs = step.avatar.lbs_model
    with torch.no_grad():
        _, _, trf = lbs(**pose)
        verts = lbs.transform_vertices(trf, step.condition["all_vertices"], lbs.v_template)
        J = lbs._joints(trf)
        joints = (trf.A[:, :3, :3] * J[:, None, :]).sum(-1) + trf.A[:, :3, 3]
        kp = torch.cat([joints, verts[step.condition["pick"]]], 0)
    cfgp = step.cfg.prompt
    rows = oc.pose_keypoints(kp.cpu().numpy(), verts.cpu().numpy(), step.condition["triangles"].cpu().numpy(), step.data["extrinsic"][0].cpu().numpy(),
                             step.condition["intrinsics"].cpu().numpy(), res, res, ignore_body_self_occlusion=cfgp.ignore_body_self_occlusion)
    ref = oc.draw_poses(rows, res, res, draw_face=cfgp.draw_face_landmarks)
    got = (img[0].permute(1, 2, 0) * 255.0).round().to(torch.uint8).cpu().numpy()
    assert (got == ref).all(2).mean() >= 0.999
    step.run(); c0 = step.data["cond_images"].clone()
    step.run(); c1 = step.data["cond_images"]
    assert c0.shape == (1, 3, res, res) and not torch.equal(c0, c1)
    assert torch.isfinite(step.optimizers.buffers.flat).all()


def test_c3_size_sync_vs_async_pair_sizing_and_overflow_recovery():
    """100 000 Gaussians, 512x512, animate + raster + Adam (no diffusion), three steps: the sync-free pair sizing renders exactly the
    image of the exact (read-back) sizing, reports the oracle's pair count, and a frame whose pair workspace is too small is
    rendered again -- in the same step, before any optimizer runs."""
    from dreamwaltz_g_amd import sds_step, synth
    res, G = 512, 100000
    s_sync = sds_step.SDSStep(n_gaussians=G, res=res, guidance=False, async_pair_count=False)
    s_async = sds_step.SDSStep(n_gaussians=G, res=res, guidance=False, async_pair_count=True)
    o1 = s_sync.run(); o2 = s_async.run()
    assert torch.equal(o1[1]["image"], o2[1]["image"]) and torch.equal(o1[1]["depth"], o2[1]["depth"])
    assert s_sync.num_pairs == s_async.num_pairs      # (pairs on 8x8 blocks after exact culling, the reference's 16x16 tile pairs)
    # the reference pair count (3-sigma square x 16x16 tiles) is the oracle's K for the same Gaussians
    with torch.no_grad():
        s_ref = sds_step.SDSStep(n_gaussians=G, res=res, guidance=False, async_pair_count=False)
        s_ref.data["smpl_inputs"] = synth.random_smpl_inputs(seed=0, device=s_ref.device)
        gs = s_ref.avatar.animate(s_ref.data["smpl_inputs"])
    from dreamwaltz_g_amd import camera
    view, proj, campos, tfx, tfy = camera.raster_matrices({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in s_ref.data.items()})
    sc = dict(means3D=gs.positions.cpu(), opacities=gs.opacities.cpu(), colors=gs.colors.cpu(), scales=gs.scales.cpu(), rotations=gs.quaternions.cpu(),
              viewmatrix=view, projmatrix=proj, campos=campos, tanfovx=tfx, tanfovy=tfy, bg=torch.tensor([0.5, 0.5, 0.5]), H=res, W=res)
    ref = rc.oracle_forward(sc)
    assert s_sync.num_pairs[1] == int(ref["num_pairs"]), (s_sync.num_pairs, int(ref["num_pairs"]))
    err = (o1[1]["image"][0].permute(2, 0, 1).cpu() - torch.from_numpy(ref["color"])).abs()
    assert float(torch.quantile(err.reshape(-1)[::7], 0.999)) <= 1e-4
    for st in (s_sync, s_async):
        for _ in range(2):
            st.run()
        assert torch.isfinite(st.optimizers.buffers.flat).all() and torch.isfinite(st.optimizers.buffers.grad).all()
    assert s_async.trainer.redone_frames == 0
    # forced overflow: shrink the async capacity below the pair count -> the frame is truncated, detected in its own backward, redone
    state = s_async.scene.renderer.pair_state(s_async.device, res, res)
    state.resolve(); state.cap = 1000
    out = s_async.run()
    assert s_async.trainer.redone_frames == 1 and state.cap > 1000 and not state.overflow
    # the re-rendered frame is complete: the exact-sizing twin (same seeds, same step index; parameters equal up to the order of
    # float atomics in earlier backward passes) renders the same image -- a frame truncated to 1000 pairs would be almost empty
    ref_out = s_sync.run()
    d = (out[1]["image"] - ref_out[1]["image"]).abs().reshape(-1)
    assert float(torch.quantile(d[::5], 0.999)) < 1e-3, float(d.max())
    assert float((out[1]["alpha"] > 0.5).float().mean()) > 0.05
