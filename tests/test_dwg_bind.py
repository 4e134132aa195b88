"""dropin/dwg_bind: the post-import binding of the reference's main.py / Trainer to the HIP path (boundaries B3, B4, B5).

CPU (here): the real reference modules under the hooks (tests/ref_bind_check.py in its own process: it installs inert stand-ins for
missing packages into that process' import system); sitecustomize activation.  -m gpu: stand-in objects with the reference's attribute
names run through the bound seams on the kernels."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_hooks_on_the_real_reference_modules_and_adoption_by_name():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_bind_check.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DWG_BIND_CHECK ")][-1]
    d = json.loads(line[len("DWG_BIND_CHECK "):])
    for k in ("avatar_hooked", "scene_hooked", "guidance_hooked", "call_time_import_sees_hook", "frozen_equal", "lbs_buffers_equal",
              "reference_attribute_passthrough", "trainable_flags_kept", "second_bind_is_identity", "scene_state_dict_has_avatar_prefix",
              "scene_surface", "optimizers_have_trainer_surface", "avatar_optimizer_has_update_learning_rate", "other_objects_untouched",
              "uninstall_restores"):
        assert d[k] is True, (k, d[k])
    assert d["bound_class"] == "dreamwaltz_g_amd.avatar.DreamWaltzG" and d["scene_class"] == "dreamwaltz_g_amd.scene.Scene"
    assert d["trainable_keys_missing"] == [] and d["trainable_max_abs_diff"] == 0.0
    assert d["optimizer_names"] == ["avatar", "mesh_hands", "nerf"]          # avatar.py:1590-1635 with the default learn flags


def test_sitecustomize_installs_the_hooks_only_when_asked():
    code = "import sys; print(any(type(f).__name__ == '_HookFinder' for f in sys.meta_path))"
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "dropin") + os.pathsep + ROOT)
    env.pop("DWG_BIND", None)
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, DWG_BIND="1"), timeout=120)
    assert off.stdout.strip() == "False" and on.stdout.strip() == "True", (off.stdout, off.stderr, on.stdout, on.stderr)


def test_unet_config_from_a_diffusers_style_config():
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import sd15
    import types
    cfg = types.SimpleNamespace(in_channels=4, out_channels=4, block_out_channels=[32, 64], layers_per_block=1, attention_head_dim=4,
                                cross_attention_dim=48, norm_num_groups=8, down_block_types=["CrossAttnDownBlock2D", "DownBlock2D"])
    u = sd15.unet_config_from(cfg)
    assert u.block_out_channels == (32, 64) and u.heads == 4 and u.cross_dim == 48 and u.attn_blocks == (True, False) and u.groups == 8
    with pytest.raises(NotImplementedError):
        sd15.unet_config_from(dict(transformer_layers_per_block=2))
    assert sd15.vae_config_from(None).scaling_factor == 0.18215


def test_plan_dtype_follows_the_reference_pipeline_dtype(monkeypatch):
    """`--guide.dtype fp16` (the reference loads UNet / ControlNet / VAE in torch.float16: core/guidance/basic.py:233) -> fp16 plans; its
    fp32 default -> the fp32-grade f32x plans (never a narrower type on the binding's own initiative); DWG_BIND_DTYPE and the explicit
    argument override."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        import dwg_bind
    finally:
        sys.path.pop(0)
    monkeypatch.delenv("DWG_BIND_DTYPE", raising=False)
    half, full = types.SimpleNamespace(torch_dtype=torch.float16), types.SimpleNamespace(torch_dtype=torch.float32)
    assert dwg_bind.plan_dtype_for(half) == "f16" and dwg_bind.plan_dtype_for(full) == "f32x" and dwg_bind.plan_dtype_for(object()) == "f32x"
    assert dwg_bind.plan_dtype_for(half, "f32") == "f32"
    monkeypatch.setenv("DWG_BIND_DTYPE", "f32")
    assert dwg_bind.plan_dtype_for(half) == "f32" and dwg_bind.plan_dtype_for(full, "bf16") == "bf16"


# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_bound_avatar_renders_like_the_avatar_it_adopted():
    """A stand-in carrying the reference DreamWaltzG's attribute names (built FROM one of our avatars, so the expected output is known):
    from_reference adopts it and animate() is bit-identical."""
    import types
    import dwg_import  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    import dwg_bind
    from dreamwaltz_g_amd import sds_step, synth
    dev = torch.device("cuda:0")
    src, N, M = sds_step.build_synthetic_avatar(6000, dev, seed=3)

    class RefLike(torch.nn.Module):
        pass
    RefLike.__name__ = "DreamWaltzG"
    lbs = torch.nn.Module(); lbs.__class__ = type("GeneralLinearBlendSkinning", (torch.nn.Module,), {})
    L = src.lbs_model
    for k in ("v_template", "posedirs", "J_regressor", "lbs_weights", "betas", "expression", "pose_mean", "jaw_pose", "leye_pose", "reye_pose"):
        setattr(lbs, k, torch.nn.Parameter(getattr(L, k).clone(), requires_grad=False))
    lbs.shapedirs = torch.nn.Parameter(L.shapedirs_all[..., :300].clone(), requires_grad=False)
    lbs.expr_dirs = torch.nn.Parameter(L.shapedirs_all[..., 300:].clone(), requires_grad=False)
    lbs.parents, lbs.use_smplx, lbs.NUM_BODY_JOINTS = L.parents.long(), True, 21
    ref = RefLike()
    ref.lbs_model, ref.deform_model = lbs, None
    for k in ("_positions", "_scales", "_quaternions", "_lbs_weights", "_betas"):
        setattr(ref, k, torch.nn.Parameter(getattr(src, k).detach().clone(), requires_grad=getattr(src, k).requires_grad))
    ref.smpl_canonical_inputs = src.smpl_canonical_inputs
    ref.register_buffer("nerf_bound", torch.tensor(2.0, device=dev))
    ref.init_offset, ref.init_scale, ref.max_scale = src.init_offset, src.init_scale, src.max_scale
    ref.nerf_encoder, ref.nerf_opacity_and_color_net, ref.nerf_scale_and_quaternion_net = (
        src.nerf_encoder, src.nerf_opacity_and_color_net, src.nerf_scale_and_quaternion_net)
    ref.mesh_binding_gaussians = src.mesh_binding_gaussians
    ref.learn_hand_betas = ref.learn_face_betas = False
    ref.nearest_triangles_buffer = {"nearest_vertex_indices": None}
    ref.cfg = None
    bound = dwg_bind.bind_avatar(ref)
    assert type(bound).__module__ == "dreamwaltz_g_amd.avatar" and bound is not src
    pose = synth.random_smpl_inputs(seed=5, device=dev)
    with torch.no_grad():
        a, b = src.animate(pose), bound.animate(pose)
    for f in ("positions", "opacities", "colors", "quaternions", "scales"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.gpu
def test_bound_guidance_seams_run_the_hip_plans():
    """A stand-in with the reference guidance object's attribute names (pipe.unet / controlnet / pipe.vae with config + state_dict(), device,
    default_image_size, cfg, timestep): bind_guidance feeds the plans from the modules' state_dict()s and binds _predict / encode_images."""
    import types
    import dwg_import  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    import dwg_bind
    from dreamwaltz_g_amd import configs, guidance as gd, sd15
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ucfg = sd15.UNetConfig(block_out_channels=(32, 64), layers_per_block=1, heads=4, cross_dim=48, groups=8, attn_blocks=(True, False),
                           cond_channels=(8, 16))
    vcfg = sd15.VAEConfig(block_out_channels=(16, 32), layers_per_block=1, groups=8)
    usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=0)
    csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=1)
    vsd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=2)
    vsd_full = dict(vsd); vsd_full["decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3)        # the real AutoencoderKL also carries a decoder

    class Mod:
        def __init__(self, sd, config):
            self._sd, self.config, self.where = sd, config, "cuda"

        def state_dict(self):
            return self._sd

        def to(self, d):
            self.where = str(d)
            return self
    ns = types.SimpleNamespace
    unet = Mod(usd, ns(in_channels=4, out_channels=4, block_out_channels=[32, 64], layers_per_block=1, attention_head_dim=4,
                       cross_attention_dim=48, norm_num_groups=8, down_block_types=["CrossAttnDownBlock2D", "DownBlock2D"]))
    # the ControlNet's conditioning embedding widths are not in the UNet config: default (16, 32, 96, 256) -> give the stand-in SD-1.5's
    ucfg_bound = sd15.unet_config_from(unet.config)
    csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg_bound), seed=1)
    cnet = Mod(csd, ns())
    vae = Mod(vsd_full, ns(in_channels=3, block_out_channels=[16, 32], layers_per_block=1, latent_channels=4, norm_num_groups=8, scaling_factor=0.18215))

    class RefGuidance:
        def encode_images(self, images):
            raise AssertionError("the reference's encode_images must not run for tensors once bound")
    ref = RefGuidance()
    ref.pipe, ref.controlnet, ref.device = ns(unet=unet, vae=vae), cnet, dev
    ref.default_image_size, ref.cfg, ref.timestep = 64, configs.GuideConfig(), torch.tensor([500], device=dev)
    assert dwg_bind.bind_guidance(ref) is ref and dwg_bind.bind_guidance(ref) is ref
    assert unet.where == "cpu" and cnet.where == "cpu" and vae.where == "cuda"
    direct = gd.ControlNetScoreDistillation(dev, unet_cfg=ucfg_bound, vae_cfg=sd15.vae_config_from(vae.config), unet_sd=usd, controlnet_sd=csd,
                                            vae_sd=vsd, image_hw=64, cfg=configs.GuideConfig(), dtype=dwg_bind.plan_dtype_for(ref))
    assert direct.dtype_name == "f32x"          # a reference pipeline without torch_dtype = its fp32 default -> the fp32-grade plans
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 4, 32, 32, generator=g).repeat(2, 1, 1, 1).to(dev)
    text = torch.randn(2, 77, 48, generator=g).to(dev)
    cond = torch.rand(1, 3, 256, 256, generator=g).to(dev)
    direct.timestep = ref.timestep
    # NB latent 32x32 at image 64 would not match: the plans were built for image_hw 64 -> 32x32 latents with a 2-level VAE
    a = ref._predict(lat, text, cond).clone()
    b = direct._predict(lat, text, cond).clone()
    assert torch.equal(a, b)
    img = torch.rand(1, 3, 64, 64, generator=g).to(dev).requires_grad_(True)
    torch.manual_seed(7); z1 = ref.encode_images(img)
    torch.manual_seed(7); z2 = direct.encode_images(img.detach())
    assert torch.equal(z1, z2)
    z1.sum().backward()
    assert img.grad is not None and float(img.grad.abs().sum()) > 0
    # round 6: the f32x range scan runs after EVERY one of the first 20 bound calls (a run that saturates at call 3 must not train 197
    # more calls on clipped values before a word is said), then every 200th
    assert ref.hip_range_checks == 1 and ref.hip_range_report["ok"]
    for _ in range(4):
        ref._predict(lat, text, cond)
    assert ref.hip_range_checks == 5
