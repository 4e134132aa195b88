"""Runs in its own process (tests/test_dwg_bind.py): the REAL reference modules from /root/reference (inert stand-ins for the packages
that are not installed, tests/golden/_ref_stubs.py) under dropin/dwg_bind's post-import hooks.  Prints one JSON object.

Checks: the hooks land on the attributes the reference's Trainer resolves at call time; a reference DreamWaltzG (real class; its
licensed-asset constructor replaced by hand-set attributes, as in tests/golden/capture_golden_r2.py) is adopted by name into the HIP-backed
mirror with every trainable tensor equal; the bound build_scene returns the mirror Scene whose avatar.get_optimizer(cfg) has the
reference's optimizer names."""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dropin"))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import animate as oa  # noqa: E402
import _ref_stubs  # noqa: E402
import dwg_bind  # noqa: E402


def main():
    out = {}
    SMPLX = _ref_stubs.install(oa)
    dwg_bind.install()
    import core.system.avatar as avmod
    import core.system.scene as scmod
    import core.guidance.controlnet as cnmod
    out["avatar_hooked"] = bool(getattr(avmod.build_gaussian_avatar, "__dwg_bound__", False))
    out["scene_hooked"] = bool(getattr(scmod.build_scene, "__dwg_bound__", False))
    out["guidance_hooked"] = bool(getattr(cnmod.ControlNetScoreDistillation.__init__, "__dwg_bound__", False))
    # what Trainer.init_gaussian_model / init_diffusion do at call time (trainer.py:446-453,529-530)
    from core.system.avatar import build_gaussian_avatar
    from core.system.scene import build_scene
    out["call_time_import_sees_hook"] = build_gaussian_avatar is avmod.build_gaussian_avatar and build_scene is scmod.build_scene

    # ---- a reference DreamWaltzG on a synthetic body (real classes, attributes set by hand) ------------------------------------------
    from configs import TrainConfig
    from core.human.inverse_lbs import GeneralLinearBlendSkinning
    from core.nerf.nerf_model import MLP
    from core.deformation.deform_model import DeformNetwork
    cfg = TrainConfig(); cfg.device = "cpu"
    g = torch.Generator().manual_seed(0)
    body = oa.SyntheticBody(V=300, F_=500, seed=3)
    fake = SMPLX()
    fake.NUM_JOINTS = 54; fake.NUM_BODY_JOINTS = 21
    fake.faces = body.faces.numpy(); fake.parents = torch.from_numpy(body.parents)
    for k in ("betas", "v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "pose_mean", "expr_dirs", "expression", "jaw_pose",
              "leye_pose", "reye_pose"):
        setattr(fake, k, getattr(body, k))
    fake.body_pose = torch.zeros(1, 63); fake.global_orient = torch.zeros(1, 3)
    fake.left_hand_pose = torch.zeros(1, 45); fake.right_hand_pose = torch.zeros(1, 45)
    fake.use_pca = False; fake.left_hand_components = torch.zeros(1); fake.right_hand_components = torch.zeros(1)
    glbs = GeneralLinearBlendSkinning(fake)
    nets = oa.init_avatar_networks(seed=4, table_std=0.3)

    class Encoder(nn.Module):            # the reference GridEncoder's learnable surface (its CUDA backend cannot be built here)
        def __init__(self):
            super().__init__()
            self.embeddings = nn.Parameter(nets["table"].clone())
            self.input_dim, self.num_levels, self.level_dim, self.base_resolution = 3, 16, 2, 16
            self.gridtype, self.interpolation, self.align_corners, self.log2_hashmap_size = "tiled", "smoothstep", False, 19

    N, Vp, Fp, n = 150, 40, 12, cfg.render.n_gaussians_per_triangle
    vi = torch.randperm(300, generator=g)[:Vp]
    tri = torch.stack([torch.randperm(Vp, generator=g)[:3] for _ in range(Fp)])
    m = object.__new__(avmod.MeshBindingGaussianModel)
    nn.Module.__init__(m)
    m.learn_bary_coords, m.learn_vertex_coords, m.learn_scales = True, False, True
    m._n_points_per_triangle, m._n_triangles, m._n_vertices, m._n_points = n, Fp, Vp, Fp * n
    m._bary_coords = nn.Parameter(m.initialize_bary_coords(n).clone() * (1.0 + 0.2 * torch.rand(Fp, n, 3, generator=g)))
    m._vertex_coords = nn.Parameter(body.v_template[vi].clone(), requires_grad=False)
    m.register_buffer("triangles", tri.clone())
    p2t = torch.arange(Fp)[..., None].expand(-1, n).reshape(-1)
    m.register_buffer("points_to_triangles", p2t); m.register_buffer("points_to_vertices", m.triangles[p2t])
    m._scales = nn.Parameter(0.3 + 2.0 * torch.rand(Fp * n, 3, generator=g))
    m.predefined_vertex_indices = vi; m.predefined_triangle_indices = torch.arange(Fp)

    a = object.__new__(avmod.DreamWaltzG)
    nn.Module.__init__(a)
    a.cfg, a.device = cfg, torch.device("cpu")
    a.lbs_model, a.deform_model = glbs, None
    a.smpl_canonical_inputs = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
                                   right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
    a.nerf_encoder = Encoder()
    a.register_buffer("nerf_bound", torch.tensor(2.0))
    a.nerf_opacity_and_color_net = MLP(32, 4, 64, 3, bias=True)
    a.nerf_scale_and_quaternion_net = DeformNetwork(xyz_input_ch=32, D=4, W=64, residual=False)
    a.init_offset, a.init_scale, a.max_scale = cfg.render.init_offset, cfg.render.init_scale, cfg.render.max_scale
    a._positions = nn.Parameter((torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([0.4, 0.9, 0.2]))
    a._scales = nn.Parameter(torch.log(torch.rand(N, 3, generator=g) * 0.018 + 0.002))
    a._quaternions = nn.Parameter(torch.randn(N, 4, generator=g))
    a._lbs_weights = nn.Parameter(torch.softmax(torch.randn(N, 55, generator=g), -1), requires_grad=False)
    a._betas = nn.Parameter(torch.randn(1, 300, generator=g) * 0.5, requires_grad=False)
    a._n_points, a._n_points_on_mesh = N, Fp * n
    a.learn_hand_betas = a.learn_face_betas = a.learn_betas = False
    a.nearest_triangles_buffer = dict(nearest_vertex_indices=torch.randint(0, 300, (N,), generator=g))
    a.mesh_binding_gaussians = nn.ModuleDict({"hands": m})
    a.canonical_vertices = body.v_template

    bound = dwg_bind.bind_avatar(a)
    out["bound_class"] = type(bound).__module__ + "." + type(bound).__name__
    ref_sd, own_sd = a.state_dict(), bound.state_dict()
    trainable = [k for k, p in a.named_parameters() if p.requires_grad]
    out["trainable_keys_missing"] = [k for k in trainable if k not in own_sd]
    out["trainable_max_abs_diff"] = max(float((own_sd[k] - ref_sd[k]).abs().max()) for k in trainable if k in own_sd)
    out["frozen_equal"] = bool(torch.equal(own_sd["_lbs_weights"], ref_sd["_lbs_weights"]) and torch.equal(own_sd["_betas"], ref_sd["_betas"]))
    out["lbs_buffers_equal"] = bool(torch.equal(bound.lbs_model.v_template, glbs.v_template.data) and torch.equal(bound.lbs_model.J_template, glbs.J_template))
    out["reference_attribute_passthrough"] = bool(torch.equal(bound.canonical_vertices, body.v_template))
    out["trainable_flags_kept"] = sorted(k for k, p in bound.named_parameters() if p.requires_grad) == sorted(trainable)
    out["second_bind_is_identity"] = dwg_bind.bind_avatar(bound) is bound

    scene = build_scene(cfg=cfg, avatar=bound)
    out["scene_class"] = type(scene).__module__ + "." + type(scene).__name__
    out["scene_state_dict_has_avatar_prefix"] = all(k.startswith("avatar.") for k in scene.state_dict().keys())
    out["scene_surface"] = all(hasattr(scene, k) for k in ("avatar", "avatars", "background", "renderer", "forward", "avatar_forward"))
    opts = scene.avatar.get_optimizer(cfg=cfg)
    out["optimizer_names"] = sorted(opts.keys())
    out["optimizers_have_trainer_surface"] = all(hasattr(o, k) for o in opts.values() for k in ("zero_grad", "step", "param_groups", "state_dict"))
    out["avatar_optimizer_has_update_learning_rate"] = hasattr(opts["avatar"], "update_learning_rate")
    # the reference's own avatar of another kind is left alone
    other = nn.Module()
    out["other_objects_untouched"] = dwg_bind.bind_avatar(other) is other
    dwg_bind.uninstall()
    out["uninstall_restores"] = not getattr(avmod.build_gaussian_avatar, "__dwg_bound__", False)
    print("DWG_BIND_CHECK " + json.dumps(out))


if __name__ == "__main__":
    main()
