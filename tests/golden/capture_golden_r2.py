"""Round-2 golden fixtures: runs MORE of the reference's own in-repo code in this container and stores inputs + outputs in
tests/golden/reference_golden_r2.npz (data only; what travels to the GPU box).  Run here only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_golden_r2.py

How the reference code is reached (tests/golden/_ref_stubs.py):
  * inert packages satisfy imports (cv2, nvdiffrast, diffusers, ...); no captured arithmetic goes through them;
  * smplx.lbs / pytorch3d.transforms are ARITHMETIC stand-ins (oracle.animate) -> keys under "sd." (stub dependent) pin the
    reference's in-repo algebra and orchestration, not those third-party functions;
  * objects whose constructors need licensed assets (SMPL-X file, NeRF checkpoint) are created with object.__new__ and given
    exactly the attributes the captured METHODS read; the methods themselves are the reference's, unmodified;
  * the grid-encoder backend (CUDA JIT) and the rasterizer (third-party CUDA package) are replaced by the CPU oracle at the
    same seam the reference calls them through, so `animate`, `GaussianRenderer.render` and `Scene.forward` run as written.

Groups of keys:  mesh.* normal.* mlp.* act.* nrt.* cam.* rs.* cov3d.* pgc.* text.* opt.* img.* rt.* (direct or inert-only)
                 sd.animate.* sd.animate_betas.* sd.scene.* sd.invlbs.* sd.sds.*  (stub dependent)
"""
import math
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

from oracle import animate as oa  # noqa: E402
from oracle import raster as oraster  # noqa: E402
from oracle import sd15 as osd  # noqa: E402
import _ref_stubs  # noqa: E402

OUT = {}


def put(key, v):
    if torch.is_tensor(v):
        v = v.detach().cpu().numpy()
    OUT[key] = np.asarray(v)


def small_mesh(seed=7, Vp=60, Fp=40):
    g = torch.Generator().manual_seed(seed)
    verts = (torch.rand(Vp, 3, generator=g) * 2 - 1) * torch.tensor([0.1, 0.1, 0.05])
    tri = torch.stack([torch.randperm(Vp, generator=g)[:3] for _ in range(Fp)])
    return verts, tri


def make_meshbind(avmod, cfg, verts, tri, seed=8):
    """A real MeshBindingGaussianModel whose attributes are set by hand (its constructor needs the SMPL-X template)."""
    g = torch.Generator().manual_seed(seed)
    Fp, n = tri.shape[0], cfg.render.n_gaussians_per_triangle
    m = object.__new__(avmod.MeshBindingGaussianModel)
    nn.Module.__init__(m)
    m.learn_bary_coords, m.learn_vertex_coords, m.learn_scales = True, False, True
    m._n_points_per_triangle, m._n_triangles, m._n_vertices, m._n_points = n, Fp, verts.shape[0], Fp * n
    bary = m.initialize_bary_coords(n).clone()
    bary = bary * (1.0 + 0.2 * torch.rand(bary.shape, generator=g))          # un-normalised on purpose (checklist Q5)
    m._bary_coords = nn.Parameter(bary)
    m._vertex_coords = nn.Parameter(verts.clone(), requires_grad=False)
    m.register_buffer('triangles', tri.clone())
    p2t = torch.arange(Fp)[..., None].expand(-1, n).reshape(-1)
    m.register_buffer('points_to_triangles', p2t)
    m.register_buffer('points_to_vertices', m.triangles[p2t])
    m._scales = nn.Parameter(0.3 + 2.0 * torch.rand(Fp * n, 3, generator=g))   # crosses both clamp bounds
    m.predefined_vertex_indices = torch.arange(verts.shape[0])
    m.predefined_triangle_indices = torch.arange(Fp)
    return m


def main():
    SMPLX = _ref_stubs.install(oa)
    from configs import TrainConfig
    cfg = TrainConfig()
    g = torch.Generator().manual_seed(0)

    # ---- normal.*  utils/mesh.py:34-94 compute_normal (inert imports only)
    from utils.mesh import compute_normal
    verts, tri = small_mesh()
    vn, fn = compute_normal(verts, tri)
    put("normal.verts", verts); put("normal.tri", tri); put("normal.vn", vn); put("normal.fn", fn)

    # ---- mesh.*  avatar.py:1016-1079 (matrix_to_quaternion / standardize come from the arithmetic stand-in)
    import core.system.avatar as avmod
    mb = make_meshbind(avmod, cfg, verts, tri)
    vobs = verts + 0.01 * torch.randn(verts.shape, generator=g)
    pos = mb.get_positions(vertex_coords=vobs)
    sc, qu = mb.get_scales_and_quaternions(vertex_coords=vobs, positions=pos)
    put("mesh.bary", mb._bary_coords); put("mesh.scales_raw", mb._scales); put("mesh.vobs", vobs)
    put("mesh.positions", pos); put("mesh.scales", sc); put("mesh.quaternions", qu)

    # ---- mlp.*  nerf_model.py:12-33
    from core.nerf.nerf_model import MLP
    torch.manual_seed(1)
    mlp = MLP(32, 4, 64, 3, bias=True)
    x = torch.randn(37, 32, generator=g) * 0.5
    for k, v in mlp.state_dict().items():
        put("mlp.sd." + k, v)
    put("mlp.x", x); put("mlp.y", mlp(x))

    # ---- act.*  gaussian_model.py:25-56 activations
    from core.gaussian.gaussian_model import GaussianModel
    gm = GaussianModel()
    gm._scales = torch.randn(21, 3, generator=g); gm._quaternions = torch.randn(21, 4, generator=g)
    gm._opacities = torch.randn(21, 1, generator=g)
    put("act.scales_raw", gm._scales); put("act.quats_raw", gm._quaternions); put("act.opac_raw", gm._opacities)
    put("act.scales", gm.get_scales()); put("act.scales_mean", gm.get_scales(return_means=True))
    put("act.quats", gm.get_quaternions()); put("act.opac", gm.get_opacities())
    put("act.inv_sigmoid", gm.opacity_inverse_activation(gm.get_opacities()))

    # ---- cam.*  data/camera/utils.py:62-201
    import data.camera.utils as cu
    rad = torch.tensor([2.0, 1.5, 3.1, 2.0]); az = torch.tensor([30.0, 200.0, -45.0, 0.0]); el = torch.tensor([80.0, 60.0, 110.0, 90.0])
    E, C2W = cu.to_extrinsic(rad, az, el)
    fov = torch.tensor([55.0, 40.0, 70.0, 55.0])
    tanfov = cu.get_tan_half_fov(fov)
    P = cu.to_projection(tanfov, z_near=0.01, z_far=1000.0)
    P_wide = cu.to_projection(tanfov, z_near=0.01, z_far=1000.0, aspect_wh=640 / 480)
    P_x = cu.to_projection(tanfov, z_near=0.01, z_far=1000.0, tanfov_x=tanfov * 1.5)
    E_at, C2W_at = cu.to_extrinsic(rad, az, el, at_vector=((0.1, -0.2, 0.05),))
    for k, v in dict(radius=rad, azimuth=az, elevation=el, fov=fov, tanfov=tanfov, extrinsic=E, c2w=C2W, projection=P,
                     projection_wide=P_wide, projection_tanfov_x=P_x, extrinsic_at=E_at, c2w_at=C2W_at).items():
        put("cam." + k, v)

    # ---- rs.*  gaussian_renderer.py:23-70 with a RECORDING rasterizer at the import seam
    import core.gaussian.gaussian_renderer as gr

    class RecSettings(types.SimpleNamespace):
        pass

    class RecRasterizer:
        def __init__(self, raster_settings):
            self.raster_settings = raster_settings
    gr.GaussianRasterizationSettings = lambda **kw: RecSettings(**kw)
    gr.GaussianRasterizer = RecRasterizer
    rend = gr.GaussianRenderer(sh_levels=4, bg_color=(0.5, 0.5, 0.5))
    data = dict(extrinsic=E[:1], projection=P[:1], c2w=C2W[:1], tanfov=tanfov[:1], image_height=96, image_width=96)
    rs = rend.build_gaussian_rasterizer(data).raster_settings
    for k in ("viewmatrix", "projmatrix", "campos", "bg"):
        put("rs." + k, getattr(rs, k))
    put("rs.scalars", np.array([rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.scale_modifier, rs.image_height, rs.image_width], dtype=np.float64))
    data_x = dict(data, tanfov_x=tanfov[:1] * 1.25, image_width=120)
    rsx = rend.build_gaussian_rasterizer(data_x).raster_settings
    put("rs.scalars_x", np.array([rsx.tanfovx, rsx.tanfovy], dtype=np.float64))

    # ---- cov3d.*  gaussian_renderer.py:107-128 (quaternion_to_matrix from the arithmetic stand-in)
    s = torch.rand(17, 3, generator=g) * 0.05; q = torch.randn(17, 4, generator=g)
    put("cov3d.scales", s); put("cov3d.quats", q); put("cov3d.out", gr.GaussianRenderer.compute_3d_covariance(s, q))

    # ---- pgc.*  core/guidance/pgc.py:15-43 build_grad_hook_func
    from core.guidance.pgc import build_grad_hook_func
    grad = torch.randn(1, 3, 16, 16, generator=g) * torch.rand(1, 3, 16, 16, generator=g) * 3
    grad[0, 0, 0, :4] = 0.0
    mask = (torch.rand(1, 1, 16, 16, generator=g) > 0.3).float()
    put("pgc.grad", grad); put("pgc.mask", mask)
    put("pgc.clip", build_grad_hook_func(True, False, 1.5)(grad.clone()))
    put("pgc.clip_mask", build_grad_hook_func(True, False, 0.7, mask=mask)(grad.clone()))
    put("pgc.norm", build_grad_hook_func(False, True, 1.0)(grad.clone()))
    put("pgc.clip_norm", build_grad_hook_func(True, True, 2.0)(grad.clone()))

    # ---- text.*  core/guidance/text.py:36-154 view-dependent prompt index
    from core.guidance.text import TextAugmentation
    ta = TextAugmentation("a person", cfg.prompt)
    azs = np.arange(0.0, 360.0, 7.5); els = np.array([5.0, 29.0, 31.0, 60.0, 90.0, 120.0, 149.0, 151.0, 175.0])
    res = np.zeros((len(els), len(azs)), dtype=np.int64)
    for i, e in enumerate(els):
        for j, a in enumerate(azs):
            res[i, j] = int(ta(torch.tensor([a]), torch.tensor([e]))[0])
    put("text.azimuths", azs); put("text.elevations", els); put("text.index", res)
    put("text.ranges", np.array(list(ta.azimuth_range) + list(ta.elevation_range), dtype=np.float64))
    OUT["text.texts"] = np.array(ta.texts)
    put("text.cfg", np.array([cfg.prompt.angle_front, cfg.prompt.angle_overhead], dtype=np.float64))

    # ---- opt.*  gaussian_optimizer.py:49-141 + avatar.py:1590-1635 (group names / learning rates per iteration)
    from core.gaussian.gaussian_optimizer import GaussianOptimizer, OptimizationParams
    fm = types.SimpleNamespace(_positions=nn.Parameter(torch.zeros(4, 3)), _sh_features_dc=None, _sh_features_rest=None,
                               _opacities=None, _scales=nn.Parameter(torch.zeros(4, 3)), _quaternions=nn.Parameter(torch.zeros(4, 4)))
    iters = 10000
    op = OptimizationParams(iterations=iters, position_lr_init=cfg.render.position_lr_init, position_lr_final=cfg.render.position_lr_final,
                            position_lr_delay_mult=0.01, position_lr_max_steps=iters * 2, feature_lr=cfg.render.feature_lr,
                            opacity_lr=cfg.render.opacity_lr, scaling_lr=cfg.render.scaling_lr, rotation_lr=cfg.render.rotation_lr)
    go = GaussianOptimizer(fm, op)
    its = np.array([0, 1, 50, 500, 5000, 9999, 20000]); spatial = 2.0 * math.tan(math.radians(55.0) / 2)
    lrs = np.zeros((len(its), 3))
    for i, it in enumerate(its):
        go.update_learning_rate(spatial, int(it))
        lrs[i] = [pg['lr'] for pg in go.param_groups]
    OUT["opt.group_names"] = np.array([pg['name'] for pg in go.param_groups])
    put("opt.iterations", its); put("opt.lrs", lrs); put("opt.spatial_scale", np.array([spatial]))
    put("opt.cfg", np.array([cfg.render.position_lr_init, cfg.render.position_lr_final, cfg.render.scaling_lr, cfg.render.rotation_lr,
                             cfg.nerf.lr, cfg.render.betas_lr, cfg.render.lbs_lr], dtype=np.float64))
    put("opt.adam", np.array([go.optimizer.defaults['eps'], go.optimizer.defaults['betas'][0], go.optimizer.defaults['betas'][1]]))

    # ---- img.*  controlnet.py:33-55 prepare_image (PIL LANCZOS resize -> float/255 -> NCHW -> repeat)
    from PIL import Image
    from core.guidance.controlnet import BasicControlNetScoreDistillation
    rs_ = np.random.RandomState(0)
    arr = rs_.randint(0, 256, (48, 40, 3), dtype=np.uint8)
    pil = Image.fromarray(arr)
    o = BasicControlNetScoreDistillation.prepare_image(None, [pil], width=64, height=64, batch_size=2, num_images_per_prompt=1,
                                                       device="cpu", dtype=torch.float32)
    put("img.in", arr); put("img.out", o)
    arr2 = rs_.randint(0, 256, (64, 64, 3), dtype=np.uint8)
    o2 = BasicControlNetScoreDistillation.prepare_image(None, Image.fromarray(arr2), width=64, height=64, batch_size=2,
                                                        num_images_per_prompt=1, device="cpu", dtype=torch.float32)
    put("img.in_same", arr2); put("img.out_same", o2)

    # ---- rt.*  inverse_lbs.py:15-260 RigidTransform (in-repo algebra; the `matrix`/`quaternion` modes use the stand-in)
    from core.human.inverse_lbs import RigidTransform, GeneralLinearBlendSkinning
    A = torch.eye(4).repeat(6, 1, 1)
    for j in range(6):
        A[j, :3, :3] = oa.batch_rodrigues(torch.randn(1, 3, generator=g))[0]
        A[j, :3, 3] = torch.randn(3, generator=g) * 0.2
    A[:, 3, :] = torch.tensor([0.1, -0.2, 0.3, 0.9])        # junk last row: inverse() must overwrite it IN PLACE (checklist Q7)
    src = A.clone()
    rt = RigidTransform(SE3=src)
    inv = rt.inverse()
    put("rt.A", A); put("rt.inverse", inv.SE3); put("rt.source_after_inverse", src)
    w = torch.softmax(torch.randn(9, 6, generator=g), -1)
    put("rt.w", w); put("rt.weighted", RigidTransform(SE3=src.clone()).weight(w).SE3)
    B = torch.eye(4).repeat(6, 1, 1); B[:, :3, 3] = torch.randn(6, 3, generator=g)
    put("rt.B", B); put("rt.compose", RigidTransform(SE3=src.clone()).compose(RigidTransform(SE3=B), RigidTransform(SE3=src.clone())).SE3)
    pts = torch.randn(9, 3, generator=g)
    wt = RigidTransform(SE3=src.clone()).weight(w)
    put("rt.pts", pts); put("rt.inv_points", RigidTransform._inverse_transform_points(pts, R=wt.R, T=wt.T))
    qs = torch.randn(9, 4, generator=g)
    put("rt.q", qs)
    put("sd.rt.quat_mode_matrix", RigidTransform(SE3=src.clone()).transform_quaternions(qs, weights=w, rotation_mode='matrix'))
    put("sd.rt.quat_mode_quaternion", RigidTransform(SE3=src.clone()).transform_quaternions(qs, weights=w, rotation_mode='quaternion'))
    idx = torch.randint(0, 6, (9,), generator=g)
    put("rt.idx", idx); put("sd.rt.quat_indexed", RigidTransform(SE3=src.clone()).transform_quaternions(qs, indices=idx, rotation_mode='matrix'))

    # ---- nrt.*  avatar.py:1464-1498 non_rigid_transform on a hand-made self (every flag combination that does not assert)
    from core.gaussian.gaussian_utils import GaussianOutput

    def fake_avatar(**flags):
        a = object.__new__(avmod.DreamWaltzG)
        nn.Module.__init__(a)
        GaussianModel.__init__(a)
        d = dict(use_non_rigid_offsets=True, use_non_rigid_scales=True, use_non_rigid_rotations=False, non_rigid_scale_mode='add',
                 non_rigid_rotation_mode='add', learn_scale=True, learn_quaternions=True, init_scale=cfg.render.init_scale,
                 init_offset=cfg.render.init_offset, max_scale=cfg.render.max_scale)
        d.update(flags)
        for k, v in d.items():
            setattr(a, k, v)
        return a
    N = 23
    base = dict(_scales=torch.randn(N, 3, generator=g) * 0.3 - 4.0, _quaternions=torch.randn(N, 4, generator=g),
                positions=torch.randn(N, 3, generator=g), offsets=torch.randn(N, 3, generator=g), mlp_scales=torch.randn(N, 3, generator=g),
                mlp_quats=torch.randn(N, 4, generator=g))
    for k, v in base.items():
        put("nrt." + k, v)
    variants = {"default": {}, "mul": dict(non_rigid_rotation_mode='mul'), "no_learn_scale": dict(learn_scale=False),
                "no_nr_scales": dict(use_non_rigid_scales=False), "no_offsets": dict(use_non_rigid_offsets=False)}
    for name, fl in variants.items():
        a = fake_avatar(**fl)
        a._scales = nn.Parameter(base["_scales"].clone()); a._quaternions = nn.Parameter(base["_quaternions"].clone())
        go_ = GaussianOutput(positions=base["positions"].clone(), offsets=base["offsets"].clone(), scales=base["mlp_scales"].clone(),
                             quaternions=base["mlp_quats"].clone())
        r = a.non_rigid_transform(go_)
        put("nrt.%s.positions" % name, r.positions); put("nrt.%s.scales" % name, r.scales); put("nrt.%s.quaternions" % name, r.quaternions)
    for name, fl in {"rot_add": dict(use_non_rigid_rotations=True), "rot_mul": dict(use_non_rigid_rotations=True, non_rigid_rotation_mode='mul'),
                     "rot_nolearn": dict(use_non_rigid_rotations=True, learn_quaternions=False)}.items():
        a = fake_avatar(**fl)
        a._scales = nn.Parameter(base["_scales"].clone()); a._quaternions = nn.Parameter(base["_quaternions"].clone())
        go_ = GaussianOutput(positions=base["positions"].clone(), offsets=base["offsets"].clone(), scales=base["mlp_scales"].clone(),
                             quaternions=base["mlp_quats"].clone())
        put("sd.nrt.%s.quaternions" % name, a.non_rigid_transform(go_).quaternions)

    # ---- sd.animate.*  the reference's DreamWaltzG.animate, as written, on a synthetic body
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
    nets = oa.init_avatar_networks(seed=4, table_std=0.3)

    class OracleEncoder(nn.Module):                     # stands at the `_gridencoder` seam (CUDA JIT cannot run here)
        def __init__(self):
            super().__init__()
            self.embeddings = nn.Parameter(nets["table"].clone())

        def forward(self, x, bound=1):
            return oa.grid_encode((x + bound) / (2 * bound), self.embeddings, nets["offsets"], nets["per_level_scale"])

    from core.deformation.deform_model import DeformNetwork
    N = 150
    Vp = 40
    vi = torch.randperm(300, generator=g)[:Vp]
    tri_l = torch.stack([torch.randperm(Vp, generator=g)[:3] for _ in range(12)])
    cnl = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
               right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
    cnl["body_pose"][0, 2] = 0.5; cnl["body_pose"][0, 5] = -0.5          # an A-pose-like canonical pose (non-trivial canonical LBS)
    obs = oa.random_smpl_inputs(seed=9)

    def build_avatar(learn_hand_betas):
        a = fake_avatar()
        a.device = torch.device("cpu")
        a.lbs_model = glbs
        a.smpl_canonical_inputs = cnl
        a.use_joint_shape_offsets = a.use_vertex_shape_offsets = a.use_vertex_pose_offsets = False
        a.render_mesh_binding_3d_gaussians_only = a.render_unconstrained_3d_gaussians_only = False
        a.use_nerf_encoded_position = True
        a.nerf_encoder = OracleEncoder()
        a.register_buffer('nerf_bound', torch.tensor(2.0))
        mlp_s = MLP(32, 4, 64, 3, bias=True)
        with torch.no_grad():
            for l in range(3):
                mlp_s.net[l].weight.copy_(nets["static_w"][l]); mlp_s.net[l].bias.copy_(nets["static_b"][l])
        a.nerf_opacity_and_color_net = mlp_s
        torch.manual_seed(5)      # identical network weights in every variant built here
        dn = DeformNetwork(xyz_input_ch=32, D=4, W=64, residual=False)
        a.nerf_scale_and_quaternion_net = dn
        gg = torch.Generator().manual_seed(21)
        a._positions = nn.Parameter((torch.rand(N, 3, generator=gg) * 2 - 1) * torch.tensor([0.4, 0.9, 0.2]))
        a._scales = nn.Parameter(torch.log(torch.rand(N, 3, generator=gg) * 0.018 + 0.002))
        a._quaternions = nn.Parameter(torch.randn(N, 4, generator=gg))
        a._lbs_weights = nn.Parameter(torch.softmax(torch.randn(N, 55, generator=gg), -1) * 1.7, requires_grad=False)   # un-normalised
        a._n_points = N
        a.nearest_triangles_buffer = dict(nearest_vertex_indices=torch.randint(0, 300, (N,), generator=gg))
        a.learn_hand_betas, a.learn_face_betas = learn_hand_betas, False
        a.learn_betas = learn_hand_betas
        a._betas = nn.Parameter(torch.randn(1, 300, generator=gg) * 0.5, requires_grad=learn_hand_betas)
        mbm = make_meshbind(avmod, cfg, body.v_template[vi], tri_l, seed=31)
        mbm.predefined_vertex_indices = vi
        a.mesh_binding_gaussians = nn.ModuleDict({"hands": mbm})
        return a, dn

    for tag, lhb in (("animate", False), ("animate_betas", True)):
        a, dn = build_avatar(lhb)
        out = a.animate(obs)
        pre = "sd.%s." % tag
        for f in ("positions", "opacities", "colors", "quaternions", "scales"):
            put(pre + "out." + f, getattr(out, f))
        wsum = dict(positions=torch.randn(out.positions.shape, generator=g), opacities=torch.randn(out.opacities.shape, generator=g),
                    colors=torch.randn(out.colors.shape, generator=g), quaternions=torch.randn(out.quaternions.shape, generator=g),
                    scales=torch.randn(out.scales.shape, generator=g))
        loss = sum((getattr(out, f) * wsum[f]).sum() for f in wsum)
        loss.backward()
        for f, v in wsum.items():
            put(pre + "lossw." + f, v)
        put(pre + "grad._positions", a._positions.grad); put(pre + "grad._scales", a._scales.grad)
        put(pre + "grad._quaternions", a._quaternions.grad); put(pre + "grad.table", a.nerf_encoder.embeddings.grad)
        put(pre + "grad.bary", a.mesh_binding_gaussians["hands"]._bary_coords.grad)
        put(pre + "grad.mesh_scales", a.mesh_binding_gaussians["hands"]._scales.grad)
        if lhb:
            put(pre + "grad._betas", a._betas.grad)
        if tag == "animate":
            for k, v in dn.state_dict().items():
                put("sd.animate.deform." + k, v)
            for k in ("_positions", "_scales", "_quaternions", "_lbs_weights", "_betas"):
                put("sd.animate.param." + k, getattr(a, k))
            mbm = a.mesh_binding_gaussians["hands"]
            put("sd.animate.mesh.vertex_indices", vi); put("sd.animate.mesh.triangles", tri_l)
            put("sd.animate.mesh.bary", mbm._bary_coords); put("sd.animate.mesh.scales", mbm._scales)
            put("sd.animate.mesh.vertex_coords", mbm._vertex_coords)
            for k, v in cnl.items():
                put("sd.animate.cnl." + k, v)
            for k, v in obs.items():
                put("sd.animate.obs." + k, v)
            OUT["sd.animate.state_dict_keys"] = np.array(sorted(a.state_dict().keys()))
            put("sd.animate.nets_seed", np.array([4])); put("sd.animate.table_std", np.array([0.3]))
            put("sd.animate.body", np.array([300, 500, 3]))
            # inverse_lbs_transform (avatar.py:1390-1424): general 3x3 inverse of the blended matrix (checklist Q8)
            pin = torch.randn(N, 3, generator=g) * 0.3
            put("sd.invlbs.in", pin)
            put("sd.invlbs.out", a.inverse_lbs_transform(pin, glbs.forward(**cnl)[-1]))

    # ---- sd.scene.*  Scene.forward -> GaussianRenderer.render with the CPU oracle standing at the rasterizer seam
    import core.system.scene as scmod
    import importlib
    importlib.reload(gr)       # undo the recording seam

    class OracleSettings(types.SimpleNamespace):
        pass

    class OracleRasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp):
            r = self.rs
            o = oraster.forward(means3D.detach().numpy(), opacities.detach().numpy().reshape(-1), r.viewmatrix.numpy(), r.projmatrix.numpy(),
                                r.tanfovx, r.tanfovy, r.bg.numpy(), r.image_height, r.image_width,
                                colors=None if colors_precomp is None else colors_precomp.detach().numpy(),
                                shs=None if shs is None else shs.detach().numpy(),
                                scales=None if scales is None else scales.detach().numpy(),
                                rotations=None if rotations is None else rotations.detach().numpy(),
                                cov3D=None if cov3D_precomp is None else cov3D_precomp.detach().numpy(),
                                campos=r.campos.numpy(), sh_degree=r.sh_degree)
            return (torch.from_numpy(o["color"]), torch.from_numpy(o["radii"]), torch.from_numpy(o["depth"])[None], torch.from_numpy(o["alpha"])[None])
    gr.GaussianRasterizationSettings = lambda **kw: OracleSettings(**kw)
    gr.GaussianRasterizer = OracleRasterizer
    scmod.GaussianRenderer = gr.GaussianRenderer
    a, _ = build_avatar(False)
    sc_ = object.__new__(scmod.Scene)
    nn.Module.__init__(sc_)
    sc_.device = torch.device("cpu"); sc_.avatar = a; sc_.avatars = None; sc_.background = None
    sc_.pure_colors = scmod.PureColorBackground()
    sc_.renderer = gr.GaussianRenderer(sh_levels=cfg.render.sh_levels, bg_color=(0.0, 0.0, 0.0))
    sc_.use_zero_scales = False; sc_.use_constant_colors = False; sc_.use_constant_opacities = False; sc_.use_fixed_n_gaussians = False
    sc_.avatar_transl = None; sc_.avatar_scale = None
    E1, C1 = cu.to_extrinsic(torch.tensor([2.0]), torch.tensor([20.0]), torch.tensor([85.0]))
    t1 = cu.get_tan_half_fov(torch.tensor([55.0]))
    data = dict(extrinsic=E1, projection=cu.to_projection(t1, 0.01, 1000.0), c2w=C1, tanfov=t1, image_height=64, image_width=64)
    with torch.no_grad():
        o_none = sc_.forward(data, smpl_observed_inputs=obs, use_densifier=True, bg_mode=None)
        o_white = sc_.forward(data, smpl_observed_inputs=obs, use_densifier=False, bg_mode='white')
        sc_.avatar_transl = torch.tensor([0.05, -0.02, 0.1]); sc_.avatar_scale = torch.tensor(1.2)
        o_ts = sc_.forward(data, smpl_observed_inputs=obs, use_densifier=False, bg_mode='gray')
    for k in ("extrinsic", "projection", "c2w", "tanfov"):
        put("sd.scene.data." + k, data[k])
    for k in ("image", "depth", "alpha", "image_fg", "radii"):
        put("sd.scene.none." + k, o_none[k])
    OUT["sd.scene.none.keys"] = np.array(sorted(o_none.keys()))
    for k in ("image", "image_fg", "image_bg", "alpha"):
        put("sd.scene.white." + k, o_white[k])
    OUT["sd.scene.white.keys"] = np.array(sorted(o_white.keys()))
    for k in ("image", "image_bg", "depth"):
        put("sd.scene.transl_scale_gray." + k, o_ts[k])

    # ---- sd.sds.*  BasicStableDiffusion.calc_gradients / __call__ (basic.py:546-663, 778-917) on a hand-made self whose
    # `_predict`, `encode_images`, scheduler are the CPU oracle's reduced-width SD graph
    import core.guidance.basic as bs
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import sd15 as nsd
    ucfg = nsd.UNetConfig(block_out_channels=(32, 64), layers_per_block=1, heads=2, cross_dim=32, groups=8, attn_blocks=(True, False),
                          cond_channels=(8, 16))
    vcfg = nsd.VAEConfig(block_out_channels=(16, 32), layers_per_block=1, groups=8)
    usd = nsd.random_state_dict(nsd.unet_param_shapes(ucfg), seed=0)
    csd = nsd.random_state_dict(nsd.controlnet_param_shapes(ucfg), seed=1)
    vsd = nsd.random_state_dict(nsd.vae_encoder_param_shapes(vcfg), seed=2)
    hw = 32
    sdm = object.__new__(bs.BasicScoreDistillation)
    gcfg = cfg.guide
    sdm.cfg = gcfg
    sdm.loss_type, sdm.weight_type = gcfg.sds_loss_type, gcfg.sds_weight_type
    sdm.initial_guidance_scale = gcfg.guidance_scale; sdm.guidance_adjust = gcfg.guidance_adjust
    sdm.do_classifier_free_guidance = True; sdm.use_negative_text = gcfg.use_negative_text
    sdm.input_interpolate = False
    sdm.guidance_scale = gcfg.guidance_scale
    sdm.alphas_cumprod = osd.sd15_alphas_cumprod()
    sdm.scheduler = types.SimpleNamespace(scale_model_input=lambda x, t: x)
    tstep = torch.tensor([417])
    sdm.timestep = tstep
    cond = torch.rand(1, 3, hw, hw, generator=g)
    sdm._predict = lambda lat, text, cond_inputs=None: osd.predict_noise(ucfg, usd, csd, lat, sdm.timestep, text, cond_inputs)
    text = {k: torch.randn(1, 7, 32, generator=g) for k in ("null", "text", "neg")}
    lat_noisy = torch.randn(1, 4, hw // 2, hw // 2, generator=g); noise = torch.randn(1, 4, hw // 2, hw // 2, generator=g)
    with torch.no_grad():
        grads, npred, temb = sdm.calc_gradients(lat_noisy, text, noise, cond_inputs=cond)
    put("sd.sds.cond", cond); put("sd.sds.latents_noisy", lat_noisy); put("sd.sds.noise", noise); put("sd.sds.timestep", tstep)
    for k, v in text.items():
        put("sd.sds.text." + k, v)
    put("sd.sds.gradients", grads); put("sd.sds.noise_pred", npred); put("sd.sds.text_embeddings", temb)
    put("sd.sds.cfg", np.array([32, 64, 1, 2, 32, 8, 8, 16, 16, 32, 1, 8], dtype=np.int64))
    put("sd.sds.guidance", np.array([gcfg.guidance_scale, gcfg.min_timestep, gcfg.max_timestep]))
    OUT["sd.sds.types"] = np.array([gcfg.sds_loss_type, gcfg.sds_weight_type, str(gcfg.use_negative_text), gcfg.time_sampling])
    # SpecifyGradient through `__call__` (noise draw, add_noise, calc_gradients, loss are the reference's)
    img = torch.rand(1, 3, hw, hw, generator=g).requires_grad_(True)
    vnoise = torch.randn(1, 4, hw // 2, hw // 2, generator=g)

    # the reference's own preprocess() runs (guidance scale 'constant', timestep passed as a keyword); only prepare_latents'
    # VAE call is the oracle's (its 512/768-pixel assertion would reject the reduced test size)
    sdm.prepare_latents = lambda inputs: (osd.sample_latents(osd.vae_encode_moments(vcfg, vsd, inputs), vnoise, vcfg.scaling_factor), inputs)
    a_t = sdm.alphas_cumprod[tstep].reshape(-1, 1, 1, 1)
    sdm.add_noise = lambda x, n, t: a_t.sqrt() * x + (1 - a_t).sqrt() * n
    torch.manual_seed(77)
    res = sdm(img, text, train_step=10, max_iteration=100, cond_inputs=cond, timestep=tstep)
    (res['diffusion_loss'] * 1.0).backward()
    torch.manual_seed(77)
    put("sd.sds.call.noise", torch.randn_like(res['latents']))
    put("sd.sds.call.image", img); put("sd.sds.call.vae_noise", vnoise)
    for k in ("latents", "gradients", "sources", "targets", "diffusion_loss"):
        put("sd.sds.call." + k, res[k])
    OUT["sd.sds.call.keys"] = np.array(sorted(res.keys()))
    put("sd.sds.call.image_grad", img.grad)

    np.savez_compressed(os.path.join(HERE, "reference_golden_r2.npz"), **OUT)
    print("wrote reference_golden_r2.npz keys:", len(OUT), "bytes:", os.path.getsize(os.path.join(HERE, "reference_golden_r2.npz")))


if __name__ == "__main__":
    main()
