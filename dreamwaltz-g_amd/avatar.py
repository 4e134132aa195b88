"""MI355X-native mirror of the reference's avatar hot path.

  GaussianOutput / merge_gaussians   /root/reference/core/gaussian/gaussian_utils.py:20-68
  GeneralLinearBlendSkinning         /root/reference/core/human/inverse_lbs.py:517-784   (forward only, frozen skeleton)
  MeshBindingGaussianModel           /root/reference/core/system/avatar.py:921-1079
  DreamWaltzG (.animate, .lbs_transform, .non_rigid_transform, get_*_gaussians)  avatar.py:1097-1588 with the default
                                     flags (configs/__init__.py:117-126,194-205)
Heavy arithmetic = HIP kernels through the C-ABI (lbs.hip, gridenc.hip, gemm.hip); the glue between them is thin torch
element-wise code on the same stream.  Parameter names match the reference so its checkpoints map one to one.
"""
from dataclasses import dataclass, fields
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import assemble as asm_ops
from . import lbs as lbs_ops
from . import meshbind as mb_ops
from .gridencoder import GridEncoder
from .mlp import MLP, DeformNetwork


@dataclass
class GaussianOutput:
    positions: Optional[torch.Tensor] = None
    sh_features: Optional[torch.Tensor] = None
    opacities: Optional[torch.Tensor] = None
    quaternions: Optional[torch.Tensor] = None
    scales: Optional[torch.Tensor] = None
    colors: Optional[torch.Tensor] = None
    cov3D: Optional[torch.Tensor] = None
    offsets: Optional[torch.Tensor] = None
    lbs_weights: Optional[torch.Tensor] = None

    def __getitem__(self, key):
        return getattr(self, key)

    def get(self, key, default=None):
        return getattr(self, key, default)

    def keys(self):
        return iter([f.name for f in fields(self)])


def merge_gaussians(*gaussians: GaussianOutput) -> GaussianOutput:
    if len(gaussians) == 1:
        return gaussians[0]
    out = {}
    for f in fields(GaussianOutput):
        parts = [g[f.name] for g in gaussians if torch.is_tensor(g[f.name])]
        out[f.name] = torch.cat(parts, dim=0) if parts else None
    return GaussianOutput(**out)


# ----------------------------------------------------------------------------------------------------------------------
# quaternion helpers (pytorch3d.transforms restatements used outside the fused LBS kernel: mesh-bound frames only)
# ----------------------------------------------------------------------------------------------------------------------
def _sqrt_positive_part(x):
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=1e-38)), torch.zeros_like(x))


def matrix_to_quaternion(matrix):
    m = matrix.reshape(matrix.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                                             1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = quat_by_rijk / (2.0 * q_abs[..., None].clamp_min(0.1))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(idx.shape + (1, 4))).squeeze(-2)


def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


# ----------------------------------------------------------------------------------------------------------------------
# skeleton
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class LBSTransforms:
    """What DreamWaltzG needs from lbs_model.forward(): `A` = compose(J_pose_rigid, G_transl_offset) [J,4,4]."""
    A: torch.Tensor
    rot_mats: torch.Tensor
    full_shape: torch.Tensor


class GeneralLinearBlendSkinning(nn.Module):
    """Forward-only mirror of inverse_lbs.py:517-784 for a frozen SMPL-X(-shaped) model given as tensors."""

    def __init__(self, body: Dict[str, torch.Tensor]):
        super().__init__()
        for k in ("v_template", "posedirs", "J_regressor", "lbs_weights", "betas", "expression", "pose_mean", "jaw_pose",
                  "leye_pose", "reye_pose"):
            self.register_buffer(k, body[k].float().contiguous())
        shapedirs = torch.cat([body["shapedirs"], body["expr_dirs"]], dim=-1).float().contiguous()   # [V,3,400]
        self.register_buffer("shapedirs_all", shapedirs)
        self.register_buffer("parents", torch.as_tensor(body["parents"]).to(torch.int32))
        self.register_buffer("J_template", torch.einsum('ik,ji->jk', self.v_template, self.J_regressor))
        # joint_shape_dirs[j,c,l] = sum_v J_regressor[j,v] shapedirs[v,c,l]: vertices2joints of the blend-shape offsets
        self.register_buffer("joint_shape_dirs", torch.einsum('jv,vcl->jcl', self.J_regressor, shapedirs).contiguous())
        self.num_joints = self.J_regressor.shape[0]
        self.NUM_BODY_JOINTS = 21
        self._subsets = {}

    def get_full_shape(self, betas=None, expression=None, extra_betas=None):
        betas = self.betas if betas is None else betas
        if extra_betas is not None:
            betas = betas + extra_betas
        expression = self.expression if expression is None else expression
        return torch.cat([betas, expression], dim=-1)

    def get_full_pose(self, body_pose=None, global_orient=None, left_hand_pose=None, right_hand_pose=None, jaw_pose=None,
                      leye_pose=None, reye_pose=None):
        """inverse_lbs.py:591-631 -- jaw/eye arguments are accepted and ignored exactly like the reference (checklist Q1)."""
        dev, z = self.v_template.device, (lambda n: torch.zeros(1, n, device=self.v_template.device))
        global_orient = z(3) if global_orient is None else global_orient
        body_pose = z(63) if body_pose is None else body_pose
        left_hand_pose = z(45) if left_hand_pose is None else left_hand_pose
        right_hand_pose = z(45) if right_hand_pose is None else right_hand_pose
        full = torch.cat([global_orient.reshape(-1, 3), body_pose.reshape(-1, 3), self.jaw_pose.reshape(-1, 3),
                          self.leye_pose.reshape(-1, 3), self.reye_pose.reshape(-1, 3), left_hand_pose.reshape(-1, 3),
                          right_hand_pose.reshape(-1, 3)], dim=0).to(dev)
        return full + self.pose_mean.reshape(-1, 3)

    @torch.no_grad()
    def forward(self, betas=None, body_pose=None, global_orient=None, left_hand_pose=None, right_hand_pose=None,
                jaw_pose=None, leye_pose=None, reye_pose=None, expression=None, transl=None, extra_betas=None, **_unused):
        full_shape = self.get_full_shape(betas=betas, expression=expression, extra_betas=extra_betas)
        full_pose = self.get_full_pose(body_pose, global_orient, left_hand_pose, right_hand_pose, jaw_pose, leye_pose, reye_pose)
        A, R = lbs_ops.joint_chain(full_pose, self.J_template, self.parents, transl=transl, return_rot_mats=True,
                                   joint_shape_dirs=self.joint_shape_dirs, shape_coeffs=full_shape)
        return LBSTransforms(A=A, rot_mats=R, full_shape=full_shape)

    @torch.no_grad()
    def transform_vertices(self, tr: LBSTransforms, vertex_indices, vertex_coords):
        """transform_V.transform_points(vertex_coords, indices=...) (avatar.py:1570,1577).  The subset's blend-shape rows are
        gathered once per distinct index tensor."""
        key = (vertex_indices.data_ptr(), int(vertex_indices.numel()))
        if key not in self._subsets:
            self._subsets[key] = lbs_ops.gather_vertex_subset(vertex_indices, self.lbs_weights, self.shapedirs_all, self.posedirs)
        return lbs_ops.vertex_transform(vertex_coords, tr.A, self._subsets[key], tr.full_shape, tr.rot_mats)


# ----------------------------------------------------------------------------------------------------------------------
# mesh-bound Gaussians (hands / face)
# ----------------------------------------------------------------------------------------------------------------------
def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def compute_normal(vertices, faces):
    """utils/mesh.py:34-94 (single mesh)."""
    i0, i1, i2 = faces[:, 0], faces[:, 1], faces[:, 2]
    v0, v1, v2 = vertices[i0], vertices[i1], vertices[i2]
    fn = safe_normalize(torch.linalg.cross(v1 - v0, v2 - v0))
    vn = torch.zeros_like(vertices).index_add(0, i0, fn).index_add(0, i1, fn).index_add(0, i2, fn)
    vn = torch.where((vn * vn).sum(-1, keepdim=True) > 1e-20, vn, torch.tensor([0.0, 0.0, 1.0], device=vertices.device))
    return safe_normalize(vn), fn


class MeshBindingGaussianModel(nn.Module):
    """avatar.py:921-1079: n Gaussians per triangle, learnable barycentric coordinates and tangent scales."""

    def __init__(self, vertex_coords, triangles, vertex_indices, n_per_triangle=6, init_scale_ratio=1.0):
        super().__init__()
        self.register_buffer("predefined_vertex_indices", vertex_indices.long())
        self.register_buffer("triangles", triangles.long())
        self._n_points_per_triangle = n_per_triangle
        Fp = triangles.shape[0]
        base = torch.tensor([[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3], [1 / 6, 5 / 12, 5 / 12],
                             [5 / 12, 1 / 6, 5 / 12], [5 / 12, 5 / 12, 1 / 6]], dtype=torch.float32)
        assert n_per_triangle == 6, "default n_gaussians_per_triangle"
        self._bary_coords = nn.Parameter(base.expand(Fp, -1, -1).clone())
        self._vertex_coords = nn.Parameter(vertex_coords.float().clone(), requires_grad=False)
        self._scales = nn.Parameter(torch.ones(Fp * n_per_triangle, 3) * init_scale_ratio)
        p2t = torch.arange(Fp)[:, None].expand(-1, n_per_triangle).reshape(-1)
        self.register_buffer("points_to_vertices", self.triangles[p2t])
        # native path (csrc/meshbind.hip): int32 topology + the static vertex -> face adjacency for the normal gather
        self.register_buffer("triangles_i32", self.triangles.to(torch.int32).contiguous())
        off, faces = mb_ops.build_vertex_face_csr(self.triangles, vertex_coords.shape[0])
        self.register_buffer("vf_offsets", off)
        self.register_buffer("vf_faces", faces)

    def forward(self, canonical_vertex_coords, observed_vertex_coords):
        """get_positions (canonical + observed) and get_scales_and_quaternions (observed) in one forward / one backward
        launch -> (canonical positions, positions, scales, quaternions).  The torch-op methods below are the same math,
        kept as the reference-named entry points."""
        vn = mb_ops.vertex_normals(observed_vertex_coords, self.triangles_i32, self.vf_offsets, self.vf_faces)
        return mb_ops.meshbind(self._bary_coords, self._scales, canonical_vertex_coords, observed_vertex_coords, vn,
                               self.triangles_i32, self._n_points_per_triangle)

    def get_positions(self, vertex_coords):
        bary = self._bary_coords / self._bary_coords.sum(dim=-1, keepdim=True)
        return torch.einsum('fnv,fvc->fnc', bary, vertex_coords[self.triangles]).reshape(-1, 3)

    def get_scales_and_quaternions(self, vertex_coords, positions, eps=1e-9):
        dot = lambda a, b: (a * b).sum(-1, keepdim=True)  # noqa: E731
        p0 = positions
        pv = vertex_coords[self.points_to_vertices]
        p1, p2, p3 = pv[:, 0], pv[:, 1], pv[:, 2]
        vn, _ = compute_normal(vertex_coords, self.triangles)
        pn = (vn[self.points_to_vertices] * self._bary_coords.reshape(-1, 3)[:, :, None]).sum(dim=1)   # raw bary (Q5)
        v0 = pn / (torch.linalg.vector_norm(pn, dim=-1, keepdim=True) + eps)
        ref = torch.tensor((1.0, 0.0, 0.0), device=p0.device).expand_as(p0)
        v1 = torch.linalg.cross(v0, ref)
        v1 = v1 / (torch.linalg.vector_norm(v1, dim=-1, keepdim=True) + eps)
        v2 = torch.linalg.cross(v0, v1)
        v2 = v2 / (torch.linalg.vector_norm(v2, dim=-1, keepdim=True) + eps)
        R = torch.stack((v0, v1, v2), dim=2) * torch.tensor([1.0, -1.0, -1.0], device=p0.device)[None, :, None]
        n = self._n_points_per_triangle
        s0 = torch.zeros_like(v0[:, :1])
        s1 = (dot(p1 - p0, v1).abs() + dot(p2 - p0, v1).abs() + dot(p3 - p0, v1).abs()) / n
        s2 = (dot(p1 - p0, v2).abs() + dot(p2 - p0, v2).abs() + dot(p3 - p0, v2).abs()) / n
        s1 = s1 * torch.clamp(self._scales[:, 1:2], min=0.5, max=2.0)
        s2 = s2 * torch.clamp(self._scales[:, 2:3], min=0.5, max=2.0)
        return torch.cat((s0, s1, s2), dim=1), standardize_quaternion(matrix_to_quaternion(R))


# ----------------------------------------------------------------------------------------------------------------------
# the avatar
# ----------------------------------------------------------------------------------------------------------------------
class DreamWaltzG(nn.Module):
    def __init__(self, lbs_model: GeneralLinearBlendSkinning, positions, scales, quaternions, lbs_weights,
                 smpl_canonical_inputs: dict, mesh_binding_gaussians: Optional[Dict[str, MeshBindingGaussianModel]] = None,
                 nerf_bound=2.0, init_offset=0.01, init_scale=0.001):
        super().__init__()
        self.lbs_model = lbs_model
        self._positions = nn.Parameter(positions.float().clone())
        self._scales = nn.Parameter(torch.log(scales.float().clone()))       # scale_activation = exp
        self._quaternions = nn.Parameter(quaternions.float().clone())
        self._lbs_weights = nn.Parameter(lbs_weights.float().clone(), requires_grad=False)   # configs/__init__.py:197
        self.smpl_canonical_inputs = smpl_canonical_inputs
        self.nerf_bound, self.init_offset, self.init_scale = nerf_bound, init_offset, init_scale
        # nerf_model.py:223-232: tiledgrid encoder L=16 C=2 base 16 -> 2048*bound, smoothstep; sigma_net 32->64->64->4
        self.nerf_encoder = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                        desired_resolution=2048 * nerf_bound, gridtype='tiled', align_corners=False,
                                        interpolation='smoothstep')
        self.nerf_opacity_and_color_net = MLP(32, 4, 64, 3, bias=True)
        self.nerf_scale_and_quaternion_net = DeformNetwork(xyz_input_ch=32, D=4, W=64)
        self.mesh_binding_gaussians = nn.ModuleDict(mesh_binding_gaussians or {})
        self._canonical_cache = None
        self._canonical_vertices = {}

    # -- checkpoints ------------------------------------------------------------------------------------------------
    def load_reference_state_dict(self, state_dict, prefix="avatar."):
        """Loads the avatar part of a reference checkpoint's `model` entry (trainer.py:238-259 saves {'train_step',
        'checkpoints', 'model': Scene.state_dict()}; the avatar sits under `avatar.` and, duplicated, `avatars.0.`).
        Per-Gaussian parameters are first resized to the checkpoint's Gaussian count exactly like
        GaussianModel.reset_by_state_dict (gaussian_model.py:58-85, avatar.py:1254-1281); parameter names match the reference,
        so everything else is a plain copy.  Returns (loaded keys, checkpoint keys with no counterpart here, our keys the
        checkpoint does not carry).  Call it BEFORE the parameters are re-homed into a FlatAdam buffer when the count changes."""
        sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        if "_positions" in sd:
            n = sd["_positions"].shape[0]
            for name in ("_positions", "_scales", "_quaternions", "_lbs_weights"):
                cur = getattr(self, name, None)
                if cur is not None and name in sd and cur.shape[0] != n:
                    new = torch.empty(n, *cur.shape[1:], dtype=cur.dtype, device=cur.device)
                    setattr(self, name, nn.Parameter(new, requires_grad=cur.requires_grad))
        own = self.state_dict()
        loaded, unknown = [], []
        for k, v in sd.items():
            if k in own and tuple(own[k].shape) == tuple(v.shape):
                own[k].copy_(v.to(own[k].dtype))
                loaded.append(k)
            else:
                unknown.append(k)
        missing = [k for k in own if k not in sd]
        self._canonical_cache = None
        self._canonical_vertices = {}
        return loaded, unknown, missing

    # -- avatar.py:913-918
    def get_lbs_weights(self):
        return self._lbs_weights

    def lbs_transform(self, positions, transforms: LBSTransforms, quaternions=None):
        """avatar.py:1426-1462 with use_*_offsets False; the weight normalisation of get_lbs_weights is fused in."""
        return lbs_ops.lbs_blend(transforms.A, self._lbs_weights, positions, quaternions, normalize_weights=True)

    def static_mlp_forward(self, enc, fix_opacities=False):
        oc = self.nerf_opacity_and_color_net(enc)
        colors = torch.sigmoid(oc[:, 1:])
        opacities = torch.ones_like(oc[:, :1]) if fix_opacities else torch.sigmoid(oc[:, :1])
        return colors, opacities

    def animate(self, smpl_observed_inputs: Optional[dict] = None) -> GaussianOutput:
        if smpl_observed_inputs is None:
            smpl_observed_inputs = self.smpl_canonical_inputs
        if self._canonical_cache is None:           # canonical inputs never change: cache the skeleton pass
            self._canonical_cache = self.lbs_model.forward(**self.smpl_canonical_inputs)
        ctr = self._canonical_cache
        otr = self.lbs_model.forward(**smpl_observed_inputs)
        positions = self._positions
        canonical_positions = self.lbs_transform(positions, ctr)
        N = positions.shape[0]
        # mesh-bound parts first: their canonical positions go through the SAME encoder / static-MLP launches as the free
        # Gaussians (the reference calls the two networks once per part, avatar.py:1544-1583; the maths is row-wise, so one pass
        # over the concatenated rows gives the same values with half the launches and one table-gradient scatter)
        mesh_parts = []
        for _name, gm in self.mesh_binding_gaussians.items():
            vc = gm._vertex_coords
            cvc = self._canonical_vertices.get(_name)
            if cvc is None:                          # canonical pose and the bound vertices are fixed: transform once
                cvc = self._canonical_vertices[_name] = self.lbs_model.transform_vertices(ctr, gm.predefined_vertex_indices, vc)
            ovc = self.lbs_model.transform_vertices(otr, gm.predefined_vertex_indices, vc)
            mesh_parts.append(gm(cvc, ovc))          # (cpos, pos_m, sc_m, q_m): one HIP launch each way (csrc/meshbind.hip)
        all_cpos = torch.cat([canonical_positions] + [mp[0] for mp in mesh_parts], dim=0) if mesh_parts else canonical_positions
        enc_all = self.nerf_encoder(all_cpos, bound=self.nerf_bound)
        oc_all = self.nerf_opacity_and_color_net(enc_all)                      # static_mlp_forward (avatar.py:1283-1290), all rows
        enc = enc_all[:N]
        body_pose = smpl_observed_inputs.get('body_pose')
        if body_pose is None:
            body_pose = torch.zeros(1, 63, device=positions.device)
        offsets, mlp_scales, _mlp_quats = self.nerf_scale_and_quaternion_net(enc, body_pose)
        # non_rigid_transform (avatar.py:1464-1498, default flags) + the sigmoid / exp / normalize activations: one HIP launch
        pos, scales, quats, col_all, op_all = asm_ops.assemble(positions, offsets, self._scales, mlp_scales, self._quaternions, oc_all,
                                                               self.init_offset, self.init_scale)
        pos, quats = self.lbs_transform(pos, otr, quats)
        if not mesh_parts:
            return GaussianOutput(positions=pos, opacities=op_all, colors=col_all, quaternions=quats, scales=scales)
        # merge_gaussians (gaussian_utils.py:56-68): colours / opacities already come out in the merged row order
        return GaussianOutput(positions=torch.cat([pos] + [mp[1] for mp in mesh_parts], dim=0), opacities=op_all, colors=col_all,
                              quaternions=torch.cat([quats] + [mp[3] for mp in mesh_parts], dim=0),
                              scales=torch.cat([scales] + [mp[2] for mp in mesh_parts], dim=0))
