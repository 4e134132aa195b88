"""oracle/animate.py -- CPU (PyTorch, fp32/fp64) restatement of the reference's LBS / attribute stage.
TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/, smoke() and bench.py's cpu_baseline leg import it).

Every function cites the reference lines it follows (paths relative to /root/reference):
  RigidTransform algebra           core/human/inverse_lbs.py:15-260
  GeneralLinearBlendSkinning       core/human/inverse_lbs.py:517-784
  DreamWaltzG.animate & friends    core/system/avatar.py:1283-1588
  MeshBindingGaussianModel         core/system/avatar.py:1016-1079, utils/mesh.py:34-94
  GridEncoder                      core/nerf/gridencoder/grid.py:99-165, src/gridencoder.cu:49-242
  MLP / DeformNetwork              core/nerf/nerf_model.py:12-33, core/deformation/deform_model.py:61-143

PARITY UNPINNED at the third-party boundary: `smplx.lbs.{blend_shapes,vertices2joints,batch_rodrigues,
batch_rigid_transform}` (smplx git HEAD, scripts/install.sh:24) and `pytorch3d.transforms.{quaternion_to_matrix,
matrix_to_quaternion,standardize_quaternion,quaternion_multiply}` (pytorch3d 0.7.5, install.sh:8) are not installed
here; the functions in section "third-party restatements" follow their published algorithms from memory.
The grid encoder cannot be executed here either (CUDA JIT); it is restated from gridencoder.cu.
What IS pinned by golden fixtures (tests/golden): DeformNetwork, eval_sh, LR schedule, and the in-repo RigidTransform /
get_full_transform algebra driven through these stand-ins (tests/golden/capture_golden.py).
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------------------------
# third-party restatements [3P-memory]
# ----------------------------------------------------------------------------------------------------------------


def blend_shapes(betas, shape_disps):
    return torch.einsum('bl,mkl->bmk', betas, shape_disps)


def vertices2joints(J_regressor, vertices):
    return torch.einsum('bik,ji->bjk', vertices, J_regressor)


def batch_rodrigues(rot_vecs):
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """rot_mats [B,J,3,3], joints [B,J,3], parents [J] (parents[0] = -1) -> posed_joints [B,J,3], A [B,J,4,4]."""
    B, J = joints.shape[:2]
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] = rel[:, 1:] - joints[:, parents[1:]]
    tm = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                    F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1.0)], dim=2).reshape(B, J, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed = transforms[:, :, :3, 3]
    jh = F.pad(joints, [0, 0, 0, 1])
    rel_t = transforms - F.pad(torch.matmul(transforms, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, rel_t


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def matrix_to_quaternion(matrix):
    """pytorch3d 0.7.5 form: best-conditioned of four candidates, floor 0.1, NOT standardised."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1, dtype=q_abs.dtype)
    cand = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    sel = F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return cand[sel, :].reshape(batch_dim + (4,))


def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a, b):
    return standardize_quaternion(quaternion_raw_multiply(a, b))


# ----------------------------------------------------------------------------------------------------------------
# RigidTransform algebra (inverse_lbs.py:139-251), as plain functions on SE3 tensors
# ----------------------------------------------------------------------------------------------------------------
def se3_from_T(T):
    SE3 = torch.eye(4, dtype=T.dtype).expand(*T.shape[:-1], 4, 4).contiguous()
    SE3[..., :3, 3] = T
    return SE3


def se3_compose(first, *others):
    """inverse_lbs.py:145-159: SE3 = other.SE3 @ SE3, in order."""
    SE3 = first.clone()
    for o in others:
        SE3 = o @ SE3
    return SE3


def transform_points(SE3, points, indices=None, weights=None):
    """inverse_lbs.py:190-210."""
    R, T = SE3[..., :3, :3], SE3[..., :3, 3]
    if indices is not None:
        R, T = R[indices], T[indices]
    if weights is not None:
        R = torch.einsum('nj,jkl->nkl', weights, R)
        T = torch.einsum('nj,jk->nk', weights, T)
    return torch.matmul(R, points.unsqueeze(-1))[..., :, 0] + T


def transform_quaternions_flip(SE3, quaternions, weights):
    """inverse_lbs.py:234-242 (flip_rotation_axis=True; blended rotation NOT re-orthonormalised, checklist Q2/Q3)."""
    R = torch.einsum('nj,jkl->nkl', weights, SE3[..., :3, :3])
    rot = quaternion_to_matrix(quaternions)
    flip = torch.tensor([1.0, -1.0, -1.0], dtype=rot.dtype)[None, :, None]
    rot = rot * flip
    rot = R @ rot
    rot = rot * flip
    return matrix_to_quaternion(rot)


def transform_quaternions(SE3, quaternions, indices=None, weights=None, rotation_mode='quaternion', flip_rotation_axis=False):
    """inverse_lbs.py:212-251, all three branches."""
    R = SE3[..., :3, :3]
    if indices is not None:
        R = R[indices]
    if weights is not None:
        R = torch.einsum('nj,jkl->nkl', weights, SE3[..., :3, :3])
    if flip_rotation_axis:
        flip = torch.tensor([1.0, -1.0, -1.0], dtype=R.dtype)[None, :, None]
        return matrix_to_quaternion((R @ (quaternion_to_matrix(quaternions) * flip)) * flip)
    if rotation_mode == 'matrix':
        return matrix_to_quaternion(R @ quaternion_to_matrix(quaternions))
    return quaternion_multiply(matrix_to_quaternion(R), quaternions)


def se3_inverse(SE3):
    """inverse_lbs.py:102-138: [R^T, -R^T t]; the last row of the INPUT is overwritten with (0,0,0,1) in place (checklist Q7)."""
    SE3[..., 3, :] = torch.tensor([0, 0, 0, 1], dtype=SE3.dtype)
    Rt = SE3[..., :3, :3].transpose(-1, -2)
    out = torch.zeros_like(SE3)
    out[..., :3, :3] = Rt
    out[..., :3, 3] = -torch.matmul(Rt, SE3[..., :3, 3].unsqueeze(-1)).squeeze(-1)
    out[..., 3, 3] = 1.0
    return out


def se3_weight(SE3, weights):
    """inverse_lbs.py:174-180 (qr_correct=False): blends the whole 4x4."""
    return torch.einsum('nj,jkl->nkl', weights, SE3)


def inverse_transform_points(points, R, T):
    """inverse_lbs.py:187-188: general 3x3 inverse (checklist Q8)."""
    return torch.matmul(torch.inverse(R), (points - T).unsqueeze(-1))[..., :, 0]


def inverse_lbs_transform(positions, transforms, lbs_weights):
    """avatar.py:1377-1424 with use_*_offsets False: inverse of the BLENDED (non-rigid) matrix per point."""
    jt = se3_weight(se3_compose(transforms["J_pose_rigid"], transforms["G_transl_offset"])[0], lbs_weights)
    return inverse_transform_points(positions, jt[..., :3, :3], jt[..., :3, 3])


# ----------------------------------------------------------------------------------------------------------------
# synthetic SMPL-X-shaped body model (the licensed model file is absent)
# ----------------------------------------------------------------------------------------------------------------
class SyntheticBody:
    """Same tensor roles/shapes as smplx.SMPLX (V vertices, J=55 joints, 300+100 shape comps, 486 pose features)."""

    def __init__(self, V=10475, F_=20908, J=55, n_betas=300, n_expr=100, seed=0, dtype=torch.float32):
        g = torch.Generator().manual_seed(seed)
        box = torch.tensor([0.4, 0.9, 0.2])
        self.V, self.J = V, J
        self.v_template = ((torch.rand(V, 3, generator=g) * 2 - 1) * box).to(dtype)
        self.shapedirs = (torch.randn(V, 3, n_betas, generator=g) * 1e-3).to(dtype)
        self.expr_dirs = (torch.randn(V, 3, n_expr, generator=g) * 1e-3).to(dtype)
        self.posedirs = (torch.randn((J - 1) * 9, V * 3, generator=g) * 1e-3).to(dtype)
        # sparse-ish regressor and skinning weights (4 non-zeros per row)
        Jr = torch.zeros(J, V)
        for j in range(J):
            idx = torch.randint(0, V, (16,), generator=g)
            Jr[j, idx] = 1.0 / 16
        self.J_regressor = Jr.to(dtype)
        logits = torch.full((V, J), -1e9)
        cols = torch.randint(0, J, (V, 4), generator=g)
        logits.scatter_(1, cols, torch.randn(V, 4, generator=g))
        self.lbs_weights = torch.softmax(logits, dim=1).to(dtype)
        parents = [-1] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, J)]
        self.parents = np.array(parents, dtype=np.int64)
        self.faces = torch.randint(0, V, (F_, 3), generator=g)
        self.betas = torch.zeros(1, n_betas, dtype=dtype)
        self.expression = torch.zeros(1, n_expr, dtype=dtype)
        self.pose_mean = torch.zeros(J * 3, dtype=dtype)
        self.jaw_pose = torch.zeros(1, 3, dtype=dtype)
        self.leye_pose = torch.zeros(1, 3, dtype=dtype)
        self.reye_pose = torch.zeros(1, 3, dtype=dtype)
        self.J_template = torch.einsum('ik,ji->jk', self.v_template, self.J_regressor)
        self.NUM_BODY_JOINTS = 21

    def to(self, dtype):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v) and v.is_floating_point():
                setattr(self, k, v.to(dtype))
        return self


def random_smpl_inputs(seed=0, pose_std=0.3, dtype=torch.float32, transl=True):
    g = torch.Generator().manual_seed(seed)
    d = dict(body_pose=torch.randn(1, 63, generator=g) * pose_std, global_orient=torch.randn(1, 3, generator=g) * pose_std,
             left_hand_pose=torch.randn(1, 45, generator=g) * pose_std, right_hand_pose=torch.randn(1, 45, generator=g) * pose_std,
             expression=torch.randn(1, 100, generator=g) * 0.5)
    if transl:
        d["transl"] = torch.randn(1, 3, generator=g) * 0.05
    return {k: v.to(dtype) for k, v in d.items()}


# ----------------------------------------------------------------------------------------------------------------
# GeneralLinearBlendSkinning (inverse_lbs.py:570-784)
# ----------------------------------------------------------------------------------------------------------------
def glbs_full_shape(body, betas=None, expression=None, extra_betas=None):
    betas = body.betas if betas is None else betas
    if extra_betas is not None:
        betas = betas + extra_betas
    expression = body.expression if expression is None else expression
    return torch.cat([betas, expression], dim=-1)


def glbs_full_pose(body, body_pose=None, global_orient=None, left_hand_pose=None, right_hand_pose=None, **_ignored):
    """inverse_lbs.py:591-631 -- jaw/eye poses passed by the caller are IGNORED (checklist Q1)."""
    dt = body.v_template.dtype
    z = lambda n: torch.zeros(1, n, dtype=dt)  # noqa: E731
    global_orient = z(3) if global_orient is None else global_orient
    body_pose = z(63) if body_pose is None else body_pose
    left_hand_pose = z(45) if left_hand_pose is None else left_hand_pose
    right_hand_pose = z(45) if right_hand_pose is None else right_hand_pose
    full = torch.cat([global_orient.reshape(-1, 1, 3), body_pose.reshape(-1, body.NUM_BODY_JOINTS, 3),
                      body.jaw_pose.reshape(-1, 1, 3), body.leye_pose.reshape(-1, 1, 3), body.reye_pose.reshape(-1, 1, 3),
                      left_hand_pose.reshape(-1, 15, 3), right_hand_pose.reshape(-1, 15, 3)], dim=1).reshape(-1, 165)
    return full + body.pose_mean


def glbs_full_transform(body, betas, pose):
    """inverse_lbs.py:652-717."""
    dt = betas.dtype
    shapedirs = torch.cat([body.shapedirs, body.expr_dirs], dim=-1)
    shape_offsets = blend_shapes(betas, shapedirs)
    v_shaped = body.v_template + shape_offsets
    Jp = vertices2joints(body.J_regressor, v_shaped)
    rot_mats = batch_rodrigues(pose.view(-1, 3)).view(1, -1, 3, 3)
    pose_feature = rot_mats[:, 1:, :, :] - torch.eye(3, dtype=dt)
    pose_offsets = torch.matmul(pose_feature.view(1, -1), body.posedirs).view(1, -1, 3)
    _, A = batch_rigid_transform(rot_mats, Jp, body.parents)
    T = torch.matmul(body.lbs_weights.unsqueeze(0), A.view(1, body.J, 16)).view(1, -1, 4, 4)
    return {"V_shape_offset": se3_from_T(shape_offsets), "V_pose_offset": se3_from_T(pose_offsets), "V_pose_rigid": T,
            "J_shape_offset": se3_from_T(Jp - body.J_template), "J_pose_rigid": A}


def glbs_forward(body, transl=None, extra_betas=None, betas=None, expression=None, **pose_kw):
    """inverse_lbs.py:719-784 -> (transform_J, transform_V, transforms) as SE3 tensors."""
    full_shape = glbs_full_shape(body, betas=betas, expression=expression, extra_betas=extra_betas)
    full_pose = glbs_full_pose(body, **pose_kw)
    tr = glbs_full_transform(body, full_shape, full_pose)
    tV = se3_compose(tr["V_shape_offset"], tr["V_pose_offset"], tr["V_pose_rigid"])
    tJ = se3_compose(tr["J_shape_offset"], tr["J_pose_rigid"])
    if transl is not None:
        tt = se3_from_T(transl)
        tV = se3_compose(tV, tt)
        tJ = se3_compose(tJ, tt)
        tr["G_transl_offset"] = tt
    else:
        tr["G_transl_offset"] = torch.eye(4, dtype=full_shape.dtype).expand(1, 4, 4)
    return tJ, tV, tr


# ----------------------------------------------------------------------------------------------------------------
# grid encoder (grid.py:119-134 table sizing; gridencoder.cu:66-84 index, :87-242 forward)
# ----------------------------------------------------------------------------------------------------------------
PRIMES = (1, 2654435761, 805459861)


def grid_offsets(input_dim=3, num_levels=16, base_resolution=16, desired_resolution=4096, log2_hashmap_size=19,
                 align_corners=False):
    per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset, max_params = [], 0, 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def grid_encode(x01, table, offsets, per_level_scale, base_resolution=16, gridtype=1, align_corners=False, interp=1):
    """x01 [B,3] in [0,1]; table [sO,C]; returns [B, L*C].  Differentiable w.r.t. x01 and table through autograd
    (the analytic dy_dx of the CUDA kernel equals d(out)/d(x01) of this expression where the floor is constant)."""
    B, D = x01.shape
    L = len(offsets) - 1
    C = table.shape[1]
    S = np.float32(np.log2(per_level_scale))
    oob = ((x01 < 0) | (x01 > 1)).any(dim=1)
    outs = []
    for l in range(L):
        hashmap_size = int(offsets[l + 1] - offsets[l])
        scale = np.float32(np.exp2(np.float32(l) * S, dtype=np.float32) * np.float32(base_resolution) - np.float32(1.0))
        resolution = int(math.ceil(float(scale))) + 1
        pos = x01 * float(scale) + (0.0 if align_corners else 0.5)
        pg = torch.floor(pos.detach())
        frac = pos - pg
        pgi = pg.to(torch.int64)
        w1 = frac * frac * (3.0 - 2.0 * frac) if interp == 1 else frac
        res = torch.zeros(B, C, dtype=table.dtype)
        for idx in range(1 << D):
            w = torch.ones(B, dtype=x01.dtype)
            index = torch.zeros(B, dtype=torch.int64)
            stride = 1
            hashed = torch.zeros(B, dtype=torch.int64)
            cells = []
            for d in range(D):
                bit = (idx >> d) & 1
                w = w * (w1[:, d] if bit else (1 - w1[:, d]))
                cells.append(pgi[:, d] + bit)
            used_hash = False
            for d in range(D):
                if stride <= hashmap_size:
                    index = index + cells[d] * stride
                    stride *= resolution if align_corners else (resolution + 1)
            if gridtype == 0 and stride > hashmap_size:
                used_hash = True
                for d in range(D):
                    hashed = hashed ^ ((cells[d] * PRIMES[d]) & 0xFFFFFFFF)
            index = (hashed if used_hash else (index & 0xFFFFFFFF)) % hashmap_size
            res = res + w[:, None] * table[int(offsets[l]) + index]
        res = torch.where(oob[:, None], torch.zeros_like(res), res)
        outs.append(res)
    return torch.stack(outs, dim=1).reshape(B, L * C)


# ----------------------------------------------------------------------------------------------------------------
# MLPs
# ----------------------------------------------------------------------------------------------------------------
def mlp_forward(x, weights, biases):
    """nerf_model.py:28-33: Linear+ReLU ... Linear."""
    n = len(weights)
    for l in range(n):
        x = F.linear(x, weights[l], biases[l])
        if l != n - 1:
            x = F.relu(x)
    return x


def deform_forward(enc, body_pose, p):
    """deform_model.py:102-143 with D=4, W=64, is_6dof=False; p holds layers.{i}.weight/bias and the three heads."""
    h = torch.cat([enc, body_pose.expand(enc.shape[0], -1)], dim=-1)
    for i in range(4):
        h = F.leaky_relu(F.linear(h, p["layers.%d.weight" % i], p["layers.%d.bias" % i]))
    return (F.linear(h, p["gaussian_warp.weight"], p["gaussian_warp.bias"]),
            F.linear(h, p["gaussian_scaling.weight"], p["gaussian_scaling.bias"]),
            F.linear(h, p["gaussian_rotation.weight"], p["gaussian_rotation.bias"]))


def init_linear(out_f, in_f, g, dtype=torch.float32):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
    return w.to(dtype), b.to(dtype)


def init_avatar_networks(seed=0, dtype=torch.float32, table_std=1e-4):
    g = torch.Generator().manual_seed(seed)
    offsets, pls = grid_offsets()
    table = ((torch.rand(int(offsets[-1]), 2, generator=g) * 2 - 1) * table_std).to(dtype)
    static_w, static_b = [], []
    for (o, i) in ((64, 32), (64, 64), (4, 64)):
        w, b = init_linear(o, i, g, dtype); static_w.append(w); static_b.append(b)
    deform = {}
    for li, (o, i) in enumerate(((64, 95), (64, 64), (64, 64), (64, 64))):
        deform["layers.%d.weight" % li], deform["layers.%d.bias" % li] = init_linear(o, i, g, dtype)
    for name, o in (("gaussian_warp", 3), ("gaussian_rotation", 4), ("gaussian_scaling", 3)):
        deform[name + ".weight"], deform[name + ".bias"] = init_linear(o, 64, g, dtype)
    return dict(offsets=offsets, per_level_scale=pls, table=table, static_w=static_w, static_b=static_b, deform=deform)


# ----------------------------------------------------------------------------------------------------------------
# mesh-bound Gaussians (avatar.py:1016-1079, utils/mesh.py:34-94)
# ----------------------------------------------------------------------------------------------------------------
def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def compute_normal(vertices, faces):
    i0, i1, i2 = faces[:, 0], faces[:, 1], faces[:, 2]
    v0, v1, v2 = vertices[i0], vertices[i1], vertices[i2]
    fn = safe_normalize(torch.linalg.cross(v1 - v0, v2 - v0))
    vn = torch.zeros_like(vertices)
    vn = vn.index_add(0, i0, fn).index_add(0, i1, fn).index_add(0, i2, fn)
    deflt = torch.tensor([0.0, 0.0, 1.0], dtype=vertices.dtype)
    vn = torch.where((vn * vn).sum(-1, keepdim=True) > 1e-20, vn, deflt)
    return safe_normalize(vn), fn


def mesh_positions(bary_raw, vertex_coords, triangles):
    bary = bary_raw / bary_raw.sum(dim=-1, keepdim=True)
    tri = vertex_coords[triangles]
    return torch.einsum('fnv,fvc->fnc', bary, tri).reshape(-1, 3)


def mesh_scales_and_quaternions(bary_raw, scales_raw, vertex_coords, triangles, positions, n_per_tri, eps=1e-9):
    Fp = triangles.shape[0]
    p2t = torch.arange(Fp)[:, None].expand(-1, n_per_tri).reshape(-1)
    p2v = triangles[p2t]
    dot = lambda a, b: (a * b).sum(-1, keepdim=True)  # noqa: E731
    p0 = positions
    pv = vertex_coords[p2v]
    p1, p2, p3 = pv[:, 0], pv[:, 1], pv[:, 2]
    vn, _ = compute_normal(vertex_coords, triangles)
    pn = (vn[p2v] * bary_raw.reshape(-1, 3)[:, :, None]).sum(dim=1)   # RAW bary coords (checklist Q5)
    v0 = pn / (torch.linalg.vector_norm(pn, dim=-1, keepdim=True) + eps)
    ref = torch.tensor((1.0, 0.0, 0.0), dtype=p0.dtype).expand_as(p0)
    v1 = torch.linalg.cross(v0, ref)
    v1 = v1 / (torch.linalg.vector_norm(v1, dim=-1, keepdim=True) + eps)
    v2 = torch.linalg.cross(v0, v1)
    v2 = v2 / (torch.linalg.vector_norm(v2, dim=-1, keepdim=True) + eps)
    R = torch.stack((v0, v1, v2), dim=2)
    R = R * torch.tensor([1.0, -1.0, -1.0], dtype=R.dtype)[None, :, None]
    s0 = torch.zeros_like(v0[:, :1])
    s1 = (dot(p1 - p0, v1).abs() + dot(p2 - p0, v1).abs() + dot(p3 - p0, v1).abs()) / n_per_tri
    s2 = (dot(p1 - p0, v2).abs() + dot(p2 - p0, v2).abs() + dot(p3 - p0, v2).abs()) / n_per_tri
    s1 = s1 * torch.clamp(scales_raw[:, 1:2], min=0.5, max=2.0)
    s2 = s2 * torch.clamp(scales_raw[:, 2:3], min=0.5, max=2.0)
    return torch.cat((s0, s1, s2), dim=1), standardize_quaternion(matrix_to_quaternion(R))


# ----------------------------------------------------------------------------------------------------------------
# DreamWaltzG.animate (avatar.py:1500-1588) with the default flags (configs/__init__.py:117-126,194-197)
# ----------------------------------------------------------------------------------------------------------------
def lbs_weight_activation(w):
    return w / w.sum(dim=-1, keepdim=True)


def lbs_transform(positions, transforms, lbs_weights, quaternions=None):
    """avatar.py:1426-1462 with use_*_offsets all False."""
    jt = se3_compose(transforms["J_pose_rigid"], transforms["G_transl_offset"])[0]
    p = transform_points(jt, positions, weights=lbs_weights)
    if quaternions is None:
        return p
    return p, transform_quaternions_flip(jt, quaternions, lbs_weights)


def non_rigid_transform(positions, offsets, mlp_scales, mlp_quats, _scales, _quaternions, init_offset=0.01, init_scale=0.001,
                        max_scale=0.01, use_non_rigid_offsets=True, use_non_rigid_scales=True, use_non_rigid_rotations=False,
                        non_rigid_rotation_mode='add', learn_scale=True, learn_quaternions=True):
    """avatar.py:1464-1498, every branch that does not assert.  NB the scale branch is selected by non_rigid_ROTATION_mode
    (checklist Q4)."""
    if use_non_rigid_offsets:
        positions = positions + offsets * init_offset
    if use_non_rigid_scales:
        if learn_scale:
            if non_rigid_rotation_mode == 'add':
                scales = torch.exp(_scales) + mlp_scales * init_scale
            else:
                scales = torch.exp(_scales) * (1.0 + mlp_scales * init_scale)
        else:
            scales = (torch.exp(mlp_scales) * init_scale).clamp(max=max_scale)
    else:
        scales = torch.exp(_scales)
    if use_non_rigid_rotations:
        if learn_quaternions:
            if non_rigid_rotation_mode == 'add':
                quats = F.normalize(_quaternions, dim=-1) + mlp_quats
            else:
                quats = quaternion_multiply(F.normalize(mlp_quats, dim=-1), F.normalize(_quaternions, dim=-1))
        else:
            quats = F.normalize(mlp_quats, dim=-1)
    else:
        quats = F.normalize(_quaternions, dim=-1)
    return positions, scales, quats


def animate(params: Dict[str, torch.Tensor], nets: dict, body: SyntheticBody, smpl_observed: dict, smpl_canonical: dict,
            mesh: Optional[dict] = None, nerf_bound=2.0, init_offset=0.01, init_scale=0.001, extra_betas=None):
    """params: _positions [N,3], _scales [N,3] (log), _quaternions [N,4], _lbs_weights [N,55].
    mesh (optional): dict(vertex_indices [Vp], triangles [Fp,3] (local), vertex_coords [Vp,3], bary [Fp,n,3], scales [M,3]).
    extra_betas (optional, [1,300]): `learn_hand_betas` -- the mesh-bound vertices go through a second pair of skeleton passes with
    `extra_betas=_betas` (avatar.py:1551-1565); the free Gaussians keep the un-shaped skeleton.
    Returns dict positions/opacities/colors/quaternions/scales in the reference's GaussianOutput layout."""
    _, cV, ctr = glbs_forward(body, **smpl_canonical)
    _, oV, otr = glbs_forward(body, **smpl_observed)
    positions = params["_positions"]
    w = lbs_weight_activation(params["_lbs_weights"])
    canonical_positions = lbs_transform(positions, ctr, w)
    x01 = (canonical_positions + nerf_bound) / (2 * nerf_bound)
    enc = grid_encode(x01, nets["table"], nets["offsets"], nets["per_level_scale"])
    oc = mlp_forward(enc, nets["static_w"], nets["static_b"])
    colors = torch.sigmoid(oc[:, 1:]); opacities = torch.sigmoid(oc[:, :1])
    body_pose = smpl_observed.get("body_pose", torch.zeros(1, 63, dtype=positions.dtype))
    offsets, mlp_scales, mlp_quats = deform_forward(enc, body_pose, nets["deform"])
    pos, scales, quats = non_rigid_transform(positions, offsets, mlp_scales, mlp_quats, params["_scales"], params["_quaternions"],
                                             init_offset=init_offset, init_scale=init_scale)
    pos, quats = lbs_transform(pos, otr, w, quats)
    out = dict(positions=pos, opacities=opacities, colors=colors, quaternions=quats, scales=scales)
    if mesh is not None:
        if extra_betas is not None:
            _, cV, _ = glbs_forward(body, **smpl_canonical, extra_betas=extra_betas)
            _, oV, _ = glbs_forward(body, **smpl_observed, extra_betas=extra_betas)
        vi = mesh["vertex_indices"]
        cvc = transform_points(cV[0], mesh["vertex_coords"], indices=vi)
        cpos = mesh_positions(mesh["bary"], cvc, mesh["triangles"])
        enc_m = grid_encode((cpos + nerf_bound) / (2 * nerf_bound), nets["table"], nets["offsets"], nets["per_level_scale"])
        ocm = mlp_forward(enc_m, nets["static_w"], nets["static_b"])
        col_m = torch.sigmoid(ocm[:, 1:]); op_m = torch.ones_like(ocm[:, :1])
        ovc = transform_points(oV[0], mesh["vertex_coords"], indices=vi)
        pos_m = mesh_positions(mesh["bary"], ovc, mesh["triangles"])
        sc_m, q_m = mesh_scales_and_quaternions(mesh["bary"], mesh["scales"], ovc, mesh["triangles"], pos_m,
                                                mesh["bary"].shape[1])
        out = dict(positions=torch.cat([pos, pos_m]), opacities=torch.cat([opacities, op_m]),
                   colors=torch.cat([colors, col_m]), quaternions=torch.cat([quats, q_m]), scales=torch.cat([scales, sc_m]))
    return out
