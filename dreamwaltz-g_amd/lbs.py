"""MI355X-native LBS stage behind the reference's seams (boundary B3).

  lbs_blend(A, weights, points, quats)   the heavy part of DreamWaltzG.lbs_transform (avatar.py:1426-1462):
                                         RigidTransform.transform_points(weights=) + transform_quaternions(weights=,
                                         flip_rotation_axis=True) (inverse_lbs.py:190-242), differentiable w.r.t. points/quats
  joint_chain(...)                       smplx batch_rodrigues + batch_rigid_transform + compose with G_transl_offset
  vertex_transform(...)                  transform_V on a vertex subset (mesh-bound Gaussians)
csrc/lbs.hip through include/dwg_lbs.h; no CPU fallback.
"""
import ctypes

import torch

from . import _lib


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _LbsBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, weights, points, quats, normalize):
        if not points.is_cuda:
            raise RuntimeError("dreamwaltz_g_amd LBS runs on the GPU only (HIP kernels)")
        A = A.detach().contiguous().float().reshape(-1, 16)
        weights = weights.detach().contiguous().float()
        points = points.contiguous().float()
        quats = None if quats is None else quats.contiguous().float()
        N, J = weights.shape
        p_out = torch.empty_like(points)
        q_out = torch.empty_like(quats) if quats is not None else None
        need_bwd = points.requires_grad or (quats is not None and quats.requires_grad)
        T12 = torch.empty(N, 12, device=points.device) if need_bwd or True else None
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_lbs_blend_forward(N, J, int(bool(normalize)), p(A), p(weights), p(points), p(quats),
                                                    p(p_out), p(q_out), p(T12), _st(points)), "dwg_lbs_blend_forward")
        ctx.save_for_backward(T12, points, quats)
        ctx.has_q = quats is not None
        if quats is None:
            return p_out
        return p_out, q_out

    @staticmethod
    def backward(ctx, g_p, g_q=None):
        T12, points, quats = ctx.saved_tensors
        N = points.shape[0]
        g_p = torch.zeros_like(points) if g_p is None else g_p.contiguous().float()
        gp = torch.empty_like(points)
        gq = None
        if ctx.has_q:
            g_q = torch.zeros_like(quats) if g_q is None else g_q.contiguous().float()
            gq = torch.empty_like(quats)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_lbs_blend_backward(N, p(T12), p(points), p(quats), p(g_p), p(g_q) if ctx.has_q else None,
                                                     p(gp), p(gq), _st(points)), "dwg_lbs_blend_backward")
        return None, None, gp, gq, None


def lbs_blend(A, weights, points, quats=None, normalize_weights=False):
    """A [J,4,4] joint transforms (incl. translation), weights [N,J] -> points' (, quats')."""
    return _LbsBlend.apply(A, weights, points, quats, normalize_weights)


def joint_chain(pose, joints, parents, transl=None, return_rot_mats=False, joint_shape_dirs=None, shape_coeffs=None):
    """pose [J,3] axis-angle, joints [J,3], parents int32 [J] -> A [J,4,4] (= compose(J_pose_rigid, G_transl_offset))."""
    if not pose.is_cuda:
        raise RuntimeError("dreamwaltz_g_amd LBS runs on the GPU only (HIP kernels)")
    J = pose.shape[0]
    pose = pose.contiguous().float(); joints = joints.contiguous().float()
    parents = parents.to(device=pose.device, dtype=torch.int32).contiguous()
    transl = None if transl is None else transl.reshape(3).contiguous().float()
    A = torch.empty(J, 4, 4, device=pose.device)
    R = torch.empty(J, 3, 3, device=pose.device) if return_rot_mats else None
    p = _lib.ptr
    jd = None if joint_shape_dirs is None else joint_shape_dirs.contiguous().float()
    sc = None if shape_coeffs is None else shape_coeffs.reshape(-1).contiguous().float()
    _lib.check(_lib.lib().dwg_lbs_joint_chain(J, p(pose), p(joints), p(parents), p(transl), p(jd), p(sc),
                                              0 if sc is None else sc.numel(), p(A), p(R), _st(pose)),
               "dwg_lbs_joint_chain")
    return (A, R) if return_rot_mats else A


def gather_vertex_subset(vertex_indices, lbs_weights, shapedirs=None, posedirs=None):
    """One-off gather of the body-model rows a fixed vertex subset needs, into the vertex-major layouts of
    dwg_lbs_vertex_transform: (lbs_weights_sub [Vp,J], shapedirs_sub [Vp,3,S] | None, posedirs_sub [Vp,3,F] | None)."""
    vi = vertex_indices.long()
    w_sub = lbs_weights[vi].contiguous().float()
    sd = None if shapedirs is None else shapedirs[vi].contiguous().float()
    pd = None
    if posedirs is not None:
        Fp, V3 = posedirs.shape
        pd = posedirs.view(Fp, V3 // 3, 3)[:, vi, :].permute(1, 2, 0).contiguous().float()
    return w_sub, sd, pd


def vertex_transform(vertex_coords, A, subset, shape_coeffs=None, rot_mats=None):
    """transform_V (compose(V_shape_offset, V_pose_offset, V_pose_rigid, transl)) applied to a fixed vertex subset;
    `subset` comes from gather_vertex_subset()."""
    w_sub, sd, pd = subset
    Vp, J = w_sub.shape
    out = torch.empty(Vp, 3, device=vertex_coords.device)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_lbs_vertex_transform(
        Vp, J, 0 if sd is None else sd.shape[-1], 0 if pd is None else pd.shape[-1], p(vertex_coords.contiguous().float()),
        p(A.contiguous().float()), p(w_sub), p(sd), p(None if shape_coeffs is None else shape_coeffs.reshape(-1).contiguous().float()),
        p(pd), p(None if rot_mats is None else rot_mats.contiguous().float()), p(out), _st(vertex_coords)),
        "dwg_lbs_vertex_transform")
    return out
