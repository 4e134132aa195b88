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
    def forward(ctx, A, weights, points, quats, normalize, grad_on=True):
        if not points.is_cuda:
            raise RuntimeError("dreamwaltz_g_amd LBS runs on the GPU only (HIP kernels)")
        A = A.detach().contiguous().float().reshape(-1, 16)
        weights = weights.detach().contiguous().float()
        points = points.contiguous().float()
        quats = None if quats is None else quats.contiguous().float()
        N, J = weights.shape
        p_out = torch.empty_like(points)
        q_out = torch.empty_like(quats) if quats is not None else None
        # the blended transforms are kept for the backward -- not under no_grad / inference_mode (`grad_on`: the CALLER's grad mode; inside
        # forward() it is always off): 48 bytes per Gaussian and launch that playback never reads
        T12 = torch.empty(N, 12, device=points.device) if grad_on else None
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
        return None, None, gp, gq, None, None


def lbs_blend(A, weights, points, quats=None, normalize_weights=False):
    """A [J,4,4] joint transforms (incl. translation), weights [N,J] -> points' (, quats')."""
    return _LbsBlend.apply(A, weights, points, quats, normalize_weights, torch.is_grad_enabled())


def lbs_blend_quaternions(A, weights, quats, normalize_weights=False):
    """RigidTransform.transform_quaternions(q, weights=, flip_rotation_axis=True) on its own (inverse_lbs.py:234-242): the same
    kernel with a zero point set (the weight rows dominate the traffic either way)."""
    pts = torch.zeros(quats.shape[0], 3, device=quats.device, dtype=torch.float32)
    return _LbsBlend.apply(A, weights, pts, quats, normalize_weights, torch.is_grad_enabled())[1]


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


class _VertexTransformFn(torch.autograd.Function):
    """out = transform_V[subset](vertex_coords); differentiable w.r.t. the shape coefficients (see dwg_lbs.h)."""

    @staticmethod
    def forward(ctx, vertex_coords, shape_coeffs, A, w_sub, sd, pd, rot_mats, pose, parents, joint_shape_dirs):
        Vp, J = w_sub.shape
        x = vertex_coords.detach().contiguous().float()
        A = A.contiguous().float()
        sc = None if shape_coeffs is None else shape_coeffs.detach().reshape(-1).contiguous().float()
        out = torch.empty(Vp, 3, device=x.device)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_lbs_vertex_transform(
            Vp, J, 0 if sd is None else sd.shape[-1], 0 if pd is None else pd.shape[-1], p(x), p(A), p(w_sub), p(sd), p(sc), p(pd),
            p(None if rot_mats is None else rot_mats.contiguous().float()), p(out), _st(x)), "dwg_lbs_vertex_transform")
        ctx.save_for_backward(A, w_sub, sd, pose, parents, joint_shape_dirs)
        ctx.shape_shape = None if shape_coeffs is None else shape_coeffs.shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        A, w_sub, sd, pose, parents, jdirs = ctx.saved_tensors
        if ctx.shape_shape is None or not ctx.needs_input_grad[1]:
            return (None,) * 10
        if sd is None or pose is None or jdirs is None:
            raise RuntimeError("gradient w.r.t. the shape coefficients needs shapedirs and the joint-chain inputs (pose, parents, "
                               "joint_shape_dirs)")
        Vp, J = w_sub.shape
        S = sd.shape[-1]
        g_shape = torch.empty(S, device=A.device)
        scratch = torch.empty(J, 3, device=A.device)
        p = _lib.ptr
        ws = torch.empty(int(_lib.lib().dwg_lbs_vertex_transform_backward_shape_workspace_floats(Vp)), device=A.device)     # one row of partial sums per workgroup
        _lib.check(_lib.lib().dwg_lbs_vertex_transform_backward_shape_ws(
            Vp, J, S, p(A), p(w_sub), p(sd), p(g_out.contiguous().float()), p(pose), p(parents), p(jdirs), p(scratch), p(g_shape), p(ws),
            _st(A)), "dwg_lbs_vertex_transform_backward_shape_ws")
        return None, g_shape.reshape(ctx.shape_shape), None, None, None, None, None, None, None, None


def vertex_transform(vertex_coords, A, subset, shape_coeffs=None, rot_mats=None, joint_chain_ctx=None, pose=None):
    """transform_V (compose(V_shape_offset, V_pose_offset, V_pose_rigid, transl)) applied to a fixed vertex subset;
    `subset` comes from gather_vertex_subset().  With joint_chain_ctx = (parents int32 [J], joint_shape_dirs [J,3,S], J_template)
    and pose [J,3] the result is differentiable w.r.t. shape_coeffs (both the vertex offsets and the rest joints inside A)."""
    w_sub, sd, pd = subset
    if not vertex_coords.is_cuda:
        raise RuntimeError("dreamwaltz_g_amd LBS runs on the GPU only (HIP kernels)")
    parents = jdirs = None
    if joint_chain_ctx is not None:
        parents, jdirs = joint_chain_ctx[0].to(torch.int32).contiguous(), joint_chain_ctx[1].contiguous().float()
    pose = None if pose is None else pose.detach().reshape(-1, 3).contiguous().float()
    return _VertexTransformFn.apply(vertex_coords, shape_coeffs, A, w_sub, sd, pd, rot_mats, pose, parents, jdirs)
