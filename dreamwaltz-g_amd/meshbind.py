"""Mesh-bound Gaussians (hands / face) on the HIP kernels of csrc/meshbind.hip (include/dwg_meshbind.h).

Mirrors MeshBindingGaussianModel.get_positions / get_scales_and_quaternions (/root/reference/core/system/avatar.py:1016-1079)
and compute_normal (/root/reference/core/utils/mesh.py:34-94): one forward launch produces canonical positions, observed
positions, scales and quaternions; one backward launch produces the gradients of `_bary_coords` and `_scales`.
"""
import ctypes

import torch

from . import _lib


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dreamwaltz_g_amd.meshbind: HIP-only path, got a CPU tensor (no CPU fallback)")


def build_vertex_face_csr(triangles, n_vertices):
    """Vertex -> incident-face adjacency (static topology; built once on the host).  Order inside a vertex: all faces
    where it is corner 0, then corner 1, then corner 2 (the order of the reference's three index_add_ calls), each by
    ascending face id."""
    tri = triangles.detach().cpu().long()
    Fp = tri.shape[0]
    verts = tri.t().reshape(-1)                               # corner-major: [i0..., i1..., i2...]
    faces = torch.arange(Fp).repeat(3)
    order = torch.argsort(verts, stable=True)
    counts = torch.bincount(verts, minlength=n_vertices)
    offsets = torch.zeros(n_vertices + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts, 0)
    return offsets.to(torch.int32), faces[order].to(torch.int32)


def vertex_normals(verts, triangles_i32, vf_offsets, vf_faces):
    _need_cuda(verts, triangles_i32, vf_offsets, vf_faces)
    verts = verts.contiguous().float()
    Vp, Fp = verts.shape[0], triangles_i32.shape[0]
    fn = torch.empty(max(Fp, 1), 3, device=verts.device)
    vn = torch.empty(Vp, 3, device=verts.device)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_mesh_vertex_normals(Vp, Fp, p(verts), p(triangles_i32), p(vf_offsets), p(vf_faces), p(fn), p(vn),
                                                  _st(verts)), "dwg_mesh_vertex_normals")
    return vn


class _MeshBind(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bary, scales, verts_cnl, verts_obs, vnormals, triangles_i32, n_per_tri):
        _need_cuda(bary, scales, verts_cnl, verts_obs, vnormals, triangles_i32)
        bary_c, scales_c = bary.contiguous().float(), scales.contiguous().float()
        vc = None if verts_cnl is None else verts_cnl.contiguous().float()
        vo, vn = verts_obs.contiguous().float(), vnormals.contiguous().float()
        Fp = triangles_i32.shape[0]
        M = Fp * n_per_tri
        dev = bary.device
        pos_c = torch.empty(M, 3, device=dev) if vc is not None else None
        pos, scl, quat = torch.empty(M, 3, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 4, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_meshbind_forward(Fp, n_per_tri, p(bary_c), p(scales_c), p(vc), p(vo), p(vn), p(triangles_i32),
                                                   p(pos_c), p(pos), p(scl), p(quat), _st(bary)), "dwg_meshbind_forward")
        ctx.save_for_backward(bary_c, scales_c, vc, vo, vn, triangles_i32)
        ctx.n_per_tri = n_per_tri
        ctx.bary_shape = bary.shape
        if pos_c is None:
            pos_c = pos.new_empty(0, 3)
            ctx.mark_non_differentiable(pos_c)
        return pos_c, pos, scl, quat

    @staticmethod
    def backward(ctx, g_pos_c, g_pos, g_scl, g_quat):
        bary_c, scales_c, vc, vo, vn, tri = ctx.saved_tensors
        Fp = tri.shape[0]
        cg = lambda g: None if g is None else g.contiguous().float()  # noqa: E731
        g_pos_c = cg(g_pos_c) if vc is not None else None
        g_bary, g_sc = torch.empty_like(bary_c), torch.empty_like(scales_c)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_meshbind_backward(Fp, ctx.n_per_tri, p(bary_c), p(scales_c), p(vc), p(vo), p(vn), p(tri),
                                                    p(g_pos_c), p(cg(g_pos)), p(cg(g_scl)), p(cg(g_quat)), p(g_bary), p(g_sc),
                                                    _st(bary_c)), "dwg_meshbind_backward")
        return g_bary.reshape(ctx.bary_shape), g_sc, None, None, None, None, None


def meshbind(bary, scales, verts_cnl, verts_obs, vnormals, triangles_i32, n_per_tri):
    """-> (canonical positions [M,3] (empty when verts_cnl is None), positions [M,3], scales [M,3], quaternions [M,4])."""
    return _MeshBind.apply(bary, scales, verts_cnl, verts_obs, vnormals, triangles_i32, n_per_tri)


class _MeshBindFull(torch.autograd.Function):
    """vertex normals + meshbind in one autograd node, with gradients to the (canonical / posed) vertices when they require them
    (learn_hand_betas / learn_face_betas: the vertices then depend on `_betas`, avatar.py:1551-1577)."""

    @staticmethod
    def forward(ctx, bary, scales, verts_cnl, verts_obs, triangles_i32, vf_offsets, vf_faces, n_per_tri):
        _need_cuda(bary, scales, verts_cnl, verts_obs, triangles_i32)
        bary_c, scales_c = bary.detach().contiguous().float(), scales.detach().contiguous().float()
        vc = None if verts_cnl is None else verts_cnl.detach().contiguous().float()
        vo = verts_obs.detach().contiguous().float()
        vn = vertex_normals(vo, triangles_i32, vf_offsets, vf_faces)
        Fp = triangles_i32.shape[0]
        M = Fp * n_per_tri
        dev = bary.device
        pos_c = torch.empty(M, 3, device=dev) if vc is not None else None
        pos, scl, quat = torch.empty(M, 3, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 4, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().dwg_meshbind_forward(Fp, n_per_tri, p(bary_c), p(scales_c), p(vc), p(vo), p(vn), p(triangles_i32),
                                                   p(pos_c), p(pos), p(scl), p(quat), _st(bary)), "dwg_meshbind_forward")
        ctx.save_for_backward(bary_c, scales_c, vc, vo, vn, triangles_i32, vf_offsets, vf_faces)
        ctx.n_per_tri = n_per_tri
        ctx.bary_shape = bary.shape
        if pos_c is None:
            pos_c = pos.new_empty(0, 3)
            ctx.mark_non_differentiable(pos_c)
        return pos_c, pos, scl, quat

    @staticmethod
    def backward(ctx, g_pos_c, g_pos, g_scl, g_quat):
        bary_c, scales_c, vc, vo, vn, tri, vf_off, vf_faces = ctx.saved_tensors
        Fp, Vp = tri.shape[0], vo.shape[0]
        cg = lambda g: None if g is None else g.contiguous().float()  # noqa: E731
        g_pos_c = cg(g_pos_c) if vc is not None else None
        g_bary, g_sc = torch.empty_like(bary_c), torch.empty_like(scales_c)
        p = _lib.ptr
        L = _lib.lib()
        want_v = ctx.needs_input_grad[3] or (vc is not None and ctx.needs_input_grad[2])
        if not want_v:
            _lib.check(L.dwg_meshbind_backward(Fp, ctx.n_per_tri, p(bary_c), p(scales_c), p(vc), p(vo), p(vn), p(tri), p(g_pos_c),
                                               p(cg(g_pos)), p(cg(g_scl)), p(cg(g_quat)), p(g_bary), p(g_sc), _st(bary_c)),
                       "dwg_meshbind_backward")
            return g_bary.reshape(ctx.bary_shape), g_sc, None, None, None, None, None, None
        # vertex gradients without float atomics: per-(Gaussian, corner) rows, then a per-vertex gather in incident-face order (the same bits
        # on every run; the three outputs are overwritten, no zero fills)
        g_vc = torch.empty_like(vc) if (vc is not None and g_pos_c is not None) else None
        g_vo, g_vn = torch.empty_like(vo), torch.empty_like(vo)
        rows = torch.empty(Fp * ctx.n_per_tri * 27, device=vo.device)
        _lib.check(L.dwg_meshbind_backward_verts_gather(Vp, Fp, ctx.n_per_tri, p(bary_c), p(scales_c), p(vc), p(vo), p(vn), p(tri), p(vf_off),
                                                        p(vf_faces), p(g_pos_c), p(cg(g_pos)), p(cg(g_scl)), p(cg(g_quat)), p(g_bary), p(g_sc),
                                                        p(rows), p(g_vc), p(g_vo), p(g_vn), _st(bary_c)), "dwg_meshbind_backward_verts_gather")
        if g_vc is None and vc is not None:
            g_vc = torch.zeros_like(vc)
        fn = torch.empty(max(Fp, 1), 3, device=vo.device)
        gs = torch.empty(Vp, 3, device=vo.device)
        _lib.check(L.dwg_mesh_vertex_normals_backward(Vp, Fp, p(vo), p(tri), p(vf_off), p(vf_faces), p(g_vn), p(fn), p(gs), p(g_vo),
                                                      _st(bary_c)), "dwg_mesh_vertex_normals_backward")
        return g_bary.reshape(ctx.bary_shape), g_sc, g_vc, g_vo, None, None, None, None


def meshbind_full(bary, scales, verts_cnl, verts_obs, triangles_i32, vf_offsets, vf_faces, n_per_tri):
    """compute_normal + get_positions (canonical, observed) + get_scales_and_quaternions
    -> (canonical positions [M,3] (empty when verts_cnl is None), positions [M,3], scales [M,3], quaternions [M,4])."""
    return _MeshBindFull.apply(bary, scales, verts_cnl, verts_obs, triangles_i32, vf_offsets, vf_faces, n_per_tri)
