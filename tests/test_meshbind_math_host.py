"""CPU check of the hand-derived mesh-bound-Gaussian forward/backward (csrc/meshbind_math.h compiled for the host)
against autograd through the oracle restatement (oracle/animate.py: mesh_positions, mesh_scales_and_quaternions)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

from oracle import animate as oa

HERE = os.path.dirname(os.path.abspath(__file__))


def _hostlib():
    src = os.path.join(HERE, "hostmath", "meshbind_math_host.c")
    out = os.path.join(HERE, "hostmath", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libmeshbind_math_host.so")
    csrc = os.path.join(HERE, "..", "dreamwaltz-g_amd", "csrc")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(csrc, "meshbind_math.h")),
                 os.path.getmtime(os.path.join(csrc, "lbs_math.h")))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src, "-lm"], check=True)
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_meshbind_point_forward_and_backward_match_autograd():
    L = _hostlib()
    g = torch.Generator().manual_seed(3)
    V, n_per = 400, 6
    verts = torch.randn(V, 3, generator=g, dtype=torch.float64) * 0.2
    tri = torch.randint(0, V, (300, 3), generator=g)
    tri = tri[(tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])]
    Fp = tri.shape[0]
    base = torch.tensor([[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3], [1 / 6, 5 / 12, 5 / 12],
                         [5 / 12, 1 / 6, 5 / 12], [5 / 12, 5 / 12, 1 / 6]], dtype=torch.float64)
    bary = (base.expand(Fp, -1, -1) * (1 + 0.3 * torch.rand(Fp, 6, 3, generator=g, dtype=torch.float64))).requires_grad_(True)
    scales = (torch.rand(Fp * n_per, 3, generator=g, dtype=torch.float64) * 2.5).requires_grad_(True)   # both clamp sides hit
    pos = oa.mesh_positions(bary, verts, tri)
    scl, quat = oa.mesh_scales_and_quaternions(bary, scales, verts, tri, pos, n_per)
    gpos = torch.randn(pos.shape, generator=g, dtype=torch.float64)
    gscl = torch.randn(scl.shape, generator=g, dtype=torch.float64)
    gquat = torch.randn(quat.shape, generator=g, dtype=torch.float64)
    verts.requires_grad_(True)
    pos = oa.mesh_positions(bary, verts, tri)
    scl, quat = oa.mesh_scales_and_quaternions(bary, scales, verts, tri, pos, n_per)
    gb_ref, gs_ref, gv_ref = torch.autograd.grad([pos, scl, quat], [bary, scales, verts], [gpos, gscl, gquat])
    verts = verts.detach()
    vn, _ = oa.compute_normal(verts, tri)
    M = Fp * n_per
    p2t = torch.arange(Fp)[:, None].expand(-1, n_per).reshape(-1)
    P = verts[tri[p2t]].float().numpy().copy()            # [M,3,3]
    Nv = vn[tri[p2t]].float().numpy().copy()
    b32 = bary.detach().reshape(M, 3).float().numpy().copy(); s32 = scales.detach().float().numpy().copy()
    o_pos = np.zeros((M, 3), np.float32); o_scl = np.zeros((M, 3), np.float32); o_q = np.zeros((M, 4), np.float32)
    L.host_meshbind_forward(M, ctypes.c_float(n_per), _p(b32), _p(s32), _p(P), _p(Nv), _p(o_pos), _p(o_scl), _p(o_q))
    assert np.abs(o_pos - pos.detach().numpy()).max() < 1e-6
    assert np.abs(o_scl - scl.detach().numpy()).max() < 1e-6
    assert np.abs(o_q - quat.detach().numpy()).max() < 2e-5
    assert (o_scl[:, 0] == 0).all()
    gb = np.zeros((M, 3), np.float32); gs = np.zeros((M, 3), np.float32)
    gP = np.zeros((M, 3, 3), np.float32); gN = np.zeros((M, 3, 3), np.float32)
    L.host_meshbind_backward(M, ctypes.c_float(n_per), _p(b32), _p(s32), _p(P), _p(Nv), _p(gpos.float().numpy().copy()),
                             _p(gscl.float().numpy().copy()), _p(gquat.float().numpy().copy()), _p(gb), _p(gs), _p(gP), _p(gN))
    assert _rel(gb, gb_ref.reshape(M, 3).numpy()) < 1e-4, _rel(gb, gb_ref.reshape(M, 3).numpy())
    assert _rel(gs, gs_ref.numpy()) < 1e-5, _rel(gs, gs_ref.numpy())
    # gradient w.r.t. the posed vertices (learn_*_betas): per-point vertex / normal gradients scattered to the mesh, then the
    # vertex-normal chain (compute_normal) back to the vertices -- against autograd through the whole oracle expression
    idx = tri[p2t].numpy()                                   # [M,3]
    g_vo = np.zeros((V, 3), np.float32); g_vn = np.zeros((V, 3), np.float32)
    np.add.at(g_vo, idx.reshape(-1), gP.reshape(-1, 3)); np.add.at(g_vn, idx.reshape(-1), gN.reshape(-1, 3))
    v32 = verts.float().numpy().copy(); t32 = tri.numpy().astype(np.int32).copy()
    L.host_vertex_normals_backward(V, Fp, _p(v32), _p(t32), _p(g_vn), _p(g_vo))
    assert _rel(g_vo, gv_ref.numpy()) < 2e-4, _rel(g_vo, gv_ref.numpy())
