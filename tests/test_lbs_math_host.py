"""CPU check of the hand-derived LBS point/quaternion backward (csrc/lbs_math.h compiled for the host)
against autograd through the oracle restatement (oracle/animate.py)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

from oracle import animate as oa

HERE = os.path.dirname(os.path.abspath(__file__))


def _hostlib():
    src = os.path.join(HERE, "hostmath", "lbs_math_host.c")
    out = os.path.join(HERE, "hostmath", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "liblbs_math_host.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "dreamwaltz-g_amd", "csrc", "lbs_math.h"))):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src, "-lm"], check=True)
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_lbs_apply_forward_and_backward_match_autograd():
    L = _hostlib()
    g = torch.Generator().manual_seed(0)
    n, J = 400, 55
    # blended (non-orthonormal) transforms from random rotations, as the reference produces them
    A = torch.eye(4, dtype=torch.float64).repeat(J, 1, 1)
    A[:, :3, :3] = oa.batch_rodrigues(torch.randn(J, 3, generator=g, dtype=torch.float64) * 0.6)
    A[:, :3, 3] = torch.randn(J, 3, generator=g, dtype=torch.float64) * 0.2
    w = torch.softmax(torch.randn(n, J, generator=g, dtype=torch.float64) * 2, -1)
    p = (torch.randn(n, 3, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    q = torch.randn(n, 4, generator=g, dtype=torch.float64).requires_grad_(True)  # deliberately not unit
    po = oa.transform_points(A, p, weights=w)
    qo = oa.transform_quaternions_flip(A, q, w)
    gpo = torch.randn(n, 3, generator=g, dtype=torch.float64); gqo = torch.randn(n, 4, generator=g, dtype=torch.float64)
    T = torch.einsum('nj,jkl->nkl', w, A)[:, :3, :].detach().requires_grad_(True)
    # autograd w.r.t. the blended transform as well
    po2 = (T[:, :, :3] @ p.unsqueeze(-1))[..., 0] + T[:, :, 3]
    flip = torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64)[None, :, None]
    qo2 = oa.matrix_to_quaternion(((T[:, :, :3] @ (oa.quaternion_to_matrix(q) * flip)) * flip))
    assert torch.allclose(po, po2) and torch.allclose(qo, qo2)
    gp_ref, gq_ref, gT_ref = torch.autograd.grad([po2, qo2], [p, q, T], [gpo, gqo])
    T32 = T.detach().reshape(n, 12).float().numpy().copy()
    p32 = p.detach().float().numpy().copy(); q32 = q.detach().float().numpy().copy()
    pout = np.zeros((n, 3), np.float32); qout = np.zeros((n, 4), np.float32)
    L.host_lbs_apply(n, _p(T32), _p(p32), _p(q32), _p(pout), _p(qout))
    assert np.abs(pout - po.detach().numpy()).max() < 1e-5
    assert np.abs(qout - qo.detach().numpy()).max() < 2e-5
    gp = np.zeros((n, 3), np.float32); gq = np.zeros((n, 4), np.float32); gT = np.zeros((n, 12), np.float32)
    gpo32 = gpo.float().numpy().copy(); gqo32 = gqo.float().numpy().copy()
    L.host_lbs_apply_bwd(n, _p(T32), _p(p32), _p(q32), _p(gpo32), _p(gqo32), _p(gp), _p(gq), _p(gT))
    for got, ref, name in ((gp, gp_ref, "gp"), (gq, gq_ref, "gq"), (gT, gT_ref.reshape(n, 12), "gT")):
        ref = ref.numpy()
        err = np.abs(got - ref) / (np.abs(ref) + np.abs(ref).mean())
        assert err.max() < 2e-3, (name, err.max())
    # all four matrix_to_quaternion branches must have been exercised
    M = ((T[:, :, :3] @ (oa.quaternion_to_matrix(q) * flip)) * flip).detach()
    x = torch.stack([1 + M[:, 0, 0] + M[:, 1, 1] + M[:, 2, 2], 1 + M[:, 0, 0] - M[:, 1, 1] - M[:, 2, 2],
                     1 - M[:, 0, 0] + M[:, 1, 1] - M[:, 2, 2], 1 - M[:, 0, 0] - M[:, 1, 1] + M[:, 2, 2]], -1)
    assert set(x.argmax(-1).tolist()) == {0, 1, 2, 3}


def test_joint_chain_rest_joint_backward_matches_autograd():
    """d loss / d (rest joints) of A[:, :3, 3] through smplx batch_rigid_transform (the `learn_*_betas` path, avatar.py:1551-1553):
    hand-derived chain (csrc/lbs_math.h) vs autograd through the oracle."""
    L = _hostlib()
    g = torch.Generator().manual_seed(4)
    J = 55
    pose = torch.randn(J, 3, generator=g, dtype=torch.float64) * 0.5
    joints = (torch.randn(J, 3, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    parents = np.array([-1] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, J)], dtype=np.int64)
    R = oa.batch_rodrigues(pose)
    _, A = oa.batch_rigid_transform(R[None], joints[None], parents)
    g_t = torch.randn(J, 3, generator=g, dtype=torch.float64)
    (ref,) = torch.autograd.grad(A[0, :, :3, 3], joints, g_t)
    dJ = np.zeros((J, 3), np.float32)
    L.host_joint_chain_rest_joint_bwd(J, _p(pose.float().numpy().copy()), _p(parents.astype(np.int32).copy()),
                                      _p(g_t.float().numpy().copy()), _p(dJ))
    err = np.linalg.norm(dJ - ref.numpy()) / np.linalg.norm(ref.numpy())
    assert err < 1e-5, err
