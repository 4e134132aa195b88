"""-m gpu parity tests for the LBS stage and the grid encoder: HIP kernels (through the C-ABI) vs the CPU oracle.
fp32 tolerances: forward 2e-5 abs on O(1) values; gradients relative L2 <= 1e-4 (table gradients use fp32 atomics)."""
import numpy as np
import pytest
import torch

from oracle import animate as oa

pytestmark = pytest.mark.gpu


def _rel_l2(a, r):
    a = a.detach().double().cpu(); r = r.detach().double().cpu()
    return float((a - r).norm() / r.norm().clamp_min(1e-30))


def _skeleton(seed=0, J=55):
    g = torch.Generator().manual_seed(seed)
    pose = torch.randn(J, 3, generator=g) * 0.4
    pose[3] = 0.0  # exercises the |r + 1e-8| branch of Rodrigues
    joints = torch.randn(J, 3, generator=g) * 0.3
    parents = torch.tensor([-1] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, J)])
    transl = torch.randn(3, generator=g) * 0.1
    return pose, joints, parents, transl


def _oracle_A(pose, joints, parents, transl):
    R = oa.batch_rodrigues(pose.double())
    _, A = oa.batch_rigid_transform(R[None], joints.double()[None], parents.numpy())
    if transl is not None:
        A = oa.se3_compose(A, oa.se3_from_T(transl.double()[None]))
    return A[0], R


@pytest.mark.parametrize("with_transl", [True, False])
def test_joint_chain(with_transl):
    from dreamwaltz_g_amd import lbs
    pose, joints, parents, transl = _skeleton()
    tr = transl if with_transl else None
    A_ref, R_ref = _oracle_A(pose, joints, parents, tr)
    A, R = lbs.joint_chain(pose.cuda(), joints.cuda(), parents, None if tr is None else tr.cuda(), return_rot_mats=True)
    assert (A.cpu().double() - A_ref).abs().max() < 2e-5
    assert (R.cpu().double() - R_ref).abs().max() < 2e-6


@pytest.mark.parametrize("N,with_q,normalize", [(1000, True, True), (257, True, False), (64, False, True), (1, True, True),
                                                (50000, True, True)])
def test_lbs_blend_forward_backward(N, with_q, normalize):
    from dreamwaltz_g_amd import lbs
    pose, joints, parents, transl = _skeleton(1)
    A_ref, _ = _oracle_A(pose, joints, parents, transl)
    g = torch.Generator().manual_seed(N)
    w_raw = torch.rand(N, 55, generator=g) * (torch.rand(N, 55, generator=g) < 0.2) + 1e-3
    p = torch.randn(N, 3, generator=g) * 0.4
    q = torch.randn(N, 4, generator=g)
    pd = p.double().requires_grad_(True); qd = q.double().requires_grad_(True)
    w = w_raw.double()
    wn = w / w.sum(-1, keepdim=True) if normalize else w
    po_ref = oa.transform_points(A_ref, pd, weights=wn)
    outs = [po_ref]
    if with_q:
        qo_ref = oa.transform_quaternions_flip(A_ref, qd, wn)
        outs.append(qo_ref)
    pc = p.cuda().requires_grad_(True); qc = q.cuda().requires_grad_(True)
    res = lbs.lbs_blend(A_ref.float().cuda(), w_raw.cuda(), pc, qc if with_q else None, normalize_weights=normalize)
    res = res if with_q else (res,)
    assert (res[0].detach().cpu().double() - po_ref.detach()).abs().max() < 2e-5
    if with_q:
        assert (res[1].detach().cpu().double() - qo_ref.detach()).abs().max() < 5e-5
    gs = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in outs]
    grads_ref = torch.autograd.grad(outs, [pd, qd] if with_q else [pd], gs)
    torch.autograd.backward(list(res), [x.float().cuda() for x in gs])
    assert _rel_l2(pc.grad, grads_ref[0]) < 1e-4
    if with_q:
        assert _rel_l2(qc.grad, grads_ref[1]) < 1e-4


def test_vertex_transform_matches_transform_V():
    from dreamwaltz_g_amd import lbs
    body = oa.SyntheticBody(V=2000, F_=3000, seed=2)
    inp = oa.random_smpl_inputs(seed=4)
    _, tV, tr = oa.glbs_forward(body, **inp)
    g = torch.Generator().manual_seed(0)
    vi = torch.randperm(2000, generator=g)[:700]
    x = body.v_template[vi]
    ref = oa.transform_points(tV[0], x, indices=vi)
    full_shape = oa.glbs_full_shape(body, expression=inp["expression"])
    full_pose = oa.glbs_full_pose(body, **{k: v for k, v in inp.items() if k.endswith("pose") or k == "global_orient"})
    shapedirs = torch.cat([body.shapedirs, body.expr_dirs], -1)
    v_shaped = body.v_template + oa.blend_shapes(full_shape, shapedirs)[0]
    joints = torch.einsum('ik,ji->jk', v_shaped, body.J_regressor)
    A, R = lbs.joint_chain(full_pose.view(55, 3).cuda(), joints.cuda(), torch.from_numpy(body.parents), inp["transl"][0].cuda(),
                           return_rot_mats=True)
    out = lbs.vertex_transform(vi.cuda(), x.cuda(), A, body.lbs_weights.cuda(), shapedirs.cuda(), full_shape.cuda(),
                               body.posedirs.cuda(), R)
    assert (out.cpu() - ref).abs().max() < 2e-5


def _grid_case(B, seed, gridtype=1, oob=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, generator=g)
    if oob and B > 8:
        x[3] = torch.tensor([1.2, 0.5, 0.5]); x[5] = torch.tensor([0.5, -0.01, 0.5])
        x[6] = torch.tensor([0.0, 1.0, 0.5])  # boundary values are in range
    offsets, pls = oa.grid_offsets()
    table = (torch.rand(int(offsets[-1]), 2, generator=g) * 2 - 1) * 0.1
    return x, table, offsets, pls, gridtype


@pytest.mark.parametrize("B,gridtype", [(4096, 1), (1000, 0), (1, 1), (33, 1)])
def test_grid_encoder_forward_backward(B, gridtype):
    from dreamwaltz_g_amd.gridencoder import grid_encode
    x, table, offsets, pls, _ = _grid_case(B, B, gridtype)
    xd = x.double().requires_grad_(True); td = table.double().requires_grad_(True)
    ref = oa.grid_encode(xd, td, offsets, pls, gridtype=gridtype)
    xc = x.cuda().requires_grad_(True); tc = table.cuda().requires_grad_(True)
    out = grid_encode(xc, tc, torch.from_numpy(offsets).cuda(), pls, 16, True, gridtype, False, 1)
    err = (out.detach().cpu().double() - ref.detach()).abs()
    # fp32 position arithmetic at the finest level (scale 4095: ulp(pos) = 2.4e-4 of a cell) against the float64 oracle:
    # |err| <= ulp * smoothstep' (1.5) * neighbour difference (<= 2 * 0.1 table amplitude) ~ 7e-5; the reference's table is
    # initialised to +-1e-4 (grid.py:142-144), i.e. 1000x smaller absolute errors in practice.  A floor() flip between
    # fma and mul+add evaluation moves a point to the neighbouring cell where the interpolant is continuous.
    assert err.max() < 7e-5, err.max()
    g = torch.Generator().manual_seed(1)
    go = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gx_ref, gt_ref = torch.autograd.grad(ref, [xd, td], go)
    out.backward(go.float().cuda())
    assert _rel_l2(tc.grad, gt_ref) < 1e-4
    assert _rel_l2(xc.grad, gx_ref) < 2e-3   # d/dx is scaled by up to 4095 per level: fp32 cancellation


def test_grid_encoder_backend_layout_and_module():
    """[L,B,C] layout of the `_gridencoder` backend + the GridEncoder module mirror (bound=2 mapping, grid.py:149-165)."""
    from dreamwaltz_g_amd import gridencoder as ge
    x, table, offsets, pls, _ = _grid_case(512, 7)
    ref = oa.grid_encode(x.double(), table.double(), offsets, pls)
    outs = torch.empty(16, 512, 2, device="cuda")
    ge.grid_encode_forward(x.cuda(), table.cuda(), torch.from_numpy(offsets).cuda(), outs, 512, 3, 2, 16,
                           float(np.log2(pls)), 16, None, 1, False, 1, 0)
    got = outs.permute(1, 0, 2).reshape(512, 32).cpu().double()
    assert (got - ref).abs().max() < 7e-5
    enc = ge.GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                         desired_resolution=4096, gridtype='tiled', align_corners=False, interpolation='smoothstep').cuda()
    assert enc.embeddings.shape == (6328848, 2)
    pos = (torch.rand(100, 3) * 4 - 2)
    y = enc(pos.cuda(), bound=2)
    ref2 = oa.grid_encode(((pos + 2) / 4).double(), enc.embeddings.detach().cpu().double(), offsets, pls)
    assert (y.detach().cpu().double() - ref2).abs().max() < 1e-6
