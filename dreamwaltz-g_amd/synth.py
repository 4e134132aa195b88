"""Seeded synthetic inputs of SURVEY.md section 8(d) (random-init Gaussians in a body-sized box)."""
import torch


def random_gaussians(G, seed=0, device="cpu", dtype=torch.float32, opacity_range=None):
    g = torch.Generator().manual_seed(seed)
    box = torch.tensor([0.4, 0.9, 0.2])
    positions = (torch.rand(G, 3, generator=g) * 2 - 1) * box
    scales = torch.rand(G, 3, generator=g) * (0.02 - 0.002) + 0.002
    quats = torch.nn.functional.normalize(torch.randn(G, 4, generator=g), dim=-1)
    if opacity_range is None:
        opacities = torch.sigmoid(torch.randn(G, 1, generator=g))
    else:
        lo, hi = opacity_range
        opacities = torch.rand(G, 1, generator=g) * (hi - lo) + lo
    colors = torch.rand(G, 3, generator=g)
    out = dict(positions=positions, scales=scales, quaternions=quats, opacities=opacities, colors=colors)
    return {k: v.to(dtype).to(device).contiguous() for k, v in out.items()}


def synthetic_body(V=10475, J=55, n_betas=300, n_expr=100, seed=0):
    """SMPL-X-shaped body tensors (the licensed model file is not available: SURVEY.md section 8c).  Same roles/shapes as the
    attributes GeneralLinearBlendSkinning copies from smplx.SMPLX (inverse_lbs.py:521-568)."""
    g = torch.Generator().manual_seed(seed)
    box = torch.tensor([0.4, 0.9, 0.2])
    Jr = torch.zeros(J, V)
    for j in range(J):
        Jr[j, torch.randint(0, V, (16,), generator=g)] = 1.0 / 16
    logits = torch.full((V, J), -1e9)
    logits.scatter_(1, torch.randint(0, J, (V, 4), generator=g), torch.randn(V, 4, generator=g))
    parents = [-1] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, J)]
    return dict(
        v_template=(torch.rand(V, 3, generator=g) * 2 - 1) * box,
        shapedirs=torch.randn(V, 3, n_betas, generator=g) * 1e-3, expr_dirs=torch.randn(V, 3, n_expr, generator=g) * 1e-3,
        posedirs=torch.randn((J - 1) * 9, V * 3, generator=g) * 1e-3, J_regressor=Jr, lbs_weights=torch.softmax(logits, dim=1),
        parents=torch.tensor(parents), betas=torch.zeros(1, n_betas), expression=torch.zeros(1, n_expr),
        pose_mean=torch.zeros(J * 3), jaw_pose=torch.zeros(1, 3), leye_pose=torch.zeros(1, 3), reye_pose=torch.zeros(1, 3))


def random_smpl_inputs(seed=0, pose_std=0.3, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    d = dict(body_pose=torch.randn(1, 63, generator=g) * pose_std, global_orient=torch.randn(1, 3, generator=g) * pose_std,
             left_hand_pose=torch.randn(1, 45, generator=g) * pose_std, right_hand_pose=torch.randn(1, 45, generator=g) * pose_std,
             expression=torch.randn(1, 100, generator=g) * 0.5, transl=torch.zeros(1, 3))
    return {k: v.to(device) for k, v in d.items()}
