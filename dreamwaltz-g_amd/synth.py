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
