"""-m gpu: fused HIP Adam on the flat parameter buffer vs torch.optim.Adam (same hyper-parameters, eps 1e-15)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_adam_matches_torch_adam():
    from dreamwaltz_g_amd import optim
    torch.manual_seed(0)
    dev = torch.device("cuda")
    shapes = [(1000, 3), (17,), (64, 95), (5, 5, 5)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_ref = torch.optim.Adam([dict(params=ref[:2], lr=1e-3), dict(params=ref[2:], lr=1e-2, betas=(0.9, 0.99))], eps=1e-15)
    opts = optim.build_flat_optimizers({"a": optim.AdamSpec([dict(params=ps[:2], lr=1e-3)], eps=1e-15),
                                        "b": optim.AdamSpec([dict(params=ps[2:], lr=1e-2)], betas=(0.9, 0.99), eps=1e-15)}, dev)
    opts.set_grad_scale(0.5)
    for it in range(5):
        gs = [torch.randn(s, device=dev) * (it + 1) for s in shapes]
        for o in opts.values():
            o.zero_grad()
        for p, g, r in zip(ps, gs, ref):
            p.grad.copy_(g * 2.0)          # "sum over 2 ranks"; Adam applies grad_scale = 1/2
            r.grad = g.clone()
        for o in opts.values():
            o.step()
        opt_ref.step()
        for p, r in zip(ps, ref):
            assert torch.allclose(p.data, r.data, rtol=2e-5, atol=2e-6), float((p.data - r.data).abs().max())
