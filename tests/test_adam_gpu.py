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


def test_one_launch_for_all_groups_equals_one_launch_per_group():
    """dwg_adam_step_groups_dev (round 6: every group of every named optimizer of the captured step in one launch) against the per-group
    device-scalar launches it replaces (FlatOptimizer.launch_step): identical bits in parameters and both moments."""
    from dreamwaltz_g_amd import optim

    def build():
        torch.manual_seed(3)
        dev = torch.device("cuda")
        shapes = [(1000, 3), (17,), (64, 95), (5, 5, 5), (4096, 2)]
        ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
        opts = optim.build_flat_optimizers({"a": optim.AdamSpec([dict(params=ps[:2], lr=1e-3), dict(params=ps[2:3], lr=3e-3)], eps=1e-15),
                                            "b": optim.AdamSpec([dict(params=ps[3:], lr=1e-2)], betas=(0.9, 0.99), eps=1e-15)}, dev)
        opts.set_grad_scale(0.5)
        return ps, opts
    outs = []
    for fused in (False, True):
        ps, opts = build()
        rows = sum(len(o.param_groups) for o in opts.values())
        hyper_host = torch.zeros(rows, 4)
        hyper_dev = torch.zeros(rows, 4, device="cuda")
        for it in range(3):
            opts.zero_grad()
            g = torch.Generator(device="cuda").manual_seed(100 + it)
            opts.buffers.grad.copy_(torch.randn(opts.buffers.grad.shape, device="cuda", generator=g))
            base = 0
            for o in opts.values():
                base += o.prepare_step(hyper_host, base)
            hyper_dev.copy_(hyper_host)
            if fused:
                opts.launch_steps(hyper_dev)
            else:
                base = 0
                for o in opts.values():
                    base += o.launch_step(hyper_dev, base)
        torch.cuda.synchronize()
        outs.append((opts.buffers.flat.clone(), opts.buffers.m.clone(), opts.buffers.v.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_segment_copies_and_adds():
    """dwg_copy_segments / dwg_add_segments through their Python users: row merges (optionally normalised like the grid encoder's input) and
    the packed deformation heads whose gradients are added straight into flat-buffer slices."""
    from dreamwaltz_g_amd import assemble as asm
    g = torch.Generator().manual_seed(1)
    a = [torch.randn(n, 3, generator=g).cuda().requires_grad_(True) for n in (100, 7, 33)]
    b = [torch.randn(n, 4, generator=g).cuda().requires_grad_(True) for n in (100, 7, 33)]
    ca, cb = asm.concat_rows([a, b])
    assert torch.equal(ca, torch.cat(a, 0)) and torch.equal(cb, torch.cat(b, 0))
    (cn,) = asm.concat_rows([a], bound=1.5)
    assert torch.equal(cn, (torch.cat(a, 0) + 1.5) / (2 * 1.5))
    w = torch.randn(140, 3, generator=g).cuda()
    (cn * w).sum().backward()
    off = 0
    for t in a:
        assert torch.equal(t.grad, w[off:off + t.shape[0]] / 3.0)
        off += t.shape[0]
