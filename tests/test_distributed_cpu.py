"""world-size-2 `gloo` test (CPU) of the multi-view data path: every trainable parameter lives in one flat buffer, autograd
writes into flat gradient views, ONE all_reduce makes the gradients identical on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import optim
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # identical parameters on every rank
    a = torch.nn.Parameter(torch.randn(7, 3)); b = torch.nn.Parameter(torch.randn(5)); c = torch.nn.Parameter(torch.randn(2, 2, 2))
    opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec([dict(params=[a, b], lr=1e-3)], eps=1e-15),
                                        "nerf": optim.AdamSpec([dict(params=[c], lr=1e-2)], betas=(0.9, 0.99), eps=1e-15)}, torch.device("cpu"))
    opt = opts.buffers
    # parameters and gradients are views into the flat buffers, slices 16-byte aligned; the named optimizers are views of it
    assert a.data.data_ptr() == opt.flat.data_ptr() and a.grad.data_ptr() == opt.grad.data_ptr()
    assert all(pg["start"] % 4 == 0 and pg["end"] % 4 == 0 for o in opts.values() for pg in o.param_groups)
    assert opts.all_grads() is opt.grad
    for o in opts.values():
        o.zero_grad()
    x = torch.full((3,), float(rank + 1))                  # distinct "view" per rank
    loss = (a @ x).sum() * (rank + 1) + (b * b).sum() + c.sum() * (10 * rank + 1)
    loss.backward()                                        # accumulates INTO the flat gradient buffer
    local = opt.grad.clone()
    dist.all_reduce(opt.grad)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(opt.grad, sum(gathered))
    q.put((rank, opt.grad.clone(), a.grad.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1])              # identical reduced gradients on both ranks
    # d/da of (a @ x).sum() * (r+1) with x = r+1  ->  (r+1)^2 per entry; summed over ranks 1 + 4 = 5
    assert torch.allclose(res[0][2], torch.full((7, 3), 5.0))
