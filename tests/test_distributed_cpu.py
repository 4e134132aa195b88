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
    q.put((rank, opt.grad.numpy().copy(), a.grad.numpy().copy()))      # by value: a tensor travels as a file descriptor its sender must outlive
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
    res = [(r, torch.from_numpy(g), torch.from_numpy(ag)) for r, g, ag in res]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1])              # identical reduced gradients on both ranks
    # d/da of (a @ x).sum() * (r+1) with x = r+1  ->  (r+1)^2 per entry; summed over ranks 1 + 4 = 5
    assert torch.allclose(res[0][2], torch.full((7, 3), 5.0))


# ---------------------------------------------------------------------------------------------------------------------------------------
# the REAL loop body (trainer.SDSTrainer.train_step) on two gloo ranks: per-rank view accumulation, the one all-reduce, the 1 / V fold
# into Adam, replica identity after the step, and equality with one process that accumulates the same views.  The HIP kernels cannot run
# here, so the scene is a small differentiable stand-in and the fused Adam launch is replaced by the same arithmetic in torch.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _cpu_adam_launch(self, pg):
    """csrc/elementwise.hip k_adam restated in torch (FlatOptimizer._launch): one group's slice of the flat buffers."""
    b = self.buf
    s = slice(pg["start"], pg["end"])
    g = b.grad[s] * self.grad_scale
    b1, b2 = pg["betas"]
    b.m[s].mul_(b1).add_(g, alpha=1 - b1)
    b.v[s].mul_(b2).addcmul_(g, g, value=1 - b2)
    mh = b.m[s] / (1 - b1 ** pg["t"])
    vh = b.v[s] / (1 - b2 ** pg["t"])
    b.flat[s].sub_(pg["lr"] * mh / (vh.sqrt() + pg["eps"]))


class _ToyScene(torch.nn.Module):
    """Scene stand-in: image[1,H,W,3] = a smooth function of the parameters and of the view's camera scalar."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(6, 3, generator=g))
        self.b = torch.nn.Parameter(torch.randn(5, generator=g))

    def forward(self, data, smpl_observed_inputs=None, use_densifier=False, bg_mode=None):
        az = data["azimuth"].float().reshape(())
        pose = smpl_observed_inputs["body_pose"].float().sum() if smpl_observed_inputs is not None else 0.0
        base = torch.linspace(0, 1, 4 * 4 * 3).reshape(1, 4, 4, 3)
        img = torch.sin(base * self.a.sum() + az * 0.01 + pose) + (self.b ** 2).sum() * torch.cos(base * (1.0 + az * 0.003))
        return {"image": img, "alpha": torch.ones(1, 4, 4, 1), "depth": torch.ones(1, 4, 4, 1)}


def _toy_views(which, step):
    views = []
    for v in which:
        g = torch.Generator().manual_seed(1000 * v + step)
        views.append({"azimuth": torch.tensor([45.0 * v]), "elevation": torch.tensor([80.0]), "radius": torch.tensor([2.0]),
                      "tanfov": torch.tensor([0.52]), "smpl_inputs": {"body_pose": torch.randn(1, 63, generator=g) * 0.1},
                      "rng_seed": 77 + v})
    return views


def _toy_trainer(rank, world, views, dist_mod):
    from dreamwaltz_g_amd import configs, optim, sds_step, trainer
    optim.FlatOptimizer._launch = _cpu_adam_launch       # the fused HIP Adam launch, restated (this process only)
    cfg = configs.TrainConfig(); cfg.device = "cpu"; cfg.prompt.text_augmentation = False
    scene = _ToyScene()
    opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec([dict(params=[scene.a], lr=1e-2)], eps=1e-15),
                                        "nerf": optim.AdamSpec([dict(params=[scene.b], lr=1e-3)], betas=(0.9, 0.99), eps=1e-15)}, torch.device("cpu"))
    w = torch.randn(1, 4, 4, 3, generator=torch.Generator().manual_seed(9))
    tr = trainer.SDSTrainer(cfg, scene, sds_step._ImageLoss(w), opts, {}, use_controlnet=False, dist=dist_mod, world=world, max_step=100)
    tr.set_views(views)
    return tr, opts


def _trainer_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dwg_import  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V = 4
    tr, opts = _toy_trainer(rank, world, V, dist)
    for step in range(3):
        mine = _toy_views(range(rank, V, world), step)           # view v -> rank v mod world: two views per rank
        tr.train_step(mine)
    q.put((rank, opts.buffers.flat.numpy().copy(), opts.buffers.grad.numpy().copy()))      # by value (no fd passing after exit)
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_multiview_two_ranks_equal_one_process_accumulating_the_same_views():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    res = [(r, torch.from_numpy(a), torch.from_numpy(b)) for r, a, b in res]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])      # replicas bit-identical after three steps
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import optim
    saved = optim.FlatOptimizer._launch
    try:
        tr, opts = _toy_trainer(0, 1, 4, None)
        for step in range(3):
            tr.train_step(_toy_views(range(4), step))                                   # one process, all four views accumulated
    finally:
        optim.FlatOptimizer._launch = saved
    assert tr.total_views == 4 and opts["avatar"].grad_scale == 0.25
    assert torch.allclose(opts.buffers.grad, res[0][2], rtol=1e-5, atol=1e-7)           # summed (not yet averaged) gradients of the last step
    assert torch.allclose(opts.buffers.flat, res[0][1], rtol=1e-4, atol=1e-6)
    # a single dict is still the reference's one-view step
    tr1, opts1 = None, None
    try:
        tr1, opts1 = _toy_trainer(0, 1, 1, None)
        before = opts1.buffers.flat.clone()
        tr1.train_step(_toy_views([0], 0)[0])
    finally:
        optim.FlatOptimizer._launch = saved
    assert opts1["avatar"].grad_scale == 1.0 and not torch.equal(before, opts1.buffers.flat)


# ---------------------------------------------------------------------------------------------------------------------------------------
# SURVEY 8e replica-consistency guard: rank 0's parameters are broadcast at construction; a checksum of the flat parameter buffer is
# compared across the ranks every `replica_check_every` steps and a drift raises on EVERY rank.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _guard_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dwg_import  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["DWG_REPLICA_CHECK_EVERY"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V = 2
    from dreamwaltz_g_amd import configs, optim, sds_step, trainer
    optim.FlatOptimizer._launch = _cpu_adam_launch
    cfg = configs.TrainConfig(); cfg.device = "cpu"; cfg.prompt.text_augmentation = False
    scene = _ToyScene()
    with torch.no_grad():
        scene.a.add_(0.25 * rank)                       # replicas that were initialised DIFFERENTLY ...
    opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec([dict(params=[scene.a], lr=1e-2)], eps=1e-15),
                                        "nerf": optim.AdamSpec([dict(params=[scene.b], lr=1e-3)], betas=(0.9, 0.99), eps=1e-15)}, torch.device("cpu"))
    w = torch.randn(1, 4, 4, 3, generator=torch.Generator().manual_seed(9))
    tr = trainer.SDSTrainer(cfg, scene, sds_step._ImageLoss(w), opts, {}, use_controlnet=False, dist=dist, world=world, max_step=100)
    tr.set_views(V)
    start = opts.buffers.flat.clone()                   # ... are rank 0's after construction
    log = []
    for step in range(4):                               # checks at steps 2 and 4: clean
        tr.train_step(_toy_views(range(rank, V, world), step))
    log.append("clean")
    if rank == 1:
        with torch.no_grad():
            scene.b[0] += 1e-6                          # one rank drifts by one fp32 word
    try:
        for step in range(4, 6):                        # the check at step 6 must raise on BOTH ranks
            tr.train_step(_toy_views(range(rank, V, world), step))
        log.append("not detected")
    except RuntimeError as e:
        log.append("raised" if "replica drift" in str(e) else "other: %s" % e)
    q.put((rank, start.numpy().copy(), log))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_guard_broadcasts_rank0_and_raises_on_drift_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert (res[0][1] == res[1][1]).all()               # the differently initialised rank took rank 0's parameters
    assert res[0][2] == ["clean", "raised"] and res[1][2] == ["clean", "raised"], (res[0][2], res[1][2])


# ---------------------------------------------------------------------------------------------------------------------------------------
# The exchange step in slices (trainer._reduce_and_step, round 5): one asynchronous all-reduce per named optimizer, each optimizer stepping
# behind ITS slice; which groups take part in the step is agreed on across the ranks (MAX of a per-group flag) -- a group one rank's
# backward reached steps on EVERY rank, a group no rank reached keeps its moments and step count everywhere.
# ---------------------------------------------------------------------------------------------------------------------------------------
class _ToyScene3(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(6, 3, generator=g))
        self.b = torch.nn.Parameter(torch.randn(5, generator=g))
        self.c = torch.nn.Parameter(torch.randn(4, generator=g))       # no view ever depends on it
        self.register_buffer("marker", torch.zeros(3))

    def forward(self, data, smpl_observed_inputs=None, use_densifier=False, bg_mode=None):
        az = data["azimuth"].float().reshape(())
        base = torch.linspace(0, 1, 4 * 4 * 3).reshape(1, 4, 4, 3)
        img = torch.sin(base * self.a.sum() + az * 0.01)
        if bool(data.get("use_b", True)):
            img = img + (self.b ** 2).sum() * torch.cos(base * (1.0 + az * 0.003))
        return {"image": img, "alpha": torch.ones(1, 4, 4, 1), "depth": torch.ones(1, 4, 4, 1)}


def _participation_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dwg_import  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreamwaltz_g_amd import configs, optim, sds_step, trainer
    optim.FlatOptimizer._launch = _cpu_adam_launch
    cfg = configs.TrainConfig(); cfg.device = "cpu"; cfg.prompt.text_augmentation = False
    scene = _ToyScene3()
    with torch.no_grad():
        scene.marker.fill_(float(rank + 1))                # a buffer outside the flat parameter storage: rank 0's after construction
    opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec([dict(params=[scene.a], lr=1e-2)], eps=1e-15),
                                        "nerf": optim.AdamSpec([dict(params=[scene.b], lr=1e-3)], betas=(0.9, 0.99), eps=1e-15),
                                        "idle": optim.AdamSpec([dict(params=[scene.c], lr=1e-1)], eps=1e-15)}, torch.device("cpu"))
    if rank == 1:                                          # a rank whose optimizer scalars differ (e.g. it alone resumed a checkpoint) ...
        opts["nerf"].param_groups[0]["t"] = 5; opts["nerf"].t = 5; opts["nerf"].param_groups[0]["lr"] = 7e-3
    w = torch.randn(1, 4, 4, 3, generator=torch.Generator().manual_seed(9))
    tr = trainer.SDSTrainer(cfg, scene, sds_step._ImageLoss(w), opts, {}, use_controlnet=False, dist=dist, world=world, max_step=100)
    synced = (float(scene.marker[0]), opts["nerf"].param_groups[0].get("t", 0), opts["nerf"].param_groups[0]["lr"])   # ... takes rank 0's
    tr.set_views(2)
    c0 = scene.c.detach().clone()
    for step in range(3):
        views = _toy_views([rank], step)
        views[0]["use_b"] = (rank == 0)                    # only rank 0's view reaches `b`; nobody reaches `c`
        tr.train_step(views)
    buf = opts.buffers
    q.put((rank, buf.flat.numpy().copy(), buf.m.numpy().copy(), [pg.get("t", 0) for o in opts.values() for pg in o.param_groups],
           bool(torch.equal(scene.c.detach(), c0)), synced, tr.allreduce_steps >= 0))
    dist.barrier()
    dist.destroy_process_group()


def test_sliced_reduce_agrees_on_participation_across_ranks_and_syncs_optimizer_scalars():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_participation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, f0, m0, t0, c_same0, s0, _), (r1, f1, m1, t1, c_same1, s1, _) = res
    assert (f0 == f1).all() and (m0 == m1).all()          # replicas bit-identical: parameters AND first moments
    assert t0 == t1 == [3, 3, 0]                          # `b` (reached by rank 0 only) stepped on both ranks, `c` (reached by nobody) on neither
    assert c_same0 and c_same1
    assert s0 == s1 == (1.0, 0, 1e-3)                     # construction: rank 0's buffer, step count and learning rate everywhere
