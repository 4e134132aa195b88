"""-m gpu: the multi-view step on the real kernels.  Two ranks share cuda:0 over gloo (a 1-GPU box); view v of the
step goes to rank v mod 2; every rank accumulates its views, ONE all-reduce of the flat gradient buffer, 1 / V folded into the fused Adam.
Asserted: replicas bit-identical after the steps, and equal (fp32 summation order, float atomics) to ONE process that accumulates the
same views; bench.py --gpus 2 starts its own ranks and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "8"
    return env


def _run_workers(prefix, world, views, steps, backend=None):
    worker = os.path.join(ROOT, "tests", "multiview_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, prefix, str(views), str(steps)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), worker, prefix, str(views), str(steps)]
    env = _env()
    if backend:
        env["DWG_WORKER_BACKEND"] = backend
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [torch.load("%s_rank%d.pt" % (prefix, k)) for k in range(world)]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_two_ranks_on_one_gpu_equal_one_process_accumulating_the_same_views(tmp_path):
    V, steps = 4, 2
    two = _run_workers(str(tmp_path / "two"), 2, V, steps)
    one = _run_workers(str(tmp_path / "one"), 1, V, steps)
    assert two[0]["views"] == [0, 2] and two[1]["views"] == [1, 3] and one[0]["views"] == [0, 1, 2, 3]
    assert two[0]["grad_scale"] == 0.25 and one[0]["grad_scale"] == 0.25
    # identical all-reduced gradient -> identical Adam update -> bit-identical replicas
    assert torch.equal(two[0]["grad"], two[1]["grad"])
    assert torch.equal(two[0]["flat"], two[1]["flat"])
    # vs one process: the same four views summed in another order (and float atomics inside each backward)
    assert _rel(two[0]["grad"], one[0]["grad"]) < 1e-4
    d = (two[0]["flat"] - one[0]["flat"]).abs()
    moved = (one[0]["flat"] - 0).abs() > 0
    # Adam's first steps move a parameter by ~lr whatever the gradient's size: an entry whose tiny gradient changes sign with the summation
    # order lands 2 lr apart.  Such entries must be rare; everything else agrees to rounding.
    assert float((d > 1e-5).float().mean()) < 2e-3, float((d > 1e-5).float().mean())
    assert bool(moved.any())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the real train_step on two ranks over RCCL (nccl backend)")
def test_two_ranks_on_two_gpus_over_rccl(tmp_path):
    """Round 6 (verdict round 5, next 7): whenever a lease has two GPUs, the REAL multi-rank step runs over RCCL / xGMI -- rank r on cuda:r,
    `nccl` backend with device_id, the sliced asynchronous all-reduce of trainer._reduce_and_step -- even when the driver's SCALE run is
    skipped.  Replicas bit-identical, equal to one process accumulating the same views, and the exchange step's duration is reported."""
    V, steps = 4, 3
    two = _run_workers(str(tmp_path / "rccl"), 2, V, steps, backend="nccl")
    one = _run_workers(str(tmp_path / "one"), 1, V, steps)
    assert [t["device"] for t in two] == ["cuda:0", "cuda:1"] and two[0]["backend"] == "nccl"
    assert two[0]["views"] == [0, 2] and two[1]["views"] == [1, 3]
    assert torch.equal(two[0]["grad"], two[1]["grad"]) and torch.equal(two[0]["flat"], two[1]["flat"])      # bit-identical replicas
    assert _rel(two[0]["grad"], one[0]["grad"]) < 1e-4
    assert two[0]["allreduce_ms"] is not None and two[0]["allreduce_ms"] > 0
    print("[rccl] two ranks on two GPUs: allreduce %.3f ms per step (flat gradient %d floats)" % (two[0]["allreduce_ms"], two[0]["grad"].numel()))
    path = os.path.join(ROOT, "gpurun_out", "rccl_two_ranks.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump({"world": 2, "backend": "nccl", "allreduce_ms_per_step": two[0]["allreduce_ms"], "grad_floats": int(two[0]["grad"].numel()),
               "replicas_bit_identical": True}, open(path, "w"))


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: two ranks (sharing the one GPU here), ONE JSON line with n_gpus = 2; c4 semantics with
    --views: one step = V views, value in steps/s, views/s beside it."""
    env = _env()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--config", "c4", "--views", "4", "--no-guidance",
           "--gaussians", "20000", "--res", "256", "--steps", "3", "--warmup", "1", "--headline-only"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["views_per_step"] == 4 and out["views_per_step_per_gpu"] == 2 and out["scaling"] == "strong"
    assert abs(out["views_per_s"] - 4 * out["value"]) < 1e-6 * out["views_per_s"]
    assert "shared_gpu" in out and out["steps"] == 3


def test_one_batched_guidance_call_equals_the_views_one_at_a_time():
    """The multi-view step in its two forms on the SAME three views (own camera / pose / condition image / RNG stream each): (a) one
    guidance call per view, gradients accumulated; (b) ONE call: VAE batch 3, ControlNet + UNet CFG batch 6.  fp32 plans at reduced width
    so that the comparison sees the batching, not bf16 rounding under CFG 50.  Also: the batched denoiser / VAE plans against their
    single-view twins on the same inputs."""
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import guidance, sd15, sds_step
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    res, V = 128, 3
    ucfg = sd15.UNetConfig(block_out_channels=(64, 128, 128, 128), cross_dim=64, cond_channels=(16, 32, 32, 64))
    vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
    usd = sd15.random_state_dict(sd15.unet_param_shapes(ucfg), seed=1)
    csd = sd15.random_state_dict(sd15.controlnet_param_shapes(ucfg), seed=2)
    vsd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
    g1 = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, usd, csd, vsd, image_hw=res, dtype="f32")
    gV = guidance.ControlNetScoreDistillation(dev, ucfg, vcfg, image_hw=res, dtype="f32", views=V, share_weights_with=g1)
    assert all(a is b for a, b in zip(gV.denoiser.weights, g1.denoiser.weights)) and gV.vae.weights is g1.vae.weights
    # ---- plan level: batched == per view
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(V, 4, res // 8, res // 8, generator=g).to(dev)
    text = torch.randn(2 * V, 77, 64, generator=g).to(dev)
    cond = torch.rand(V, 3, res, res, generator=g).to(dev)
    t = torch.tensor([100, 500, 900], device=dev)
    gV.denoiser.set_inputs(torch.cat([lat, lat]), t, text, cond)
    eps_b = gV.denoiser.run().clone()
    for v in range(V):
        g1.denoiser.set_inputs(torch.cat([lat[v:v + 1]] * 2), t[v:v + 1], torch.stack([text[v], text[V + v]]), cond[v:v + 1])
        e = g1.denoiser.run()
        assert _rel(eps_b[v], e[0]) < 2e-5 and _rel(eps_b[V + v], e[1]) < 2e-5, v
    img = torch.rand(V, 3, res, res, generator=g).to(dev)
    gm = torch.randn(V, 8, res // 8, res // 8, generator=g).to(dev)
    mom_b = gV.vae.encode(img).clone(); gi_b = gV.vae.backward(gm).clone()
    for v in range(V):
        m = g1.vae.encode(img[v:v + 1]); gi = g1.vae.backward(gm[v:v + 1])
        assert _rel(mom_b[v], m[0]) < 2e-5 and _rel(gi_b[v], gi[0]) < 2e-5, v
    # ---- step level
    flats = []
    for gd in (g1, gV):
        step = sds_step.SDSStep(n_gaussians=6000, res=res, device=dev, guidance=True, guidance_obj=gd, views=V, async_pair_count=False, iters=1000)
        assert step.my_views == [0, 1, 2]
        if gd is gV:
            # the batched step renders its views concurrently on their own streams once the constant canonical-pose caches exist (its first
            # step would fill them on one stream): fill them here, so that THIS step is the multi-stream one
            from dreamwaltz_g_amd import synth
            with torch.no_grad():
                step.avatar.animate(synth.random_smpl_inputs(seed=99, device=dev))
            torch.cuda.synchronize()
            step.trainer._views_warm = True
        step.run()
        if gd is gV:
            assert len(step.trainer._view_streams) == V and step.scene.renderer.per_stream_pair_states
        torch.cuda.synchronize()
        b = step.optimizers.buffers
        flats.append((b.grad.clone(), b.flat.clone(), step.optimizers["avatar"].grad_scale))
    (ga, fa, sa), (gb, fb, sb) = flats
    assert sa == sb == 1.0 / V
    assert float(ga.abs().sum()) > 0 and _rel(gb, ga) < 2e-3, _rel(gb, ga)
    d = (fa - fb).abs()
    assert float((d > 1e-5).float().mean()) < 5e-3


@pytest.mark.slow
def test_c4_workload_on_one_gpu_batched_call_equals_eight_sequential_calls():
    """BASELINE config c4 at its own size as far as ONE GPU allows: 100 000 Gaussians, 512^2, V = 8 views per step, full-width SD-1.5 +
    ControlNet + VAE at the headline precision (f32x), one step.  The rank's share as ONE guidance call (VAE batch 8, ControlNet + UNet CFG
    batch 16, per-view render chains on their own streams) against the same 8 views one call at a time (gradients accumulated): the flat
    gradient buffer -- the all-reduce operand of the multi-GPU step -- agrees to rel-L2 <= 2e-3 (the rasterizer / encoder float atomics;
    the guidance itself is batch-invariant to 2e-5, see the test above), and so do the parameters after the Adam step."""
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import guidance, sds_step, synth
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    V, res, G = 8, 512, 100000
    g1 = guidance.ControlNetScoreDistillation(dev, image_hw=res, seed=0, dtype="f32x")
    gV = guidance.ControlNetScoreDistillation(dev, image_hw=res, dtype="f32x", views=V, share_weights_with=g1)
    flats = []
    for gd in (g1, gV):
        step = sds_step.SDSStep(n_gaussians=G, res=res, device=dev, guidance=True, guidance_obj=gd, views=V, async_pair_count=True, iters=1000)
        assert step.my_views == list(range(V)) and step.G == G
        if gd is gV:
            with torch.no_grad():
                step.avatar.animate(synth.random_smpl_inputs(seed=99, device=dev))      # constant canonical-pose caches: the step is multi-stream
            torch.cuda.synchronize()
            step.trainer._views_warm = True
        step.run()
        torch.cuda.synchronize()
        if gd is gV:
            assert len(step.trainer._view_streams) == V
        b = step.optimizers.buffers
        flats.append((b.grad.clone(), b.flat.clone(), step.optimizers["avatar"].grad_scale, step.trainer.redone_frames))
    (ga, fa, sa, ra), (gb, fb, sb, rb) = flats
    assert sa == sb == 1.0 / V
    e = _rel(gb, ga)
    print("[parity] c4_size_batched_vs_sequential rel_l2 %.3e (redone frames %d / %d)" % (e, ra, rb))
    assert float(ga.abs().sum()) > 0 and e < 2e-3, e
    d = (fa - fb).abs()
    assert float((d > 1e-5).float().mean()) < 5e-3
