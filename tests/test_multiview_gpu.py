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


def _run_workers(prefix, world, views, steps):
    worker = os.path.join(ROOT, "tests", "multiview_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, prefix, str(views), str(steps)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), worker, prefix, str(views), str(steps)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
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
