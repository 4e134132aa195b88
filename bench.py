"""bench.py -- headline benchmark of the SDS render-and-distill hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0) following the driver contract: metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline.
A "step" is one pass of the hot path over one batch of synthetic input; see
dreamwaltz-g_amd/sds_step.py for exactly which stages run.  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd import _lib  # noqa: E402
from dreamwaltz_g_amd import sds_step  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=100000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-guidance", action="store_true", help="raster+LBS sub-path only (not the headline workload)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel eagerly (no hipGraph replay of the denoiser/VAE plans)")
    return ap.parse_args()


def cpu_baseline(args, workload):
    """Oracle (CPU restatement, ONE core) timed on a bounded sample of the same workload: `animate` (LBS x2 + grid encoder +
    both MLPs, forward + backward through autograd) and the tile rasterizer forward + backward.  The reference has no CPU
    diffusion path, so the diffusion half of the step has no CPU counterpart and is NOT in this number."""
    import numpy as np
    from oracle import animate as oa
    from tests import raster_cases as rc
    torch.set_num_threads(1)
    G = min(args.gaussians, 100000)      # the full headline size (about 8 s on the GPU box host, 20 s on a slow core)
    body = oa.SyntheticBody(seed=0)
    nets = oa.init_avatar_networks(seed=0)
    g = torch.Generator().manual_seed(1)
    params = dict(_positions=((torch.rand(G, 3, generator=g) * 2 - 1) * torch.tensor([0.4, 0.9, 0.2])).requires_grad_(True),
                  _scales=torch.log(torch.rand(G, 3, generator=g) * 0.018 + 0.002).requires_grad_(True),
                  _quaternions=torch.randn(G, 4, generator=g).requires_grad_(True),
                  _lbs_weights=torch.softmax(torch.randn(G, 55, generator=g), -1))
    nets["table"].requires_grad_(True)
    cnl = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
               right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
    obs = oa.random_smpl_inputs(seed=3)
    sc = rc.make_scene(G, args.res, args.res, seed=0)
    wc = np.random.RandomState(0).randn(3, args.res, args.res).astype(np.float32)
    # repeated passes (median) until about 10 s of CPU work have been spent, at most 5
    ta, tr, spent = [], [], 0.0
    while len(ta) < 5 and (spent < 10.0 or len(ta) < 2):
        for v in list(params.values()) + [nets["table"]]:
            v.grad = None
        t0 = time.perf_counter()
        out = oa.animate(params, nets, body, obs, cnl)
        sum(v.sum() for v in out.values()).backward()
        t1 = time.perf_counter()
        rc.oracle_forward(sc)
        rc.oracle_backward(sc, wc, None, None, dtype=np.float32)
        t2 = time.perf_counter()
        ta.append(t1 - t0); tr.append(t2 - t1); spent += t2 - t0
    t0, t1 = 0.0, float(np.median(ta))
    t2 = t1 + float(np.median(tr))
    dt = t2 - t0
    scale = args.gaussians / G          # per-Gaussian extrapolation to the full workload size (flagged in `sample`)
    return {"value": 1.0 / (dt * scale), "unit": "steps/s (animate + rasterizer, fwd+bwd, no diffusion)", "cores": 1,
            "kind": "port",
            "sample": "median of %d passes on 1 core (%.0f s of CPU work): oracle animate fwd+bwd (%.1f s) + tile raster fwd+bwd "
                      "(%.1f s) of %d Gaussians @%dx%d, scaled x%.1f per-Gaussian to %d"
                      % (len(ta), spent, t1 - t0, t2 - t1, G, args.res, args.res, scale, args.gaussians)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (HIP kernels, no CPU fallback)"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # one real (non-default) HIP stream for the whole step: graph-safe
    step = sds_step.SDSStep(n_gaussians=args.gaussians, res=args.res, device=dev, rank=rank, world=world,
                            guidance=not args.no_guidance, dist=dist)
    if not args.eager:
        step.capture_graphs()       # denoiser / VAE plans replay as hipGraphs (identical kernels, one launch each)
    for _ in range(args.warmup):
        step.run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if args.eager:
        _lib.prof_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof_steps = args.steps
    if not args.eager:
        # Per-kernel durations (HIP events on the launch stream) cannot be bracketed inside a graph replay: the same steps
        # are replayed eagerly right after the timed region with the event timers on (same kernels, same inputs).
        step.set_use_graphs(False)
        prof_steps = min(args.steps, 3)
        _lib.prof_enable(True)
        for _ in range(prof_steps):
            step.run()
        torch.cuda.synchronize()
    prof = _lib.prof_table()
    prof_sym = _lib.prof_symbols()
    _lib.prof_enable(False)
    if rank != 0:
        return
    info = step.describe()
    ms = dt / args.steps * 1e3
    views_per_step = world  # one view per rank per step (weak scaling)
    out = {
        "metric": "SDS steps/sec @512^2, 100k Gaussians, SD1.5+ControlNet; raster Mpix/s vs HBM roofline",
        "value": views_per_step * args.steps / dt, "unit": "SDS steps/s (one view each; whole job)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": info["dtype"], "data": "synthetic",
        "config": info["config"],
    }
    out["roofline"] = step.roofline(prof, HBM_PEAK_GBS, BF16_PEAK_TFLOPS, prof_sym)
    # HBM traffic of the dominant kernel per launch: PMC counters need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE
    # separately), so the table is produced by tools/pmc_traffic.py from those passes and committed under profiles/.
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if out["roofline"] and os.path.exists(tpath):
        tk = json.load(open(tpath))["kernels"].get(out["roofline"]["kernel"].replace(" ", ""))
        if tk:
            out["roofline"]["traffic"] = tk["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections)"
    out["raster_mpix_per_s"] = args.res * args.res * views_per_step * args.steps / dt / 1e6
    out["kernel_ms_per_step"] = {k: round(v[1] / prof_steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:40]}
    out["launch_mode"] = "eager" if args.eager else "hipGraph replay of denoiser/VAE plans; kernel timers from an eager replay after the timed region"
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, info)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
