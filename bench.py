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
    """Oracle (CPU restatement, single core) timed on a bounded sample of the same workload: LBS blend + tile
    rasterizer forward+backward (the reference has no CPU diffusion path: BASELINE.md section 4)."""
    import numpy as np
    from oracle import animate as oa
    from tests import raster_cases as rc
    G = min(args.gaussians, 20000)
    sc = rc.make_scene(G, args.res, args.res, seed=0)
    t0 = time.perf_counter()
    g = torch.Generator().manual_seed(0)
    A = torch.eye(4).repeat(55, 1, 1)
    A[:, :3, :3] = oa.batch_rodrigues(torch.randn(55, 3, generator=g) * 0.3)
    w = torch.softmax(torch.randn(G, 55, generator=g), -1)
    torch.set_num_threads(1)
    p = oa.transform_points(A, sc["means3D"], weights=w)
    q = oa.transform_quaternions_flip(A, sc["rotations"], w)
    sc2 = dict(sc); sc2["means3D"] = p; sc2["rotations"] = q
    rc.oracle_forward(sc2)
    wc = np.random.RandomState(0).randn(3, args.res, args.res).astype(np.float32)
    rc.oracle_backward(sc2, wc, None, None, dtype=np.float32)
    dt = time.perf_counter() - t0
    # per-Gaussian extrapolation to the full workload size (flagged in `sample`)
    scale = args.gaussians / G
    return {"value": 1.0 / (dt * scale), "unit": "steps/s (LBS + rasterizer fwd+bwd only, no diffusion)", "cores": 1,
            "kind": "port",
            "sample": "1 pass: LBS blend + tile raster fwd+bwd of %d Gaussians @%dx%d on 1 core (%.2f s), scaled x%.1f "
                      "per-Gaussian to %d" % (G, args.res, args.res, dt, scale, args.gaussians)}


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
