"""bench.py -- headline benchmark of the SDS render-and-distill hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)
    python bench.py --config c2|c5                         (the other single-GPU BASELINE.json configurations; not the headline)

Prints ONE JSON line (rank 0) following the driver contract: metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline.
A "step" is one pass of the hot path over one batch of synthetic input; see
dreamwaltz-g_amd/sds_step.py for exactly which stages run.  Inputs are resident in HBM before the timed region.
  c3 (default)  full SDS step, 100k Gaussians, 512^2, SD-1.5 + ControlNet      -> SDS steps/s            (the headline)
  c2            50k Gaussians + LBS/encoder/MLPs, 512^2 raster fwd+bwd, no guidance -> steps/s + raster ms / Mpix/s / GB/s
  c5            300k Gaussians, 1024^2, per-frame animate + raster forward (inference) -> frames/s
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
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 20 (config c3); 200 for c2 / c5, whose steps take ~2 ms")
    ap.add_argument("--warmup", type=int, default=None, help="default: 3 (c3); 20 for c2 / c5")
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default="c3")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-guidance", action="store_true", help="raster+LBS sub-path only (not the headline workload)")
    ap.add_argument("--frame-graph", action="store_true", help="config c5: replay each frame as one captured hipGraph (player.GraphedAnimation)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel eagerly (no hipGraph replay of the denoiser/VAE plans)")
    ap.add_argument("--no-gpu-condition", action="store_true", help="fixed condition image instead of the per-step GPU OpenPose image of the posed body")
    ap.add_argument("--sync-pairs", action="store_true", help="exact pair-buffer sizing through a 16-byte read-back per frame")
    args = ap.parse_args()
    short = args.config in ("c2", "c5")          # a 20-step window of 2-ms steps is 40 ms: too short to time a host-fed loop
    if args.steps is None:
        args.steps = 200 if short else 20
    if args.warmup is None:
        args.warmup = 20 if short else 3
    return args


def cpu_baseline(args, G, res):
    """The oracle (CPU restatement of the reference's PyTorch LBS / encoder / MLP path + the tile rasterizer) timed on ALL host cores
    (torch.set_num_threads(os.cpu_count())) on the SAME kind of workload the GPU step renders: 90 % free
    Gaussians with 4 non-zero skinning weights per row + 10 % mesh-bound Gaussians, `animate` forward + backward through autograd and
    the rasterizer forward + backward.  The reference has no CPU diffusion path, so the diffusion half of the step
    has no CPU counterpart and is NOT in this number.  (The C rasterizer oracle is single-threaded; the torch part uses every core.)"""
    import numpy as np
    from oracle import animate as oa
    from tests import raster_cases as rc
    # every host core up to 32: beyond that the oracle's element-wise torch ops on 1e5-row tensors only get slower (measured on the
    # 256-core GPU box: the same pass takes minutes with 256 OpenMP threads)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    M = (G // 10) // 6 * 6
    N = G - M
    body = oa.SyntheticBody(seed=0)
    nets = oa.init_avatar_networks(seed=0)
    g = torch.Generator().manual_seed(1)
    logits = torch.full((N, 55), -1e9)
    logits.scatter_(1, torch.randint(0, 55, (N, 4), generator=g), torch.randn(N, 4, generator=g))
    params = dict(_positions=((torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([0.4, 0.9, 0.2])).requires_grad_(True),
                  _scales=torch.log(torch.rand(N, 3, generator=g) * 0.018 + 0.002).requires_grad_(True),
                  _quaternions=torch.randn(N, 4, generator=g).requires_grad_(True), _lbs_weights=torch.softmax(logits, -1))
    nets["table"].requires_grad_(True)
    Vp, Fp = 1200, M // 6
    vi = torch.randperm(body.V, generator=g)[:Vp]
    tri = torch.stack([torch.randint(0, Vp, (Fp,), generator=g) for _ in range(3)], 1)
    tri[:, 1] = (tri[:, 0] + 1 + tri[:, 1] % (Vp // 2 - 1)) % Vp; tri[:, 2] = (tri[:, 0] + Vp // 2 + tri[:, 2] % (Vp // 2)) % Vp    # distinct corners
    base = torch.tensor([[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3], [1 / 6, 5 / 12, 5 / 12],
                         [5 / 12, 1 / 6, 5 / 12], [5 / 12, 5 / 12, 1 / 6]])
    mesh = dict(vertex_indices=vi, triangles=tri, vertex_coords=body.v_template[vi], bary=base.expand(Fp, -1, -1).clone().requires_grad_(True),
                scales=torch.ones(Fp * 6, 3).requires_grad_(True)) if M > 0 else None
    cnl = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
               right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
    obs = oa.random_smpl_inputs(seed=3)
    sc = rc.make_scene(G, res, res, seed=0)
    wc = np.random.RandomState(0).randn(3, res, res).astype(np.float32)
    # repeated passes (median) until about 10 s of CPU work have been spent, at most 5
    ta, tr, spent = [], [], 0.0
    while len(ta) < 5 and (spent < 10.0 or len(ta) < 2):
        for v in list(params.values()) + [nets["table"]] + ([mesh["bary"], mesh["scales"]] if mesh else []):
            v.grad = None
        t0 = time.perf_counter()
        out = oa.animate(params, nets, body, obs, cnl, mesh=mesh)
        sum(v.sum() for v in out.values()).backward()
        t1 = time.perf_counter()
        rc.oracle_forward(sc)
        rc.oracle_backward(sc, wc, None, None, dtype=np.float32)
        t2 = time.perf_counter()
        ta.append(t1 - t0); tr.append(t2 - t1); spent += t2 - t0
    t_an, t_ra = float(np.median(ta)), float(np.median(tr))
    return {"value": 1.0 / (t_an + t_ra), "unit": "steps/s (animate + rasterizer, fwd+bwd, no diffusion)", "cores": cores,
            "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": "median of %d passes (%.0f s of CPU work, %d torch threads; the C rasterizer oracle is one thread): oracle animate "
                      "fwd+bwd %.2f s (%d free Gaussians with 4 non-zero skinning weights + %d mesh-bound) + tile raster fwd+bwd %.2f s "
                      "(%d Gaussians @%dx%d)" % (len(ta), spent, cores, t_an, N, M, t_ra, G, res, res)}


def raster_report(prof, G, Kref, K, P, steps, pmc=False):
    """Rasterizer against the HBM roofline with SURVEY 8d's byte formula (K = reference tile-pair count; sort traffic not counted)."""
    rf = sum(v[1] for k, v in prof.items() if k.startswith("raster_") and not k.endswith("_bwd")) / steps
    rb = sum(v[1] for k, v in prof.items() if k.startswith("raster_") and k.endswith("_bwd")) / steps
    out = {}
    if rf > 0:
        b = 56 * G + 44 * Kref + 20 * P
        out["raster_forward"] = {"bytes": b, "pairs_reference": Kref, "pairs_after_exact_culling": K, "ms": rf, "mpix_per_s": P / (rf * 1e-3) / 1e6,
                                 "achieved_GBps": b / (rf * 1e-3) / 1e9, "frac_of_hbm_peak": b / (rf * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if rb > 0:
        b = 80 * Kref + 20 * P + 152 * G
        out["raster_backward"] = {"bytes": b, "ms": rb, "achieved_GBps": b / (rb * 1e-3) / 1e9,
                                  "frac_of_hbm_peak": b / (rb * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if pmc and os.path.exists(TRAFFIC_JSON):       # HBM bytes per frame from the PMC passes (taken on the default c3 workload only)
        tk = json.load(open(TRAFFIC_JSON))["kernels"]
        fwd = ("k_preprocess", "k_scan_tiles", "k_scatter", "k_tile_sort", "k_render_fwd", "k_camera_setup")
        bwd = ("k_render_bwd", "k_preprocess_bwd")
        for key, names in (("raster_forward", fwd), ("raster_backward", bwd)):
            if key in out:
                t = sum(v["hbm_bytes_per_launch"] for k, v in tk.items() if k.split("<")[0] in names)
                out[key]["traffic"] = t
                out[key]["traffic_over_algorithmic"] = t / out[key]["bytes"]
    return out


def roofline(prof, prof_sym, prof_steps, G, Kref, K, P):
    """Roofline entry for the dominant kernel = the kernel SYMBOL (as rocprofv3 --kernel-trace names it) with the largest total time
    in the profiled region; achieved = algorithmic work per launch / average launch duration (HIP events on the launch stream)."""
    out = {}
    if prof_sym:
        name, (count, total_ms, work) = max(prof_sym.items(), key=lambda kv: kv[1][1])
        avg_ms = total_ms / max(1, count)
        if work > 0:
            ach = work / count / (avg_ms * 1e-3) / 1e12
            out = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_PEAK_TFLOPS,
                   "traffic": None, "avg_launch_ms": avg_ms, "launches": count, "flops_per_launch": work / count}
        else:
            out = {"kernel": name, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                   "avg_launch_ms": avg_ms, "launches": count}
        out["mfma_kernels"] = {k: {"launches": c, "avg_launch_ms": ms / c, "tflops": w / (ms * 1e-3) / 1e12,
                                   "frac": w / (ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS}
                               for k, (c, ms, w) in sorted(prof_sym.items(), key=lambda kv: -kv[1][1]) if w > 0 and ms > 0}
        tw = sum(w for (_, _, w) in prof_sym.values()); tms = sum(ms for (_, ms, w) in prof_sym.values() if w > 0)
        if tms > 0:
            out["mfma_all"] = {"tflops": tw / (tms * 1e-3) / 1e12, "frac": tw / (tms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS,
                               "flops_per_step": tw / prof_steps}
        if os.path.exists(TRAFFIC_JSON) and out.get("kernel"):
            tk = json.load(open(TRAFFIC_JSON))["kernels"].get(out["kernel"].replace(" ", ""))
            if tk:
                out["traffic"] = tk["hbm_bytes_per_launch"]
                out["traffic_source"] = "profiles/r02_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections)"
    out.update(raster_report(prof, G, Kref, K, P, prof_steps, pmc=(G == 100000 and P == 512 * 512)))
    return out


def run_c5(args, dev):
    """Config c5: 300k-Gaussian avatar, per-frame animate (LBS + encoder + MLPs + mesh binding) + raster forward at 1024^2, inference."""
    G, res = args.gaussians or 300000, args.res or 1024
    from dreamwaltz_g_amd import camera, configs, scene as sc, synth
    cfg = configs.TrainConfig(); cfg.device = str(dev); cfg.render.bg_color = (0.5, 0.5, 0.5)
    avatar, N, M = sds_step.build_synthetic_avatar(G, dev, seed=0)
    scene = sc.Scene(cfg, avatar, async_pair_count=not args.sync_pairs).to(dev).eval()
    data = camera.make_camera(radius=2.0, azimuth=0.0, elevation=80.0, fovy=55.0, height=res, width=res, device=dev)
    poses = [synth.random_smpl_inputs(seed=i, device=dev) for i in range(240)]        # 240 pose frames

    player = None
    if args.frame_graph:        # the frame as ONE captured graph replayed per pose (player.GraphedAnimation); measured: no gain, the frame is GPU-bound
        from dreamwaltz_g_amd import player as pl
        player = pl.GraphedAnimation(scene, data, poses[0], warmup_poses=poses[:24])

    def frame(i):
        if player is not None:
            return player.replay(poses[i % 240])
        with torch.inference_mode():
            return scene.forward(data, smpl_observed_inputs=poses[i % 240], use_densifier=False, bg_mode=None)
    for i in range(args.warmup):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graphed = player is not None
    if graphed:                 # per-kernel timers need eager launches: same kernels, same inputs, after the timed region
        assert player.check(), "a replayed frame was truncated by the frozen pair capacity"
        player.close(); player = None
    _lib.prof_enable(True)
    ps = min(args.steps, 5)
    for i in range(ps):
        frame(i)
    torch.cuda.synchronize()
    prof = _lib.prof_table(); _lib.prof_enable(False)
    K, Kref = scene.renderer.last_rasterizer.last_num_pairs
    out = {"metric": "AIST++-style animation inference (config c5): frames/s, %dk-Gaussian avatar, per-frame LBS+raster at %d^2" % (G // 1000, res),
           "value": args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "c5: animate (LBS x2, grid encoder, MLPs, %d free + %d mesh-bound Gaussians) + raster forward %dx%d, "
                                  "inference_mode, 240 seeded random pose frames" % (N, M, res, res), "gaussians": G, "resolution": res},
           "roofline": raster_report(prof, G, Kref, K, res * res, ps),
           "launch_mode": "one hipGraph per frame (player.GraphedAnimation); kernel timers from eager frames after the timed region" if graphed else "eager",
           "kernel_ms_per_step": {k: round(v[1] / ps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:20]}}
    print(json.dumps(out))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (HIP kernels, no CPU fallback)"
    # Functional test of the N > 1 flow on a 1-GPU box: DWG_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and DWG_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device).  Never set by the driver; numbers from that mode are not a measurement.
    if os.environ.get("DWG_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DWG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # one real (non-default) HIP stream for the whole step: graph-safe
    if args.config == "c5":
        return run_c5(args, dev)
    guidance = not args.no_guidance and args.config == "c3"
    G = args.gaussians or (50000 if args.config == "c2" else 100000)
    res = args.res or 512
    step = sds_step.SDSStep(n_gaussians=G, res=res, device=dev, rank=rank, world=world, guidance=guidance, dist=dist,
                            async_pair_count=not args.sync_pairs, gpu_condition=not args.no_gpu_condition)
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
        dist.destroy_process_group()
        return
    info = step.describe()
    ms = dt / args.steps * 1e3
    views_per_step = world  # one view per rank per step (weak scaling)
    K, Kref = step.num_pairs
    headline = args.config == "c3" and guidance
    out = {
        "metric": "SDS steps/sec @512^2, 100k Gaussians, SD1.5+ControlNet; raster Mpix/s vs HBM roofline" if headline else
                  "config %s sub-path (NOT the headline): animate + raster fwd+bwd + Adam steps/s, %dk Gaussians @%d^2, no guidance" % (args.config, G // 1000, res),
        "value": views_per_step * args.steps / dt, "unit": "SDS steps/s (one view each; whole job)" if headline else "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": info["dtype"], "data": "synthetic",
        "config": info["config"],
    }
    out["roofline"] = roofline(prof, prof_sym, prof_steps, step.G, Kref, K, res * res)
    rf = out["roofline"].get("raster_forward")
    out["raster_mpix_per_s"] = rf["mpix_per_s"] if rf else None      # the rasterizer's own forward rate (pixels / forward-chain time)
    out["redone_frames"] = step.trainer.redone_frames
    out["kernel_ms_per_step"] = {k: round(v[1] / prof_steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:40]}
    out["launch_mode"] = "eager" if args.eager else "hipGraph replay of denoiser/VAE plans; kernel timers from an eager replay after the timed region"
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, G, res)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
