"""GPU diagnostics for the rasterizer: parity statistics for every test scene (no asserts) + kernel timings.
Writes gpurun_out/diag_raster.json.  Run on the GPU box: python tools/gpu_diag_raster.py"""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dwg_import  # noqa: E402,F401
from tests import raster_cases as rc  # noqa: E402

out = {"cases": [], "timing": []}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
CASES = [
    (10000, 256, 256, {}), (1, 64, 64, {}), (333, 250, 130, {}), (3000, 128, 128, dict(scale_mul=6.0)),
    (6000, 64, 64, dict(cluster=0.05, opacity_range=(0.01, 0.05))),
    (12000, 64, 64, dict(cluster=0.02, opacity_range=(0.004, 0.02))),
    (2000, 128, 128, dict(same_depth=True)),
]
for G, H, W, kw in CASES:
    rec = dict(G=G, H=H, W=W, kw=str(kw))
    try:
        sc = rc.make_scene(G, H, W, seed=G, **kw)
        ref = rc.oracle_forward(sc)
        o = rc.hip_render(sc, requires_grad=True)
        torch.cuda.synchronize()
        rec["fwd"] = rc.image_err_stats(o, ref)
        rec["K"] = int(ref["num_pairs"])
        rs = np.random.RandomState(3)
        wc = rs.randn(3, H, W).astype(np.float32); wd = rs.randn(H, W).astype(np.float32); wa = rs.randn(H, W).astype(np.float32)
        refb = rc.oracle_backward(sc, wc, wd, wa)
        loss = (o["color"] * torch.from_numpy(wc).cuda()).sum() + (o["depth"][0] * torch.from_numpy(wd).cuda()).sum() + (o["alpha"][0] * torch.from_numpy(wa).cuda()).sum()
        loss.backward()
        torch.cuda.synchronize()
        rec["bwd"] = {k: rc.grad_err(o["leaves"][k].grad.cpu().numpy(), refb[k]) for k in ("means3D", "means2D", "opacities", "colors", "scales", "rotations")}
    except Exception as e:  # noqa
        rec["error"] = traceback.format_exc()
    out["cases"].append(rec)
    print(json.dumps(rec)[:600], flush=True)

for G, H, W in [(10000, 256, 256), (50000, 512, 512), (100000, 512, 512), (300000, 1024, 1024)]:
    try:
        sc = rc.make_scene(G, H, W, seed=1)
        o = rc.hip_render(sc, requires_grad=True)
        wc = torch.randn(3, H, W, device="cuda")
        (o["color"] * wc).sum().backward()
        torch.cuda.synchronize()
        tf, tb = [], []
        for it in range(10):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            o = rc.hip_render(sc, requires_grad=True)
            e1.record()
            (o["color"] * wc).sum().backward()
            e2.record()
            torch.cuda.synchronize()
            tf.append(e0.elapsed_time(e1)); tb.append(e1.elapsed_time(e2))
        rec = dict(G=G, H=H, W=W, fwd_ms=float(np.median(tf)), bwd_ms=float(np.median(tb)), fwd_min=float(min(tf)), bwd_min=float(min(tb)))
    except Exception:
        rec = dict(G=G, H=H, W=W, error=traceback.format_exc())
    out["timing"].append(rec)
    print(json.dumps(rec), flush=True)

json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_raster.json"), "w"), indent=1)
