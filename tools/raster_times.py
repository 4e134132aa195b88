"""Per-kernel times of the rasterizer (library profiler: HIP events on the launch stream) at the BASELINE sizes, on the random-init
scenes of tests/raster_cases.py.  Run on the GPU box:  python tools/raster_times.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd import _lib  # noqa: E402
from tests import raster_cases as rc  # noqa: E402

out = []
for G, H, W in [(10000, 256, 256), (50000, 512, 512), (100000, 512, 512), (300000, 1024, 1024)]:
    sc = rc.make_scene(G, H, W, seed=1)
    wc = torch.randn(3, H, W, device="cuda")
    for _ in range(3):
        o = rc.hip_render(sc, requires_grad=True)
        (o["color"] * wc).sum().backward()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    N = 10
    for _ in range(N):
        o = rc.hip_render(sc, requires_grad=True)
        (o["color"] * wc).sum().backward()
    torch.cuda.synchronize()
    t = {k: v[1] / N * 1e3 for k, v in _lib.prof_table().items() if k.startswith("raster_")}
    _lib.prof_enable(False)
    fwd = sum(v for k, v in t.items() if not k.endswith("_bwd")); bwd = sum(v for k, v in t.items() if k.endswith("_bwd"))
    rec = dict(G=G, H=H, W=W, fwd_us=round(fwd, 1), bwd_us=round(bwd, 1), kernels_us={k: round(v, 1) for k, v in sorted(t.items(), key=lambda kv: -kv[1])})
    out.append(rec)
    print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
