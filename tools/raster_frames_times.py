"""Forward rasterizer time PER FRAME when F frames share one launch chain (rasterizer.rasterize_frames) against F single-frame calls, at the
playback sizes (c1: 10 k / 256^2, c5: 300 k / 1024^2; random-init scenes of tests/raster_cases.py, each frame its own Gaussians).
Run on the GPU box:  python tools/raster_frames_times.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd import rasterizer as R  # noqa: E402
from tests import raster_cases as rc  # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for G, H, W in [(10000, 256, 256), (50000, 512, 512), (100000, 512, 512), (300000, 1024, 1024)]:
    Fmax = 8
    scs = [rc.make_scene(G, H, W, seed=s) for s in range(Fmax)]
    dev = "cuda"
    st = lambda k, F: torch.stack([scs[f][k].to(dev) for f in range(F)]).contiguous()        # noqa: E731
    cam1 = torch.cat([scs[0]["viewmatrix"].reshape(-1), scs[0]["projmatrix"].reshape(-1), scs[0]["campos"].reshape(-1)]).to(dev)
    line = "G=%d %dx%d:" % (G, H, W)
    with torch.inference_mode():
        for F in (1, 2, 4, 8):
            m, o, c, s, r = st("means3D", F), st("opacities", F), st("colors", F), st("scales", F), st("rotations", F)
            cams = cam1[None].expand(F, -1).contiguous()
            info = {}

            def run():
                out = R.rasterize_frames(m, o, colors_precomp=c, scales=s, rotations=r, cameras=cams, image_height=H, image_width=W,
                                         tanfovx=scs[0]["tanfovx"], tanfovy=scs[0]["tanfovy"], bg=scs[0]["bg"].to(dev),
                                         pair_capacity=info.get("cap"))
                info["cap"] = out[4]["capacity"]
            run()
            t = timed(run)
            line += "  F=%d %.1f us/frame" % (F, t / F)
    print(line, flush=True)
