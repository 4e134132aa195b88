"""Every GEMM / conv descriptor of the denoiser (and VAE) plans timed in plan order (HIP events around each launch, weights as cold as in the
step), grouped by shape: where the MFMA time of a step goes, per layer shape.   python tools/plan_gemm_shapes.py [bf16|f16] [out.json]"""
import collections, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import guidance as gd, gemm

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.cuda.set_stream(torch.cuda.Stream())
g = gd.ControlNetScoreDistillation(torch.device("cuda"), image_hw=512, seed=0, dtype=dt)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for pname, plan in (("denoiser", g.denoiser.plan), ("vae_fwd", g.vae.fwd), ("vae_bwd", g.vae.bwd)):
    descs = [d for d in plan.keep if isinstance(d, gemm.GemmDesc)]
    for _ in range(2):
        plan.run_eager()
    torch.cuda.synchronize()
    REP = 5
    acc = [0.0] * len(descs)
    for _ in range(REP):
        evs = []
        for d in descs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gemm.run_desc(d, st); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs):
            acc[i] += e0.elapsed_time(e1) / REP
    groups = collections.OrderedDict()
    for d, ms in zip(descs, acc):
        conv = "conv%dx%d s%d %dx%d" % (d.conv_kh, d.conv_kw, d.conv_stride, d.conv_hout, d.conv_wout) if d.conv_enabled else "gemm"
        key = (d.name.decode() if d.name else "", conv, d.M, d.N, d.K, d.batch1 * d.batch2, d.act, bool(d.A2))
        gq = groups.setdefault(key, [0, 0.0])
        gq[0] += 1; gq[1] += ms
    tot = sum(acc)
    print("== %s (%s): %d gemm launches, %.3f ms" % (pname, dt, len(descs), tot))
    rows = []
    for key, (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        name, conv, M, N, K, b, act, cat = key
        fl = 2.0 * M * N * K * b
        rows.append(dict(name=name, kind=conv, M=M, N=N, K=K, batch=b, act=act, concat=cat, launches=n, ms=ms, us_each=ms / n * 1e3, tflops=fl * n / ms / 1e9))
    for r in rows[:28]:
        print("  %-18s %-22s M=%-6d N=%-5d K=%-6d b=%-3d x%-3d %7.3f ms  %6.1f us  %6.0f TF/s%s" % (
            r["name"][:18], r["kind"], r["M"], r["N"], r["K"], r["batch"], r["launches"], r["ms"], r["us_each"], r["tflops"], " +cat" if r["concat"] else ""))
    out[pname] = rows
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
