"""GPU time per phase of the SDS step (torch events on the step's stream, graph replay on), averaged over steps."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sds_step, synth, rasterizer, guidance as gd

torch.cuda.set_stream(torch.cuda.Stream())
st = sds_step.SDSStep(n_gaussians=100000, res=512, device="cuda")
st.capture_graphs()
for _ in range(3):
    st.run()
torch.cuda.synchronize()
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))


# instrument the guidance internals
g = st.guidance
orig_enc, orig_pred = g.encode_images, g._predict
def enc(*a, **k):
    mark("raster_fwd+glue"); r = orig_enc(*a, **k); mark("vae_fwd"); return r
def pred(*a, **k):
    mark("sds_glue"); r = orig_pred(*a, **k); mark("denoiser"); return r
g.encode_images, g._predict = enc, pred
acc = collections.OrderedDict()
N = 10
for it in range(N):
    marks.clear()
    mark("start")
    [o.zero_grad() for o in st.opt.values()]
    pose = synth.random_smpl_inputs(seed=it, device=st.device)
    gs = st.avatar.animate(pose); mark("animate_fwd")
    out = st.renderer.render(st.data, gs)
    res = st.guidance(out["image"].permute(0, 3, 1, 2), dict(st.text, text=st.text["pos"]), cond_inputs=st.data["cond_images"]); mark("sds_tail")
    (res["diffusion_loss"] * 1.0).backward(); mark("backward(vae_bwd+raster_bwd+animate_bwd)")
    [o.step() for o in st.opt.values()]; mark("adam")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = 0
for k, v in acc.items():
    print("%-45s %7.3f ms" % (k, v / N)); tot += v / N
print("total %.3f" % tot)
