"""Which torch ops (and how many device kernels) each phase of the SDS step still launches outside the HIP library."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from torch.profiler import profile, ProfilerActivity
from dreamwaltz_g_amd import sds_step, synth, rasterizer

torch.cuda.set_stream(torch.cuda.Stream())
st = sds_step.SDSStep(n_gaussians=100000, res=512, device="cuda")
for _ in range(2):
    st.run()
torch.cuda.synchronize()


def phase(name, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    ops = collections.Counter()
    nk = 0
    for e in prof.events():
        if str(e.device_type).endswith("CUDA"):
            nk += 1
        elif e.name.startswith("aten::") and e.cpu_parent is None or (e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::") and e.name.startswith("aten::")):
            ops[e.name] += 1
    print("== %s: %d device kernels (incl. HIP-library launches); top-level aten ops: %d" % (name, nk, sum(ops.values())))
    print("   " + ", ".join("%s x%d" % kv for kv in ops.most_common(25)))
    return out


[o.zero_grad() for o in st.opt.values()]
pose = phase("pose", lambda: synth.random_smpl_inputs(seed=5, device=st.device))
g = phase("animate", lambda: st.avatar.animate(pose))
out = phase("render", lambda: st.renderer.render(st.data, g))
res = phase("guidance", lambda: st.guidance(out["image"].permute(0, 3, 1, 2), dict(st.text, text=st.text["pos"]), cond_inputs=st.data["cond_images"]))
phase("backward", lambda: (res["diffusion_loss"] * 1.0).backward())
phase("adam", lambda: [o.step() for o in st.opt.values()])
