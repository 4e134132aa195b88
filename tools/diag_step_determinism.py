"""Which parameter gradients of ONE avatar-side training step differ between two identical runs (bitwise)?  python tools/diag_step_determinism.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sds_step

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
runs = []
for r in range(2):
    s = sds_step.SDSStep(n_gaussians=int(sys.argv[1]) if len(sys.argv) > 1 else 8000, res=128, device=dev, guidance=False, async_pair_count=True, iters=1000)
    out = s.run()
    torch.cuda.synchronize()
    named = {}
    for name, p in s.scene.named_parameters():
        if p.grad is not None:
            named[name] = p.grad.detach().clone()
    runs.append((named, out[1]["image"].detach().clone(), s.optimizers.buffers.flat.detach().clone()))
a, b = runs
print("image equal:", torch.equal(a[1], b[1]), " params after step equal:", torch.equal(a[2], b[2]))
for k in a[0]:
    ga, gb = a[0][k], b[0][k]
    eq = torch.equal(ga, gb)
    d = float((ga - gb).abs().max())
    print("%-60s %-10s equal=%s maxdiff=%.3e nnz=%d" % (k, tuple(ga.shape), eq, d, int((ga != gb).sum())))
