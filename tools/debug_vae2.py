import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15
# poison the caching allocator with NaNs so that any read-before-write shows up deterministically
junk = [torch.full((1 << 24,), float("nan"), device="cuda") for _ in range(8)]
del junk
vcfg = sd15.VAEConfig(block_out_channels=(32, 64, 64, 64))
sd = sd15.random_state_dict(sd15.vae_encoder_param_shapes(vcfg), seed=3)
plan = sd15.VAEEncoderPlan(vcfg, sd, torch.device("cuda"), image_hw=128)
img = torch.rand(1, 3, 128, 128)
plan.x[..., :3].copy_((img * 2 - 1).permute(0, 2, 3, 1).cuda())
s = None
import ctypes
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def scan(pl, name):
    outs = {}
    for i, op in enumerate(pl.ops):
        op(st)
        torch.cuda.synchronize()
    # which buffers contain non-finite values after the whole plan ran?
    for (opidx, shape), t in zip(pl.tags, [t for t in pl.keep if torch.is_tensor(t)]):
        pass
    bad = []
    tensors = [t for t in pl.keep if torch.is_tensor(t) and t.is_floating_point()]
    for k, t in enumerate(tensors):
        if not torch.isfinite(t.float()).all():
            bad.append((k, tuple(t.shape), str(t.dtype), float((~torch.isfinite(t.float())).float().mean())))
    print(name, "non-finite buffers:", bad[:12])
scan(plan.fwd, "fwd")
plan.dmoments.copy_(torch.randn(1, 16, 16, 8).cuda())
scan(plan.bwd, "bwd")
tensors = [t for t in plan.bwd.keep if torch.is_tensor(t) and t.is_floating_point()]
dn = tensors[18]; dq, dk, dv = tensors[16], tensors[17], tensors[14]
print("dn shape", dn.shape, "dq", dq.shape, dq.dtype)
bad = ~torch.isfinite(dn[0])
print("bad rows:", bad.any(1).nonzero().flatten().tolist()[:40])
print("bad cols:", bad.any(0).nonzero().flatten().tolist()[:70])
from dreamwaltz_g_amd import gemm
w = sd15.Weights(sd, torch.device("cuda"))
pre = "encoder.mid_block.attentions.0"
ref = dq[0].float() @ w.lin(pre + ".to_q").float() + dk[0].float() @ w.lin(pre + ".to_k").float() + dv[0].float() @ w.lin(pre + ".to_v").float()
ok = torch.isfinite(dn[0])
print("max err on finite entries:", float((dn[0][ok] - ref[ok]).abs().max()), "ref absmax", float(ref.abs().max()))
# isolated re-run of the first product into a NaN-filled buffer
out = torch.full((256, 64), float("nan"), device="cuda")
gemm.gemm_raw(dq[0], w.lin(pre + ".to_q"), out, 256, 64, 64, (64, 1), (1, 64), 64)
print("isolated gemm non-finite:", int((~torch.isfinite(out)).sum()))
out2 = torch.full((256, 64), float("nan"), device="cuda")
gemm.gemm_raw(dq[0], w.lin(pre + ".to_q"), out2, 256, 64, 64, (64, 1), (1, 64), 64, accumulate=False, name="x")
print("isolated gemm2 non-finite:", int((~torch.isfinite(out2)).sum()))
