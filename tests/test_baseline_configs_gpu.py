"""-m gpu: the BASELINE.json / SURVEY.md section 8(d) configurations at their FULL sizes, through the C-ABI.

  c2  50 000 Gaussians + LBS, 512x512, forward + backward      -> parity vs the oracle chain (LBS oracle -> raster oracle)
  c3  100 000 Gaussians, 512x512 rasterizer forward + backward  -> parity vs the raster oracle
  c5  300 000 Gaussians, 1024x1024 LBS + raster forward         -> parity vs the oracle, plus size-independent properties
                                                                  (colour linearity, permutation invariance, alpha range)
(c1 = 10 000 / 256x256 forward is the first case of tests/test_raster_gpu.py; c3's diffusion half is tests/test_guidance_gpu.py;
 c4 = c3 x 8 GPUs is bench.py --gpus 8, its reduce + Adam bookkeeping is tests/test_distributed_cpu.py.)
Tolerances as in tests/test_raster_gpu.py: max |err| <= 1e-4 per pixel off the algorithm's hard thresholds, threshold flips counted and
bounded (raster_cases.check_images), gradients rel-L2 <= 2e-3.
"""
import numpy as np
import pytest
import torch

from oracle import animate as oa
from tests import raster_cases as rc

pytestmark = pytest.mark.gpu


def _check_images(st, name):
    rc.check_images(st, name, note=name)


def _skeleton(seed, J=55):
    g = torch.Generator().manual_seed(seed)
    A = torch.eye(4).repeat(J, 1, 1)
    A[:, :3, :3] = oa.batch_rodrigues(torch.randn(J, 3, generator=g) * 0.3)     # pose = 0.3 N(0,1) axis-angle per joint
    A[:, :3, 3] = torch.randn(J, 3, generator=g) * 0.02
    return A


def _lbs_weights(N, seed, J=55):
    g = torch.Generator().manual_seed(seed)
    logits = torch.full((N, J), -1e9)
    logits.scatter_(1, torch.randint(0, J, (N, 4), generator=g), torch.randn(N, 4, generator=g))   # 4 non-zeros per row
    return torch.softmax(logits, dim=1)


def test_c3_raster_100k_512_forward_backward():
    G, H, W = 100000, 512, 512
    sc = rc.make_scene(G, H, W, seed=3)
    ref = rc.oracle_forward(sc)
    out = rc.hip_render(sc, requires_grad=True)
    _check_images(rc.image_err_stats(out, ref), "c3")
    wc = np.random.RandomState(7).randn(3, H, W).astype(np.float32)
    gref = rc.oracle_backward(sc, wc, None, None, dtype=np.float64)
    (out["color"] * torch.from_numpy(wc).cuda()).sum().backward()
    for name in ("means3D", "scales", "rotations", "opacities", "colors"):
        e = rc.grad_err(out["leaves"][name].grad.cpu().numpy(), gref[name])
        assert e["rel_l2"] <= 2e-3, (name, e)


def test_c2_lbs_plus_raster_50k_512_forward_backward():
    from dreamwaltz_g_amd import lbs
    G, H, W = 50000, 512, 512
    sc = rc.make_scene(G, H, W, seed=2)
    A, w = _skeleton(20), _lbs_weights(G, 21)
    # oracle chain in float64: LBS (torch autograd) -> raster oracle (C)
    p64 = sc["means3D"].double().requires_grad_(True); q64 = sc["rotations"].double().requires_grad_(True)
    po = oa.transform_points(A.double(), p64, weights=w.double())
    qo = oa.transform_quaternions_flip(A.double(), q64, w.double())
    sc_o = dict(sc); sc_o["means3D"] = po.detach().float(); sc_o["rotations"] = qo.detach().float()
    ref = rc.oracle_forward(sc_o)
    # product chain: HIP LBS blend -> HIP rasterizer
    pg = sc["means3D"].cuda().requires_grad_(True); qg = sc["rotations"].cuda().requires_grad_(True)
    ph, qh = lbs.lbs_blend(A.cuda(), w.cuda(), pg, qg, normalize_weights=True)
    assert (ph.detach().cpu().double() - po.detach()).abs().max() < 5e-6
    assert (qh.detach().cpu().double() - qo.detach()).abs().max() < 5e-5
    sc_h = dict(sc); sc_h["means3D"] = ph.detach().cpu(); sc_h["rotations"] = qh.detach().cpu()
    out = rc.hip_render(sc_h, requires_grad=True)
    _check_images(rc.image_err_stats(out, rc.oracle_forward(sc_h)), "c2 raster on the HIP-deformed Gaussians")
    st = rc.image_err_stats(out, ref)                 # end to end: the fp32 LBS differences move a few pixels across thresholds
    for k in ("color", "alpha"):
        assert st[k]["q999"] <= 2e-4, (k, st[k])
    wimg = np.random.RandomState(5).randn(3, H, W).astype(np.float32)      # dense grad_image: loss = sum(image * W_rand)
    gref = rc.oracle_backward(sc_o, wimg, None, None, dtype=np.float64)
    gp_ref, gq_ref = torch.autograd.grad([po, qo], [p64, q64], [torch.from_numpy(gref["means3D"]), torch.from_numpy(gref["rotations"])])
    (out["color"] * torch.from_numpy(wimg).cuda()).sum().backward()
    lv = out["leaves"]
    torch.autograd.backward([ph, qh], [lv["means3D"].grad, lv["rotations"].grad])
    assert rc.grad_err(pg.grad.cpu().numpy(), gp_ref.numpy())["rel_l2"] <= 2e-3
    assert rc.grad_err(qg.grad.cpu().numpy(), gq_ref.numpy())["rel_l2"] <= 2e-3


def test_c5_lbs_plus_raster_300k_1024_forward_and_properties():
    from dreamwaltz_g_amd import lbs
    G, H, W = 300000, 1024, 1024
    sc = rc.make_scene(G, H, W, seed=5)
    A, w = _skeleton(50), _lbs_weights(G, 51)
    with torch.inference_mode():
        ph, qh = lbs.lbs_blend(A.cuda(), w.cuda(), sc["means3D"].cuda(), sc["rotations"].cuda(), normalize_weights=True)
    po = oa.transform_points(A, sc["means3D"], weights=w)
    assert (ph.cpu() - po).abs().max() < 5e-6
    sc_h = dict(sc); sc_h["means3D"] = ph.cpu(); sc_h["rotations"] = qh.cpu()
    ref = rc.oracle_forward(sc_h)
    out = rc.hip_render(sc_h)
    _check_images(rc.image_err_stats(out, ref), "c5")
    a = out["alpha"]
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-6
    # colour linearity: with the geometry fixed the image is affine in the colours,
    #   I(u*c1 + v*c2) - bg*(1-alpha) = u*(I(c1) - bg*(1-alpha)) + v*(I(c2) - bg*(1-alpha))
    g = torch.Generator().manual_seed(8)
    c1, c2 = torch.rand(G, 3, generator=g), torch.rand(G, 3, generator=g)
    imgs = []
    for c in (c1, c2, 0.3 * c1 + 0.6 * c2):
        s2 = dict(sc_h); s2["colors"] = c
        imgs.append(rc.hip_render(s2)["color"])
    bgterm = 0.5 * (1.0 - a)                                   # bg = 0.5 in every channel
    lhs = imgs[2] - bgterm
    rhs = 0.3 * (imgs[0] - bgterm) + 0.6 * (imgs[1] - bgterm)
    assert float((lhs - rhs).abs().max()) < 2e-5
    # permutation invariance: the Gaussian order is not part of the result, exact depth ties aside (the id breaks them; 300 000
    # fp32 depths in one binade hold a few thousand tied pairs, of which the overlapping ones move a pixel by ~1e-5)
    perm = torch.randperm(G, generator=g)
    s3 = {k: (v[perm] if torch.is_tensor(v) and v.shape[:1] == (G,) else v) for k, v in sc_h.items()}
    out_p = rc.hip_render(s3)
    assert float((out_p["color"] - out["color"]).abs().max()) < 1e-4
    assert torch.equal(out_p["radii"].cpu(), out["radii"].cpu()[perm])
