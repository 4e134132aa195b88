"""-m gpu: csrc/gridenc.hip against the REFERENCE's own grid-encoder backend built for gfx950 (oracle/build_ref_gridencoder.py ->
oracle/_ref/_gridencoder_ref.so), on the c3 point set, hash and tiled grids, linear and smoothstep interpolation: forward, dy_dx, table
and input gradients.  Skipped when the shared object does not exist -- on the ROCm 7.2 image of this build the reference's .cu does not
compile without a hand-written atomicAdd(__half2*) (recorded in oracle/_ref/gridencoder_ref.unbuildable.txt), so
row L10 is then held by oracle/animate.py's restatement alone (tests/test_animate_gpu.py::test_grid_encoder_forward_backward)."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    spec = importlib.util.spec_from_file_location("dwg_build_ref", os.path.join(ROOT, "oracle", "build_ref_gridencoder.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.load()


@pytest.mark.parametrize("gridtype", [0, 1])
@pytest.mark.parametrize("interp", [0, 1])
def test_gridenc_hip_equals_the_reference_backend(gridtype, interp):
    ref = _ref()
    if ref is None:
        pytest.skip("oracle/_ref/_gridencoder_ref.so not built (the reference's gridencoder.cu does not compile on this ROCm image)")
    from dreamwaltz_g_amd import gridencoder as ge
    dev = torch.device("cuda")
    enc = ge.GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096,
                         gridtype="hash" if gridtype == 0 else "tiled", interpolation="linear" if interp == 0 else "smoothstep").to(dev)
    g = torch.Generator().manual_seed(0)
    B, L, C, D = 100000, 16, 2, 3
    x = torch.rand(B, 3, generator=g).to(dev)
    x[:50] = x[:50] * 1.4 - 0.2                      # some points outside [0, 1]: zeros forward, no gradient (checklist Q14)
    emb = (torch.rand(enc.embeddings.shape, generator=g) * 2e-1 - 1e-1).to(dev)
    S, H = float(torch.log2(torch.tensor(enc.per_level_scale))), enc.base_resolution
    out_ref = torch.empty(L, B, C, device=dev); dydx_ref = torch.empty(B, L * D * C, device=dev)
    ref.grid_encode_forward(x, emb, enc.offsets, out_ref, B, D, C, L, S, H, dydx_ref, gridtype, False, interp)
    out = torch.empty(L, B, C, device=dev); dydx = torch.empty(B, L * D * C, device=dev)
    ge.grid_encode_forward(x, emb, enc.offsets, out, B, D, C, L, S, H, dydx, gridtype, False, interp)
    assert torch.allclose(out, out_ref, rtol=1e-5, atol=1e-7)
    assert torch.allclose(dydx, dydx_ref, rtol=1e-4, atol=1e-6)
    grad = torch.randn(L, B, C, generator=g).to(dev)
    ge_ref = torch.zeros_like(emb); gi_ref = torch.zeros_like(x)
    ref.grid_encode_backward(grad, x, emb, enc.offsets, ge_ref, B, D, C, L, S, H, dydx_ref, gi_ref, gridtype, False, interp)
    ge_hip = torch.zeros_like(emb); gi_hip = torch.zeros_like(x)
    ge.grid_encode_backward(grad, x, emb, enc.offsets, ge_hip, B, D, C, L, S, H, dydx, gi_hip, gridtype, False, interp)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))   # noqa: E731
    assert rel(ge_hip, ge_ref) < 1e-5 and rel(gi_hip, gi_ref) < 1e-5
