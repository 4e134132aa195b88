"""Pins the CPU rasterizer oracle: analytic backward vs central finite differences of its own forward (float64),
float32 vs float64 agreement, and structural properties (empty input, culling, determinism)."""
import numpy as np
import pytest
import torch

import dreamwaltz_g_amd.camera as camera
import dreamwaltz_g_amd.synth as synth
from oracle import raster as oraster


def _scene(G, H=64, W=64, seed=0, opacity_range=(0.05, 0.9), scale_mul=4.0):
    g = synth.random_gaussians(G, seed=seed, opacity_range=opacity_range, dtype=torch.float64)
    cam = camera.make_camera(height=H, width=W, dtype=torch.float64)
    view, proj, campos, tfx, tfy = camera.raster_matrices(cam)
    d = dict(means3D=g["positions"].numpy(), opacities=g["opacities"].numpy().reshape(-1), colors=g["colors"].numpy(),
             scales=g["scales"].numpy() * scale_mul, rotations=g["quaternions"].numpy(), viewmatrix=view.numpy(),
             projmatrix=proj.numpy(), campos=campos.numpy(), tanfovx=tfx, tanfovy=tfy,
             bg=np.array([0.5, 0.5, 0.5]), H=H, W=W)
    return d


def _loss_weights(H, W, seed=1):
    r = np.random.RandomState(seed)
    return r.randn(3, H, W), r.randn(H, W), r.randn(H, W)


def _loss(d, wc, wd, wa, **over):
    dd = dict(d); dd.update(over)
    o = oraster.forward(dtype=np.float64, **dd)
    return float((o["color"] * wc).sum() + (o["depth"] * wd).sum() + (o["alpha"] * wa).sum())


def _fd_check(lossfn, base, ana, idx, eps, name):
    """Central differences at two step sizes; the forward has genuine discontinuities (alpha<1/255, T<1e-4,
    radius/tile rect), so a sample is only compared where both step sizes agree (smooth neighbourhood)."""
    compared = 0
    for ix in idx:
        fds = []
        for e in (eps, eps * 0.25):
            p = base.copy(); p[ix] += e
            m = base.copy(); m[ix] -= e
            fds.append((lossfn(p) - lossfn(m)) / (2 * e))
        if abs(fds[0] - fds[1]) > 1e-3 * max(1.0, abs(fds[0])):
            continue
        compared += 1
        a = ana[ix]
        assert abs(fds[1] - a) <= 1e-4 * max(1.0, abs(fds[1]), abs(a)), (name, ix, fds, a)
    assert compared >= len(idx) * 0.7, (name, compared)


def test_forward_f32_matches_f64():
    d = _scene(400)
    o64 = oraster.forward(dtype=np.float64, **d)
    o32 = oraster.forward(dtype=np.float32, **d)
    assert o64["num_pairs"] == o32["num_pairs"] > 0
    assert np.array_equal(o64["radii"], o32["radii"])
    for k in ("color", "depth", "alpha"):
        err = np.abs(o64[k] - o32[k])
        # threshold flips (alpha<1/255, T<1e-4) may differ at isolated pixels between precisions
        assert np.quantile(err, 0.999) < 1e-4, k
        assert err.max() < 2e-2, k


def test_empty_and_culled():
    d = _scene(8)
    d0 = dict(d)
    for k in ("means3D", "colors", "scales", "rotations"):
        d0[k] = d[k][:0]
    d0["opacities"] = d["opacities"][:0]
    o = oraster.forward(dtype=np.float32, **d0)
    assert o["num_pairs"] == 0
    assert np.allclose(o["color"], 0.5) and np.all(o["alpha"] == 0)
    # behind the camera -> culled (radius 0)
    d1 = dict(d)
    d1["means3D"] = d["means3D"] + np.array([0.0, 100.0, 0.0]) * 0  # keep
    cam_dir = d["campos"] / np.linalg.norm(d["campos"])
    d1["means3D"] = d["means3D"] + cam_dir * 10.0  # move past the camera
    o = oraster.forward(dtype=np.float32, **d1)
    assert np.all(o["radii"] == 0) and o["num_pairs"] == 0


@pytest.mark.parametrize("field", ["means3D", "scales", "rotations", "opacities", "colors"])
def test_backward_matches_finite_differences(field):
    d = _scene(48, H=48, W=48, seed=3)
    wc, wd, wa = _loss_weights(48, 48)
    g = oraster.backward(dtype=np.float64, g_color=wc, g_depth=wd, g_alpha=wa, **d)
    base = d[field]
    rs = np.random.RandomState(7)
    idx = [tuple(rs.randint(0, s) for s in base.shape) for _ in range(24)]
    _fd_check(lambda v: _loss(d, wc, wd, wa, **{field: v}), base, g[field], idx, 1e-6, field)


def test_backward_cov3d_precomp_and_sh():
    d = _scene(32, H=32, W=32, seed=5)
    wc, wd, wa = _loss_weights(32, 32, seed=2)
    # precomputed covariance path
    from oracle.raster import forward
    R = []
    for q, s in zip(d["rotations"], d["scales"]):
        r, x, y, z = q
        Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                       [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                       [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        S = Rm @ np.diag(s ** 2) @ Rm.T
        R.append([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])
    cov = np.array(R)
    dc = dict(d); dc.pop("scales"); dc.pop("rotations"); dc["cov3D"] = cov
    o_a = forward(dtype=np.float64, **d)
    o_b = forward(dtype=np.float64, **dc)
    assert np.allclose(o_a["color"], o_b["color"], atol=1e-12)
    g = oraster.backward(dtype=np.float64, g_color=wc, g_depth=wd, g_alpha=wa, **dc)
    rs = np.random.RandomState(0)
    idx = [(rs.randint(0, 32), rs.randint(0, 6)) for _ in range(16)]
    _fd_check(lambda v: _loss(dc, wc, wd, wa, cov3D=v), cov, g["cov3D"], idx, 1e-8, "cov3D")
    # SH colours (degree 3), gradient w.r.t. coefficients and positions
    ds = dict(d); ds.pop("colors")
    shs = rs.randn(32, 16, 3) * 0.3
    ds["shs"] = shs; ds["sh_degree"] = 3
    g = oraster.backward(dtype=np.float64, g_color=wc, g_depth=wd, g_alpha=wa, **ds)
    idx = [(rs.randint(0, 32), rs.randint(0, 16), rs.randint(0, 3)) for _ in range(16)]
    _fd_check(lambda v: _loss(ds, wc, wd, wa, shs=v), shs, g["shs"], idx, 1e-6, "shs")
    idx = [(rs.randint(0, 32), rs.randint(0, 3)) for _ in range(12)]
    _fd_check(lambda v: _loss(ds, wc, wd, wa, means3D=v), ds["means3D"], g["means3D"], idx, 1e-6, "means3D(sh)")


def test_openmp_build_of_the_oracle_equals_the_checker_build():
    """bench.py's cpu_baseline times the -fopenmp build (tiles / Gaussians across the host cores); the checker build stays single-threaded.
    Same forward bit for bit; gradients equal up to the order of the fp32 sums."""
    import numpy as np
    from tests import raster_cases as rc
    sc = rc.make_scene(3000, 96, 80, seed=4)
    a, b = rc.oracle_forward(sc), rc.oracle_forward(sc, omp=True)
    for k in ("color", "depth", "alpha", "radii", "n_contrib"):
        assert np.array_equal(a[k], b[k]), k
    assert a["num_pairs"] == b["num_pairs"]
    wc = np.random.RandomState(0).randn(3, 96, 80).astype(np.float32)
    ga = rc.oracle_backward(sc, wc, None, None, dtype=np.float32)
    gb = rc.oracle_backward(sc, wc, None, None, dtype=np.float32, omp=True)
    for k in ("means3D", "scales", "rotations", "opacities", "colors", "means2D"):
        assert rc.grad_err(gb[k], ga[k])["rel_l2"] < 1e-5, k
