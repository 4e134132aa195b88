"""densifier.GaussianDensifier: (CPU) against the reference's own GaussianDensifier + GaussianModel + torch.optim.Adam
run in the build container (tests/golden/capture_golden_densifier.py -> reference_golden_densifier.npz): ten steps of statistics, then
clone + split + prune on the flat optimizer buffers, every parameter / Adam moment / step count / statistic compared; (-m gpu) a real
SDS sub-path step with use_densifier on the kernels."""
import os

import numpy as np
import pytest
import torch

import dwg_import  # noqa: F401
from tests.test_distributed_cpu import _cpu_adam_launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "reference_golden_densifier.npz")


class _Gaussians(torch.nn.Module):
    """The GaussianModel surface the densifier reads (gaussian_model.py:12-56)."""

    def __init__(self, g):
        super().__init__()
        for k in ("_positions", "_scales", "_quaternions", "_opacities"):
            setattr(self, k, torch.nn.Parameter(torch.from_numpy(g["in." + k]).clone()))
        self._lbs_weights = torch.nn.Parameter(torch.from_numpy(g["in._lbs_weights"]).clone(), requires_grad=False)
        self._n_points = self._positions.shape[0]
    scale_activation = staticmethod(torch.exp)
    scale_inverse_activation = staticmethod(torch.log)

    @staticmethod
    def opacity_inverse_activation(x):
        return torch.log(x / (1 - x))

    def get_scales(self):
        return torch.exp(self._scales)

    def get_opacities(self):
        return torch.sigmoid(self._opacities.view(-1, 1))

    def get_quaternions(self):
        return torch.nn.functional.normalize(self._quaternions)


def test_densifier_matches_the_reference_densifier_on_its_own_optimizer():
    from dreamwaltz_g_amd import densifier as dn, optim
    g = np.load(GOLD)
    saved = optim.FlatOptimizer._launch
    optim.FlatOptimizer._launch = _cpu_adam_launch          # the fused Adam launch restated in torch (no GPU here)
    try:
        m = _Gaussians(g)
        names = [str(n) for n in g["opt.param_names"]]
        assert names == ["positions", "opacities", "scales", "quaternions"]
        lr = dict(positions=1.6e-4, opacities=0.05, scales=5e-3, quaternions=1e-3)
        groups = [{'params': [getattr(m, "_" + n)], 'lr': lr[n], 'name': n} for n in names]           # gaussian_optimizer.py:67-90 order
        opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec(groups, eps=1e-15, gaussian=dict(
            iterations=1500, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01, position_lr_max_steps=3000,
            scaling_lr=5e-3))}, torch.device("cpu"))
        params = dn.DensificationParams(max_iteration=1500, densify_grad_threshold=0.5, prune_opacity_threshold=0.1,
                                        densify_screen_size_threshold=20.0, densification_percent_distinction=0.01, disable_reset=True)
        assert [params.densify_from_iter, params.densify_until_iter, params.densification_interval, params.opacity_reset_interval] == list(g["params.iters"])
        den = dn.GaussianDensifier(m, params, opts)
        assert den.params_to_densify == names
        extent = float(g["extent"][0])
        for step in range(51, 61):
            opts["avatar"].update_learning_rate(spatial_scale=1.0, iteration=step)
            for o in opts.values():
                o.zero_grad()
            for n in names:
                getattr(m, "_" + n).grad.copy_(torch.from_numpy(g["step%d.grad._%s" % (step, n)]))
            vsp = torch.zeros(m._n_points, 3)
            vsp.grad = torch.from_numpy(g["step%d.vsp_grad" % step])
            den(viewspace_points=vsp, radii=torch.from_numpy(g["step%d.radii" % step]), spatial_extent=extent, train_step=step,
                split_samples=torch.from_numpy(g["sd.split_samples"]))
            opts["avatar"].step()
    finally:
        optim.FlatOptimizer._launch = saved
    n_out = int(g["sd.out.n_points"][0])
    assert m._n_points == n_out == m._positions.shape[0] and den.last_report[3] == n_out and n_out != 60
    buf = opts.buffers
    assert opts.all_grads() is buf.grad and m._positions.data.data_ptr() == buf.flat.data_ptr()       # re-homed into the NEW flat buffers
    for n in names + ["lbs_weights"]:
        got, ref = getattr(m, "_" + n).detach(), torch.from_numpy(g["sd.out._" + n])
        assert got.shape == ref.shape, n
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-7), (n, float((got - ref).abs().max()))
    for pg in opts["avatar"].param_groups:
        n = pg["name"]
        mm, vv = den._moments(getattr(m, "_" + n))
        assert torch.allclose(mm, torch.from_numpy(g["sd.out.exp_avg." + n]), rtol=1e-5, atol=1e-8), n
        assert torch.allclose(vv, torch.from_numpy(g["sd.out.exp_avg_sq." + n]), rtol=1e-5, atol=1e-10), n
        assert pg["t"] == int(g["sd.out.step." + n][0]), (n, pg["t"])       # the resized groups sat step 60 out, as torch's Adam does on grad None
        assert "skip_once" not in pg
    assert torch.allclose(den.points_gradient_accum, torch.from_numpy(g["sd.out.accum"])) and torch.equal(den.denom, torch.from_numpy(g["sd.out.denom"]))
    assert torch.equal(den.max_radii2D, torch.from_numpy(g["sd.out.max_radii2D"]))


def test_an_avatar_without_opacity_parameters_fails_in_prune_like_the_reference():
    """DreamWaltzG has no `_opacities` (its opacities come out of the MLP): GaussianModel.get_opacities fails on None in the reference
    (gaussian_model.py:43-47) -- the densifier needs densify_disable_prune there, and here."""
    from dreamwaltz_g_amd.avatar import DreamWaltzG
    with pytest.raises(AttributeError):
        DreamWaltzG.get_opacities(type("A", (), {"_opacities": None})())


@pytest.mark.gpu
def test_training_steps_with_the_densifier_on_the_kernels():
    """c2-like sub-path (animate + raster fwd+bwd + Adam) with cfg.render.use_densifier: statistics every step from the rasterizer's
    means2D gradient and radii, clone + split at the interval, the flat buffers re-laid, training goes on at the new Gaussian count."""
    from dreamwaltz_g_amd import configs, sds_step
    dev = torch.device("cuda:0")
    cfg = configs.TrainConfig()
    cfg.render.use_densifier = True
    cfg.render.densify_disable_prune = True          # no per-Gaussian opacity parameters on this avatar (see the CPU test above)
    cfg.render.densify_from_iter, cfg.render.densify_until_iter = 2, 1000
    cfg.render.densify_grad_threshold = 1e-7
    step = sds_step.SDSStep(n_gaussians=6000, res=128, device=dev, guidance=False, cfg=cfg, iters=300, async_pair_count=True)   # interval = 300 * 100 / 15000 = 2
    den = step.trainer.densifiers["avatar"]
    assert den.densification_interval == 2 and den.params_to_densify == ["positions", "scales", "quaternions"]
    n0, m0 = step.avatar._n_points, step.avatar._n_points_on_mesh
    counts = []
    for _ in range(6):
        loss, ro, _, _ = step.run()
        assert ro["image"].shape == (1, 128, 128, 3) and "radii" in ro and "viewspace_points" in ro
        counts.append(step.avatar._n_points)
    torch.cuda.synchronize()
    assert counts[-1] > n0 and step.avatar._n_points_on_mesh == m0, counts
    a = step.avatar
    assert a._positions.shape[0] == a._scales.shape[0] == a._quaternions.shape[0] == a._lbs_weights.shape[0] == counts[-1]
    b = step.optimizers.buffers
    assert a._positions.data.data_ptr() >= b.flat.data_ptr() and a._positions.grad.data_ptr() >= b.grad.data_ptr()
    assert den.points_gradient_accum.shape[0] == counts[-1] and torch.isfinite(b.flat).all()
    assert step.trainer.redone_frames >= 0
