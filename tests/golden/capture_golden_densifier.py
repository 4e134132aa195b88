"""Golden fixture of the reference's GaussianDensifier (/root/reference/core/gaussian/gaussian_densifier.py:81-387) running on the
reference's own GaussianModel + GaussianOptimizer (torch.optim.Adam): ten training steps of statistics, then one densification step
(clone + split + prune) -- inputs and the complete state afterwards -> tests/golden/reference_golden_densifier.npz (data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_golden_densifier.py       (build container only: needs /root/reference)

Inert stand-ins satisfy the imports (tests/golden/_ref_stubs.py); the arithmetic that runs is the reference's, except the
pytorch3d.quaternion_to_matrix used by densify_and_split (arithmetic stand-in from oracle.animate -> key prefix "sd.").  The N(0, scale)
samples densify_and_split draws from the global generator are RECORDED (torch.normal wrapped) so that the mirror can be fed the same."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE); sys.path.insert(0, "/root/reference")
from oracle import animate as oa  # noqa: E402
import _ref_stubs  # noqa: E402

OUT = {}


def put(k, v):
    OUT[k] = np.array(v.detach().cpu().numpy() if torch.is_tensor(v) else v, copy=True)      # a COPY: Adam updates the parameters in place


def main():
    _ref_stubs.install(oa)
    from core.gaussian.gaussian_model import GaussianModel
    from core.gaussian.gaussian_optimizer import GaussianOptimizer, OptimizationParams
    from core.gaussian.gaussian_densifier import GaussianDensifier, DensificationParams
    g = torch.Generator().manual_seed(0)
    N = 60
    m = GaussianModel()
    m.device = torch.device("cpu")
    m._positions = nn.Parameter((torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([0.4, 0.9, 0.2]))
    m._scales = nn.Parameter(torch.log(torch.rand(N, 3, generator=g) * 0.05 + 0.002))
    m._quaternions = nn.Parameter(torch.randn(N, 4, generator=g))
    m._opacities = nn.Parameter(torch.randn(N, 1, generator=g) * 2.0)
    m._lbs_weights = nn.Parameter(torch.softmax(torch.randn(N, 55, generator=g), -1), requires_grad=False)
    m._n_points = N
    for k in ("_positions", "_scales", "_quaternions", "_opacities", "_lbs_weights"):
        put("in." + k, getattr(m, k))
    op = OptimizationParams(iterations=1500, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                            position_lr_max_steps=3000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
    opt = GaussianOptimizer(m, op)
    put("opt.param_names", np.array(opt.param_names))
    params = DensificationParams(max_iteration=1500, densify_grad_threshold=0.5, prune_opacity_threshold=0.1,
                                 densify_screen_size_threshold=20.0, densification_percent_distinction=0.01, disable_reset=True)
    put("params.iters", np.array([params.densify_from_iter, params.densify_until_iter, params.densification_interval, params.opacity_reset_interval]))
    den = GaussianDensifier(m, params, opt)
    extent = 2.0
    recorded = []
    real_normal = torch.normal

    def rec_normal(*a, **k):
        s = real_normal(*a, **k)
        recorded.append(s.clone())
        return s
    torch.normal = rec_normal
    try:
        for step in range(51, 61):
            # a training step: gradients for the four groups, then densify (trainer.py:876-886), then Adam
            opt.update_learning_rate(spatial_scale=1.0, iteration=step)
            for k in ("_positions", "_scales", "_quaternions", "_opacities"):
                p = getattr(m, k)
                p.grad = torch.randn(p.shape, generator=g) * 0.1
                put("step%d.grad.%s" % (step, k), p.grad)
            vsp = torch.zeros(m._n_points, 3)
            vsp.grad = torch.randn(m._n_points, 3, generator=g) * 0.6
            radii = torch.randint(0, 30, (m._n_points,), generator=g).int()
            radii[torch.rand(m._n_points, generator=g) < 0.2] = 0
            put("step%d.vsp_grad" % step, vsp.grad); put("step%d.radii" % step, radii)
            den(viewspace_points=vsp, radii=radii, spatial_extent=extent, train_step=step)
            opt.step()
        put("sd.split_samples", recorded[0] if recorded else torch.zeros(0, 3))
        assert len(recorded) == 1
    finally:
        torch.normal = real_normal
    for k in ("_positions", "_scales", "_quaternions", "_opacities", "_lbs_weights"):
        put("sd.out." + k, getattr(m, k))
    for grp in opt.optimizer.param_groups:
        st = opt.optimizer.state.get(grp["params"][0], None)
        put("sd.out.has_state." + grp["name"], np.array([st is not None and len(st) > 0]))
        if st:
            put("sd.out.exp_avg." + grp["name"], st["exp_avg"]); put("sd.out.exp_avg_sq." + grp["name"], st["exp_avg_sq"])
            put("sd.out.step." + grp["name"], np.array([float(st["step"])]))
    put("sd.out.n_points", np.array([m._n_points]))
    put("sd.out.accum", den.points_gradient_accum); put("sd.out.denom", den.denom); put("sd.out.max_radii2D", den.max_radii2D)
    put("extent", np.array([extent]))
    np.savez_compressed(os.path.join(HERE, "reference_golden_densifier.npz"), **OUT)
    print("wrote", len(OUT), "arrays; N:", N, "->", m._n_points)


if __name__ == "__main__":
    main()
