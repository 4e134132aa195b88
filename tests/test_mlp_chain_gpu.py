"""-m gpu: the per-Gaussian MLPs (nerf_model.py:12-33 MLP, deform_model.py:61-143 DeformNetwork) as ONE forward launch and ONE backward launch
+ reduce (csrc/elementwise.hip k_mlp_chain / k_mlp_chain_bwd) against plain PyTorch float64 autograd of the same chain, and the fused
backward against the layer-by-layer kernels it replaces.  Tolerance: fp32 summation order only (exact-f32 MFMA) -- 2e-5 rel-L2 on every
gradient, stated here; the fused backward is bit-reproducible run to run (fixed-order reduce, no atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, r):
    a = a.detach().double().cpu(); r = r.detach().double().cpu()
    return float((a - r).norm() / r.norm().clamp_min(1e-30))


def _act64(v, a):
    if a == "relu":
        return torch.relu(v)
    if a == "leaky_relu":
        return torch.nn.functional.leaky_relu(v, 0.01)
    if a == "sigmoid":
        return torch.sigmoid(v)
    return v


def _reference(x, layers, acts, extra):
    """float64 autograd on the CPU: h = act(h W[:, :K]^T + b + W[:, K:] extra)."""
    x = x.detach().double().cpu().requires_grad_(True)
    ps = [(w.detach().double().cpu().requires_grad_(True), None if b is None else b.detach().double().cpu().requires_grad_(True)) for w, b in layers]
    e = None if extra is None else extra.detach().double().cpu().reshape(-1)
    h = x
    for l, ((w, b), a) in enumerate(zip(ps, acts)):
        K = h.shape[1]
        z = h @ w[:, :K].T
        if l == 0 and e is not None:
            z = z + (w[:, K:] @ e)[None, :]
        if b is not None:
            z = z + b
        h = _act64(z, a)
    return x, ps, h


def _without_rows_on_a_kink(x, layers, acts, extra, eps=1e-5):
    h = x.double()
    e = None if extra is None else extra.detach().double().cpu().reshape(-1)
    near = torch.zeros(x.shape[0], dtype=torch.bool)
    for l, ((w, b), a) in enumerate(zip(layers, acts)):
        w64 = w.detach().double().cpu()
        K = h.shape[1]
        z = h @ w64[:, :K].T
        if l == 0 and e is not None:
            z = z + (w64[:, K:] @ e)[None, :]
        if b is not None:
            z = z + b.detach().double().cpu()
        if a in ("relu", "leaky_relu"):
            near |= (z.abs() < eps).any(1)
        h = _act64(z, a)
    good = torch.nonzero(~near).flatten()
    if near.any() and good.numel():
        x = x.clone()
        x[near] = x[good[0]].clone()
    return x


CASES = [
    # (M, Kin, widths, acts, n_extra, bias)
    (50001, 32, (64, 64, 4), ("relu", "relu", None), 0, True),                                                  # the static MLP (colour | opacity)
    (30011, 32, (64, 64, 64, 64, 10), ("leaky_relu",) * 4 + (None,), 63, True),                                # DeformNetwork + fused heads
    (1, 8, (8,), (None,), 0, True),
    (63, 64, (16, 64, 8, 24, 40, 3), ("relu", "leaky_relu", "sigmoid", "relu", "leaky_relu", "sigmoid"), 5, True),
    (64, 16, (32, 5), ("sigmoid", None), 0, False),
    (65, 40, (48, 56, 64), ("relu", "relu", "relu"), 0, True),
    (16385, 32, (64, 33), ("leaky_relu", None), 2, True),
]


@pytest.mark.parametrize("M,Kin,widths,acts,n_extra,bias", CASES)
def test_chain_forward_and_fused_backward_match_float64_autograd(M, Kin, widths, acts, n_extra, bias):
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import mlp
    assert not mlp.PER_LAYER_BACKWARD
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M + Kin)
    x = torch.randn(M, Kin, generator=g)
    layers, k = [], Kin
    for l, n in enumerate(widths):
        kin = k + (n_extra if l == 0 else 0)
        w = (torch.randn(n, kin, generator=g) / kin ** 0.5).to(dev).requires_grad_(True)
        b = (0.1 * torch.randn(n, generator=g)).to(dev).requires_grad_(True) if bias else None
        layers.append((w, b)); k = n
    extra = torch.randn(n_extra, generator=g).to(dev) if n_extra else None
    gy = torch.randn(M, widths[-1], generator=g).to(dev)
    # ReLU / leaky ReLU have a kink at 0: a row with a pre-activation within rounding distance of it takes EITHER slope legitimately (the
    # fp32 sum's order decides), and one such unit among 50 001 x 128 moves the rel-L2 of every gradient by ~1e-3.  Rows whose float64
    # pre-activations come closer than 1e-5 to a kink are replaced by a row that does not (same M, same tails).
    x = _without_rows_on_a_kink(x, layers, acts, extra)
    x = x.to(dev).requires_grad_(True)
    y = mlp.mlp_chain(x, layers, list(acts), extra=extra)
    y.backward(gy)
    xr, pr, yr = _reference(x, layers, acts, extra)
    yr.backward(gy.double().cpu())
    assert _rel(y, yr) < 2e-6
    errs = {"dx": _rel(x.grad, xr.grad)}
    for l, ((w, b), (wr, br)) in enumerate(zip(layers, pr)):
        errs["dw%d" % l] = _rel(w.grad, wr.grad)
        if b is not None:
            errs["db%d" % l] = _rel(b.grad, br.grad)
    print("[parity] mlp chain M=%d widths=%s:" % (M, widths), {k: "%.1e" % v for k, v in errs.items()})
    assert max(errs.values()) < 2e-5, errs
    # bit-reproducible
    first = [x.grad.clone()] + [w.grad.clone() for w, _ in layers] + [b.grad.clone() for _, b in layers if b is not None]
    x.grad = None
    for w, b in layers:
        w.grad = None
        if b is not None:
            b.grad = None
    mlp.mlp_chain(x, layers, list(acts), extra=extra).backward(gy)
    again = [x.grad] + [w.grad for w, _ in layers] + [b.grad for _, b in layers if b is not None]
    assert all(torch.equal(a, b) for a, b in zip(first, again))


def test_fused_backward_equals_the_layer_by_layer_kernels(monkeypatch):
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import mlp
    dev = torch.device("cuda")
    M, Kin, widths, acts, n_extra = 20000, 32, (64, 64, 64, 64, 10), ["leaky_relu"] * 4 + [None], 63
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(M, Kin, generator=g).to(dev)
    ws, k = [], Kin
    for l, n in enumerate(widths):
        kin = k + (n_extra if l == 0 else 0)
        ws.append(((torch.randn(n, kin, generator=g) / kin ** 0.5).to(dev), (0.1 * torch.randn(n, generator=g)).to(dev))); k = n
    extra = torch.randn(n_extra, generator=g).to(dev)
    gy = torch.randn(M, widths[-1], generator=g).to(dev)
    outs = []
    for per_layer in (False, True):
        monkeypatch.setattr(mlp, "PER_LAYER_BACKWARD", per_layer)
        x = x0.clone().requires_grad_(True)
        layers = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in ws]
        mlp.mlp_chain(x, layers, acts, extra=extra).backward(gy)
        outs.append([x.grad] + [w.grad for w, _ in layers] + [b.grad for _, b in layers])
    for a, b in zip(*outs):
        assert _rel(a, b) < 1e-5


def test_chain_backward_without_an_input_gradient_and_on_an_empty_batch():
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import mlp
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    layers = [((torch.randn(64, 32, generator=g) / 6).to(dev).requires_grad_(True), torch.zeros(64, device=dev, requires_grad=True)),
              ((torch.randn(4, 64, generator=g) / 8).to(dev).requires_grad_(True), None)]
    x = torch.randn(777, 32, generator=g).to(dev)                     # no gradient wanted for the input
    y = mlp.mlp_chain(x, layers, ["relu", None])
    y.sum().backward()
    xr, pr, yr = _reference(x, layers, ["relu", None], None)
    yr.sum().backward()
    assert _rel(layers[0][0].grad, pr[0][0].grad) < 2e-5 and _rel(layers[1][0].grad, pr[1][0].grad) < 2e-5
    for w, b in layers:
        w.grad = None
    e = torch.zeros(0, 32, device=dev, requires_grad=True)
    mlp.mlp_chain(e, layers, ["relu", None]).sum().backward()
    assert e.grad.shape == (0, 32) and float(layers[0][0].grad.abs().sum()) == 0.0


def test_gradients_of_flat_buffer_parameters_are_added_in_place():
    """Parameters whose .grad is a slice of a flat gradient buffer (optim.FlatBuffers) get their MLP gradients ADDED into the slice by the
    reduce kernel (no AccumulateGrad launch): two backwards accumulate, participation is recorded for the optimizer, a parameter that is not
    part of the flat buffer (here: the last layer's weight, passed as a concatenation like the deformation network's heads) still gets its
    gradient through autograd, and `gridencoder.table_grad_inplace(False)` switches the in-place path off."""
    import dwg_import  # noqa: F401
    from dreamwaltz_g_amd import gridencoder, mlp, optim
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    net = mlp.MLP(32, 4, 64, 3).to(dev)
    head_a = torch.nn.Parameter((torch.randn(2, 64, generator=g) / 8).to(dev)); head_b = torch.nn.Parameter((torch.randn(2, 64, generator=g) / 8).to(dev))
    params = [p for p in net.parameters()]
    flat = optim.FlatBuffers(params + [head_a, head_b], dev)
    x = torch.randn(5000, 32, generator=g).to(dev)
    gy = torch.randn(5000, 4, generator=g).to(dev)

    def run():
        layers = [(m.weight, m.bias) for m in net.net[:2]] + [(torch.cat([head_a, head_b], 0), net.net[2].bias)]
        return mlp.mlp_chain(x, layers, ["relu", "relu", None])
    xr, pr, yr = _reference(x, [(m.weight, m.bias) for m in net.net[:2]] + [(torch.cat([head_a, head_b], 0), net.net[2].bias)], ["relu", "relu", None], None)
    yr.backward(gy.double().cpu())
    for rounds in (1, 2):
        flat.grad.zero_()
        for p in params + [head_a, head_b]:
            p._dwg_touched = False
        for _ in range(rounds):
            run().backward(gy)
        assert all(p._dwg_touched for p in (net.net[0].weight, net.net[0].bias, net.net[1].weight, net.net[1].bias, net.net[2].bias, head_a, head_b))
        assert not net.net[2].weight._dwg_touched                       # not on this path at all
        for l in range(2):
            assert net.net[l].weight.grad.data_ptr() == flat.grad[flat.slices[2 * l][0]:].data_ptr()     # still the flat slice
            assert _rel(net.net[l].weight.grad, rounds * pr[l][0].grad) < 2e-5 and _rel(net.net[l].bias.grad, rounds * pr[l][1].grad) < 2e-5
        assert _rel(torch.cat([head_a.grad, head_b.grad], 0), rounds * pr[2][0].grad) < 2e-5 and _rel(net.net[2].bias.grad, rounds * pr[2][1].grad) < 2e-5
    inplace = flat.grad.clone()
    flat.grad.zero_()
    with gridencoder.table_grad_inplace(False):
        out = run()
    for _ in range(1):
        out.backward(gy, retain_graph=True); out.backward(gy)
    assert _rel(flat.grad, inplace) < 1e-6
