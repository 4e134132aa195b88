"""CPU check of the hand-derived per-Gaussian activation / non-rigid composition backward (csrc/assemble_math.h compiled for
the host) against autograd through the same torch expressions the oracle's animate uses (oracle/animate.py:456-466)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))


def _hostlib():
    src = os.path.join(HERE, "hostmath", "assemble_math_host.c")
    out = os.path.join(HERE, "hostmath", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libassemble_math_host.so")
    hdr = os.path.join(HERE, "..", "dreamwaltz-g_amd", "csrc", "assemble_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src, "-lm"], check=True)
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("fix_opacity", [0, 1])
def test_assemble_forward_and_backward_match_autograd(fix_opacity):
    L = _hostlib()
    g = torch.Generator().manual_seed(4 + fix_opacity)
    n, io, isc = 500, 0.01, 0.001
    t = lambda *s, scale=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * scale).requires_grad_(True)  # noqa: E731
    p, off, ls, ms, q, h = t(n, 3), t(n, 3), t(n, 3, scale=0.5), t(n, 3), t(n, 4), t(n, 4, scale=2.0)
    pos = p + off * io
    scl = torch.exp(ls) + ms * isc
    qn = F.normalize(q, dim=-1)
    col = torch.sigmoid(h[:, 1:])
    op = torch.ones_like(h[:, :1]) if fix_opacity else torch.sigmoid(h[:, :1])
    gs = [torch.randn(x.shape, generator=g, dtype=torch.float64) for x in (pos, scl, qn, col, op)]
    no = 4 if fix_opacity else 5                      # a constant opacity carries no gradient
    grads = torch.autograd.grad([pos, scl, qn, col, op][:no], [p, off, ls, ms, q, h], gs[:no], allow_unused=True)
    f32 = lambda x: x.detach().float().numpy().copy()  # noqa: E731
    o = [np.zeros((n, k), np.float32) for k in (3, 3, 4, 3, 1)]
    L.host_assemble_forward(n, ctypes.c_float(io), ctypes.c_float(isc), fix_opacity, _p(f32(p)), _p(f32(off)), _p(f32(ls)), _p(f32(ms)),
                            _p(f32(q)), _p(f32(h)), *[_p(x) for x in o])
    for got, ref in zip(o, (pos, scl, qn, col, op)):
        assert np.abs(got - ref.detach().numpy()).max() < 2e-6
    d = [np.zeros((n, k), np.float32) for k in (3, 3, 3, 3, 4, 4)]
    L.host_assemble_backward(n, ctypes.c_float(io), ctypes.c_float(isc), fix_opacity, _p(f32(ls)), _p(f32(q)), _p(f32(h)),
                             *[_p(f32(x)) for x in gs], *[_p(x) for x in d])
    for got, ref in zip(d, grads):
        ref = np.zeros_like(got) if ref is None else ref.numpy()
        assert np.linalg.norm(got - ref) <= 1e-5 * max(np.linalg.norm(ref), 1e-12) + 1e-9
