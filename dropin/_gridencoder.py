"""Import-compatible stand-in for the JIT-built pybind module `_gridencoder`
(/root/reference/core/nerf/gridencoder/backend.py:19-27, src/bindings.cpp:5-9).
With `<repo>/dropin` on PYTHONPATH, `import _gridencoder as _backend` (grid.py:9-16) resolves here and the
reference's own grid.py drives the HIP kernels unchanged ([L,B,C] backend layout)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd import gridencoder as _g  # noqa: E402


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    _g.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, 0)


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp):
    _g.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                            gridtype, align_corners, interp, 0)


def grad_total_variation(*args, **kwargs):
    raise NotImplementedError("grad_total_variation is not used by the trainer (SURVEY.md section 2.1)")
