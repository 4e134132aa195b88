"""Import-compatible stand-in for the JIT-built pybind module `_gridencoder`
(/root/reference/core/nerf/gridencoder/backend.py:19-27, src/bindings.cpp:5-9).
With `<repo>/dropin` on PYTHONPATH, `import _gridencoder as _backend` (grid.py:9-16) resolves here and the
reference's own grid.py drives the HIP kernels unchanged ([L,B,C] backend layout).

dtypes: the kernels compute in fp32.  Under autocast (--optim.fp16 True: the NeRF stages of scripts/train_w_expr.sh:28,46 -- the 3DGS stages :56-94 run fp32; trainer.py:844,859) the
reference's grid.py hands over HALF embeddings / outputs / dy_dx / gradients (grid.py:28-93 casts with `embeddings.to(inputs.dtype)`
and allocates with that dtype; its CUDA backend dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF).  Here every non-fp32 buffer goes
through an fp32 temporary and is copied back into the caller's tensor in ITS dtype; nothing is ever reinterpreted.  Non-CUDA,
non-contiguous or wrongly sized buffers raise (the reference's TORCH_CHECKs, gridencoder.cu:15-18,446-462)."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd import gridencoder as _g  # noqa: E402

_FLOAT = (torch.float32, torch.float16, torch.bfloat16, torch.float64)


def _check(name, t, numel=None, floating=True):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if floating and t.dtype not in _FLOAT:
        raise RuntimeError("%s must be a floating tensor, got %s" % (name, t.dtype))
    if not floating and t.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)
    if numel is not None and t.numel() != numel:
        raise RuntimeError("%s has %d elements, expected %d" % (name, t.numel(), numel))


def _f32(t):
    return t if t is None or t.dtype == torch.float32 else t.float()


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    _check("inputs", inputs, B * D); _check("embeddings", embeddings); _check("offsets", offsets, L + 1, floating=False)
    _check("outputs", outputs, L * B * C); _check("dy_dx", dy_dx, B * L * D * C)
    out32 = outputs if outputs.dtype == torch.float32 else torch.empty(outputs.shape, dtype=torch.float32, device=outputs.device)
    dy32 = dy_dx if dy_dx is None or dy_dx.dtype == torch.float32 else torch.empty(dy_dx.shape, dtype=torch.float32, device=dy_dx.device)
    _g.grid_encode_forward(_f32(inputs), _f32(embeddings), offsets, out32, B, D, C, L, S, H, dy32, gridtype, align_corners, interp, 0)
    if out32 is not outputs:
        outputs.copy_(out32)
    if dy32 is not dy_dx and dy_dx is not None:
        dy_dx.copy_(dy32)


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp):
    _check("grad", grad, L * B * C); _check("inputs", inputs, B * D); _check("embeddings", embeddings)
    _check("offsets", offsets, L + 1, floating=False); _check("grad_embeddings", grad_embeddings, embeddings.numel())
    _check("dy_dx", dy_dx, B * L * D * C); _check("grad_inputs", grad_inputs, B * D)
    ge32 = grad_embeddings if grad_embeddings.dtype == torch.float32 else torch.zeros(grad_embeddings.shape, dtype=torch.float32,
                                                                                     device=grad_embeddings.device)
    gi32 = grad_inputs if grad_inputs is None or grad_inputs.dtype == torch.float32 else torch.empty(grad_inputs.shape, dtype=torch.float32,
                                                                                                     device=grad_inputs.device)
    if B >= 16384:
        # the slab-binned table gradient (no global float atomics, ~4x faster) OVERWRITES a zero-filled table: it runs into a zeroed
        # temporary that is then added, which keeps the backend's accumulate-into contract whatever the caller's buffer holds
        tmp = torch.zeros(ge32.shape, dtype=torch.float32, device=ge32.device)
        ws = _g.slab_workspace_for(inputs.device, B, L, int(embeddings.shape[0]))
        _g.grid_encode_backward(_f32(grad), _f32(inputs), _f32(embeddings), offsets, tmp, B, D, C, L, S, H, _f32(dy_dx), gi32, gridtype,
                                align_corners, interp, 0, slab_workspace=ws)
        ge32.add_(tmp)
    else:
        _g.grid_encode_backward(_f32(grad), _f32(inputs), _f32(embeddings), offsets, ge32, B, D, C, L, S, H, _f32(dy_dx), gi32, gridtype,
                                align_corners, interp, 0)
    if ge32 is not grad_embeddings:
        grad_embeddings.add_(ge32.to(grad_embeddings.dtype))        # the backend contract: accumulate into a pre-zeroed buffer
    if gi32 is not grad_inputs and grad_inputs is not None:
        grad_inputs.copy_(gi32)


def grad_total_variation(*args, **kwargs):
    raise NotImplementedError("grad_total_variation is not used by the trainer (SURVEY.md section 2.1)")
