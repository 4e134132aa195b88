"""The "f32x" split-precision tensor format (csrc/dwg_xfmt.h, include/dwg_types.h: DWG_DTYPE_F32X) on the Python side.

An fp32 value x is held as two fp16 halves, hi = fp16(x) and lo = fp16((x - hi) * 2^11); eight consecutive channels occupy 32 bytes --
their eight hi halves, then their eight lo halves.  Such a tensor has the byte size and the logical shape of the fp32 tensor it stands for
and is carried as a `torch.int32` tensor (opaque 4-byte words), so shapes, strides, `view`s and channel slices at multiples of 8 keep
working on it.  `pack` / `unpack` are the input / output converters of the f32x plans (sd15.py) and of the tests; the kernels split
what they produce themselves.  The reference runs this stage in fp32 (/root/reference/configs/__init__.py:236,241): f32x is how the
build reaches that precision on the 16-bit MFMA pipe.
"""
import torch

LO_SCALE = 2048.0
X_MAX = 65504.0
DTYPE = torch.int32        # storage type of an f32x tensor


def pack(x: torch.Tensor) -> torch.Tensor:
    """fp32 [..., C] (C % 8 == 0) -> f32x (int32 [..., C])."""
    if x.shape[-1] % 8:
        raise ValueError("f32x tensors need a channel count that is a multiple of 8, got %d" % x.shape[-1])
    x = x.float().clamp(-X_MAX, X_MAX).contiguous()
    hi = x.half()
    lo = ((x - hi.float()) * LO_SCALE).half()
    g = x.shape[:-1] + (x.shape[-1] // 8, 8)
    both = torch.stack([hi.view(g), lo.view(g)], dim=-2)              # [..., C/8, 2, 8] halves = 32 bytes per group
    return both.reshape(x.shape[:-1] + (x.shape[-1] * 2,)).view(DTYPE)


def unpack(t: torch.Tensor) -> torch.Tensor:
    """f32x (int32 [..., C]) -> fp32 [..., C]."""
    if t.dtype != DTYPE:
        raise TypeError("expected an f32x (int32) tensor, got %s" % t.dtype)
    t = t.contiguous()
    h = t.view(torch.float16).view(t.shape[:-1] + (t.shape[-1] // 8, 2, 8)).float()
    return (h[..., 0, :] + h[..., 1, :] * (1.0 / LO_SCALE)).reshape(t.shape)


def store(dst: torch.Tensor, x: torch.Tensor):
    """dst (f32x, possibly a channel slice of a wider buffer) <- fp32 x of the same logical shape."""
    dst.copy_(pack(x))
