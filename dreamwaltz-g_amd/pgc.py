"""Gradient hooks on the rendered RGB image.  `build_grad_hook_func` has the behaviour of
/root/reference/core/guidance/pgc.py:15-43 (off by default: guide.grad_rgb_clip / grad_rgb_norm False; pinned on the goldens `pgc.*`):
clip the image gradient to +- grad_clip_scale times its RMS over the non-zero (optionally masked) entries, then optionally L2-normalise
it per image.  `build_pgc_hook_func` (pgc.py:46-79) is NOT mirrored: in the reference its hook raises UnboundLocalError on first use
(it assigns to a closure variable), so there is no behaviour to reproduce -- asking for it raises here too, with that explanation."""
import torch


class _RgbGradientHook:
    def __init__(self, clip, norm, clip_scale, scaler, mask):
        self.clip, self.norm, self.clip_scale, self.scaler, self.mask = bool(clip), bool(norm), clip_scale, scaler, mask

    @staticmethod
    def _rms_of_nonzero(values: torch.Tensor) -> torch.Tensor:
        return torch.sqrt(values.square().sum() / values.count_nonzero())

    def _clipped(self, grad: torch.Tensor) -> torch.Tensor:
        sample = grad
        if self.mask is not None:
            m = self.mask.expand_as(grad)
            grad.mul_(m)                                   # in place on the incoming gradient, as the reference does
            sample = grad[m > 0.5]
        sample = torch.nan_to_num(sample, nan=0.0, posinf=0.0, neginf=0.0)
        bound = self._rms_of_nonzero(sample) * self.clip_scale
        return torch.nan_to_num(torch.minimum(torch.maximum(grad, -bound), bound), nan=0.0)

    def __call__(self, grad: torch.Tensor) -> torch.Tensor:
        out = self._clipped(grad) if self.clip else grad
        if self.norm:
            out = torch.nn.functional.normalize(out, p=2, dim=(1, 2, 3))
            if self.scaler is not None and self.scaler._enabled:
                out *= self.scaler._get_scale_async()
        return out


def build_grad_hook_func(grad_clip: bool, grad_norm: bool, grad_clip_scale: float, scaler=None, mask=None):
    return _RgbGradientHook(grad_clip, grad_norm, grad_clip_scale, scaler, mask)


def build_pgc_hook_func(clip_value: float, pgc_suppress_type: int, scaler=None):
    raise NotImplementedError(
        "guide.pgc_clip_rgb >= 0: the reference's build_pgc_hook_func hook fails with UnboundLocalError when it runs "
        "(core/guidance/pgc.py:49-51 assigns to the closure variable clip_value), so this option has no reference behaviour")
