"""Gradient hooks on the rendered RGB image: mirror of build_grad_hook_func
(/root/reference/core/guidance/pgc.py:15-43; off by default: guide.grad_rgb_clip / grad_rgb_norm False).
`build_pgc_hook_func` (pgc.py:46-79) is NOT mirrored: in the reference its hook raises UnboundLocalError on first use
(`clip_value *= ...` assigns to a closure variable), so there is no behaviour to reproduce -- asking for it raises here too, with
that explanation."""
import torch


def build_grad_hook_func(grad_clip: bool, grad_norm: bool, grad_clip_scale: float, scaler=None, mask=None):
    def _hook(grad: torch.Tensor):
        if grad_clip:
            if mask is not None:
                grad *= mask.expand_as(grad)
                grad_for_std = grad[mask.expand_as(grad) > 0.5]
            else:
                grad_for_std = grad
            grad_for_std = grad_for_std.nan_to_num(0.0, 0.0, 0.0)
            std = ((grad_for_std ** 2).sum() / grad_for_std.count_nonzero()) ** 0.5 * grad_clip_scale
            grad_new = torch.minimum(torch.maximum(grad, -std), std).nan_to_num(0.0)
        else:
            grad_new = grad
        if grad_norm:
            grad_new = torch.nn.functional.normalize(grad_new, p=2, dim=(1, 2, 3))
            if scaler is not None and scaler._enabled:
                grad_new *= scaler._get_scale_async()
        return grad_new
    return _hook


def build_pgc_hook_func(clip_value: float, pgc_suppress_type: int, scaler=None):
    raise NotImplementedError(
        "guide.pgc_clip_rgb >= 0: the reference's build_pgc_hook_func hook fails with UnboundLocalError when it runs "
        "(core/guidance/pgc.py:49-51 assigns to the closure variable clip_value), so this option has no reference behaviour")
