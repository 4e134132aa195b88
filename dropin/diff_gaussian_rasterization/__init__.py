"""Import-compatible stand-in for the third-party CUDA package `diff_gaussian_rasterization`.

Put `<repo>/dropin` and `<repo>` on PYTHONPATH in a DreamWaltz-G checkout and
`core/gaussian/gaussian_renderer.py:5` resolves to the MI355X-native HIP rasterizer unchanged.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd.rasterizer import (  # noqa: E402,F401
    GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians)
