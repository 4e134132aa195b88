"""dreamwaltz_g_amd -- MI355X-native SDS render-and-distill hot path of DreamWaltz-G.

Only what the hot path needs lives here:
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (libdwg_hip.so)
  _lib.py      ctypes binding of the C-ABI (fails loudly when the library is missing)
  rasterizer   drop-in for `diff_gaussian_rasterization` (boundary B1)
  camera       synthetic camera builder following the reference's matrix conventions
"""
__version__ = "0.1.0"
