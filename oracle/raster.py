"""ctypes front-end of oracle/raster_oracle.c (CPU restatement of the 3DGS tile rasterizer).

TEST INFRASTRUCTURE.  PARITY UNPINNED against the third-party CUDA package (see raster_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_libs = {}


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _lib(dtype, omp=False):
    """omp=True: the multi-threaded f32 build (bench.py's cpu_baseline only; the checker build is single-threaded and deterministic)."""
    key = "f64" if np.dtype(dtype) == np.float64 else "f32"
    if omp:
        if key != "f32":
            raise ValueError("the OpenMP build of the raster oracle is f32 only")
        key = "f32_omp"
    if key not in _libs:
        path = os.path.join(_BUILD, "liboracle_raster_%s.so" % key)
        if not os.path.exists(path):
            build()
        _libs[key] = ctypes.CDLL(path)
    return _libs[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _prep(dtype, *arrs):
    out = []
    for a in arrs:
        out.append(None if a is None else np.ascontiguousarray(np.asarray(a, dtype=dtype)))
    return out


def forward(means3D, opacities, viewmatrix, projmatrix, tanfovx, tanfovy, bg, H, W, colors=None, shs=None,
            sh_degree=0, campos=None, scales=None, rotations=None, cov3D=None, scale_modifier=1.0,
            dtype=np.float32, omp=False, near=None):
    """Returns dict(color[3,H,W], depth[H,W], alpha[H,W], radii[G], final_T, n_contrib, num_pairs).  `near=(rel_alpha, rel_T, pos_ulps)`:
    also `near` [H,W] uint8, the hard thresholds each pixel sits on within those margins (raster_oracle.c composite_forward; pos_ulps: how
    many fp32 ulps of the pixel coordinate two projections of one centre may differ by)."""
    L = _lib(dtype, omp)
    real = ctypes.c_double if np.dtype(dtype) == np.float64 else ctypes.c_float
    means3D, opacities, viewmatrix, projmatrix, bg, colors, shs, campos, scales, rotations, cov3D = _prep(
        dtype, means3D, opacities, viewmatrix, projmatrix, bg, colors, shs, campos, scales, rotations, cov3D)
    G = means3D.shape[0]
    ncoef = 0 if shs is None else shs.shape[1]
    color = np.zeros((3, H, W), dtype); depth = np.zeros((H, W), dtype); alpha = np.zeros((H, W), dtype)
    radii = np.zeros(G, np.int32); fT = np.zeros((H, W), dtype); nc = np.zeros((H, W), np.int32)
    K = ctypes.c_int64(0)
    args = [ctypes.c_int(G), ctypes.c_int(H), ctypes.c_int(W), _p(means3D), _p(colors), _p(shs), ctypes.c_int(sh_degree),
            ctypes.c_int(ncoef), _p(campos), _p(opacities.reshape(-1)), _p(scales), _p(rotations), _p(cov3D),
            _p(viewmatrix.reshape(-1)), _p(projmatrix.reshape(-1)), real(tanfovx), real(tanfovy), _p(bg),
            real(scale_modifier), _p(color), _p(depth), _p(alpha), _p(radii), _p(fT), _p(nc), ctypes.byref(K)]
    out = dict(color=color, depth=depth, alpha=alpha, radii=radii, final_T=fT, n_contrib=nc)
    if near is not None:
        nr = np.zeros((H, W), np.uint8)
        L.dwg_oracle_raster_forward_near.restype = ctypes.c_int
        pos_eps = float(near[2]) * 1.1920929e-07 * max(H, W) if len(near) > 2 else 0.0
        L.dwg_oracle_raster_forward_near(*args, _p(nr), real(near[0]), real(near[1]), real(pos_eps))
        out["near"] = nr
    else:
        L.dwg_oracle_raster_forward.restype = ctypes.c_int
        L.dwg_oracle_raster_forward(*args)
    out["num_pairs"] = K.value
    return out


def backward(means3D, opacities, viewmatrix, projmatrix, tanfovx, tanfovy, bg, H, W, g_color, g_depth=None,
             g_alpha=None, colors=None, shs=None, sh_degree=0, campos=None, scales=None, rotations=None, cov3D=None,
             scale_modifier=1.0, dtype=np.float32, omp=False):
    L = _lib(dtype, omp)
    real = ctypes.c_double if np.dtype(dtype) == np.float64 else ctypes.c_float
    (means3D, opacities, viewmatrix, projmatrix, bg, colors, shs, campos, scales, rotations, cov3D, g_color, g_depth,
     g_alpha) = _prep(dtype, means3D, opacities, viewmatrix, projmatrix, bg, colors, shs, campos, scales, rotations,
                      cov3D, g_color, g_depth, g_alpha)
    G = means3D.shape[0]
    ncoef = 0 if shs is None else shs.shape[1]
    out = dict(
        means3D=np.zeros((G, 3), dtype), means2D=np.zeros((G, 3), dtype), opacities=np.zeros(G, dtype),
        colors=np.zeros((G, 3), dtype), shs=None if shs is None else np.zeros_like(shs),
        scales=np.zeros((G, 3), dtype), rotations=np.zeros((G, 4), dtype), cov3D=np.zeros((G, 6), dtype))
    L.dwg_oracle_raster_backward.restype = ctypes.c_int
    L.dwg_oracle_raster_backward(
        ctypes.c_int(G), ctypes.c_int(H), ctypes.c_int(W), _p(means3D), _p(colors), _p(shs), ctypes.c_int(sh_degree),
        ctypes.c_int(ncoef), _p(campos), _p(opacities.reshape(-1)), _p(scales), _p(rotations), _p(cov3D),
        _p(viewmatrix.reshape(-1)), _p(projmatrix.reshape(-1)), real(tanfovx), real(tanfovy), _p(bg),
        real(scale_modifier), _p(g_color), _p(g_depth), _p(g_alpha), _p(out["means3D"]), _p(out["means2D"]),
        _p(out["colors"]), _p(out["shs"]), _p(out["opacities"]), _p(out["scales"]), _p(out["rotations"]),
        _p(out["cov3D"]))
    return out


def sh_colors(shs, positions, campos, deg, dtype=np.float32):
    L = _lib(dtype)
    shs, positions, campos = _prep(dtype, shs, positions, campos)
    N, M = shs.shape[0], shs.shape[1]
    out = np.zeros((N, 3), dtype)
    L.dwg_oracle_sh_colors(ctypes.c_int(N), ctypes.c_int(deg), ctypes.c_int(M), _p(shs), _p(positions), _p(campos), _p(out))
    return out
