"""ctypes binding of the C-ABI in include/*.h (libdwg_hip.so).

The product path has no CPU fallback: if the HIP library is missing this module raises at first use.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DWG_LIB") or os.path.join(_HERE, "csrc", "libdwg_hip.so")      # DWG_LIB: an alternative build of the library (A/B experiments)
_lib = None

c_f32p = ctypes.c_void_p
c_i32p = ctypes.c_void_p


class RasterSettingsC(ctypes.Structure):
    """struct dwg_raster_settings (include/dwg_raster.h)."""
    _fields_ = [
        ("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
        ("sh_degree", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32),
        ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
        ("bg", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p), ("projmatrix", ctypes.c_void_p),
        ("campos", ctypes.c_void_p), ("visit_order", ctypes.c_void_p), ("tanfov", ctypes.c_void_p),
    ]


class AdamGroupC(ctypes.Structure):
    """include/dwg_elementwise.h dwg_adam_group."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("hyper_row", ctypes.c_int32)]


class SegmentC(ctypes.Structure):
    """include/dwg_gaussian.h dwg_segment."""
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p), ("count", ctypes.c_int64)]


class RasterFramesC(ctypes.Structure):
    """struct dwg_raster_frames (include/dwg_raster.h)."""
    _fields_ = [("num_frames", ctypes.c_int32), ("gaussian_stride", ctypes.c_int64), ("camera_stride", ctypes.c_int64)]


# name -> (restype, argtypes); every symbol include/*.h declares must be listed here (tests check it)
_vp, _i32, _i64, _f32, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
_u32 = ctypes.c_uint32
SIGNATURES = {
    "dwg_raster_workspace_sizes": (ctypes.c_int, [_i32, _i32, _i32, _i64, ctypes.POINTER(_sz), ctypes.POINTER(_sz),
                                                  ctypes.POINTER(_sz)]),
    "dwg_raster_num_pairs_ptr": (_vp, [_vp]),
    "dwg_raster_camera_setup": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "dwg_raster_camera_block": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_raster_forward_bin": (ctypes.c_int, [ctypes.POINTER(RasterSettingsC), _i32] + [_vp] * 9 + [_vp]),
    "dwg_raster_forward_render": (ctypes.c_int, [ctypes.POINTER(RasterSettingsC), _i32, _vp, _vp, _i64, _vp, _vp, _vp,
                                                 _vp, _vp]),
    "dwg_raster_forward_bin_frames": (ctypes.c_int, [ctypes.POINTER(RasterSettingsC), ctypes.POINTER(RasterFramesC), _i32] + [_vp] * 9 + [_vp]),
    "dwg_raster_forward_render_frames": (ctypes.c_int, [ctypes.POINTER(RasterSettingsC), ctypes.POINTER(RasterFramesC), _i32, _vp, _vp, _i64, _vp,
                                                        _vp, _vp, _vp, _vp]),
    "dwg_raster_backward": (ctypes.c_int, [ctypes.POINTER(RasterSettingsC), _i32] + [_vp] * 7 + [_vp, _vp, _i64, _vp, _vp]
                            + [_vp] * 3 + [_vp] * 8 + [_vp]),
    "dwg_raster_backward_frames": (ctypes.c_int, [ctypes.POINTER(RasterSettingsC), ctypes.POINTER(RasterFramesC), _i32] + [_vp] * 7
                                   + [_vp, _vp, _i64, _vp, _vp] + [_vp] * 3 + [_vp] * 8 + [_vp]),
    # include/dwg_lbs.h
    "dwg_lbs_joint_chain": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "dwg_lbs_blend_forward": (ctypes.c_int, [_i32, _i32, _i32] + [_vp] * 7 + [_vp]),
    "dwg_lbs_blend_backward": (ctypes.c_int, [_i32] + [_vp] * 7 + [_vp]),
    "dwg_lbs_vertex_transform": (ctypes.c_int, [_i32] * 4 + [_vp] * 8 + [_vp]),
    "dwg_lbs_vertex_transform_backward_shape": (ctypes.c_int, [_i32] * 3 + [_vp] * 9 + [_vp]),
    "dwg_lbs_vertex_transform_backward_shape_ws": (ctypes.c_int, [_i32] * 3 + [_vp] * 10 + [_vp]),
    "dwg_lbs_vertex_transform_backward_shape_workspace_floats": (ctypes.c_size_t, [_i32]),
    # include/dwg_gridenc.h
    "dwg_grid_encode_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _u32, _u32,
                                               _u32, _vp]),
    "dwg_grid_encode_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32,
                                                _u32, _u32, _u32, _vp, _vp]),
    "dwg_grid_encode_backward_xcd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32,
                                                    _u32, _u32, _u32, _vp, _vp, _vp]),
    "dwg_grid_encode_backward_owner": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32,
                                                      _u32, _u32, _u32, _vp, _vp, _vp]),
    "dwg_grid_backward_slabs_workspace_bytes": (_sz, [_u32, _u32, _u32]),
    "dwg_grid_encode_backward_slabs": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _u32,
                                                      _u32, _u32, _vp, _vp, _sz, _vp]),
    "dwg_grid_encode_backward_slabs_accumulate": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32,
                                                                 _u32, _u32, _u32, _vp, _vp, _sz, _vp]),
    # include/dwg_gemm.h
    "dwg_gemm": (ctypes.c_int, [_vp, _vp]),
    "dwg_gemm_workspace_bytes": (_sz, [_vp]),
    "dwg_transpose_2byte": (ctypes.c_int, [_i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "dwg_transpose_dt": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "dwg_xfmt_pack": (ctypes.c_int, [_i64, _vp, _vp, _vp]),
    "dwg_xfmt_unpack": (ctypes.c_int, [_i64, _vp, _vp, _vp]),
    "dwg_vae_image_pack": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp]),
    "dwg_vae_grad_prescale_pack": (ctypes.c_int, [_i32, _i32, _vp, ctypes.c_float, _vp, _vp, _vp]),
    "dwg_vae_dx_unpack": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dwg_xfmt_range_scan": (ctypes.c_int, [_i64, _vp, _vp, _vp]),
    # include/dwg_elementwise.h
    "dwg_act_backward_colsum": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dwg_mlp_chain_forward": (ctypes.c_int, [_i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "dwg_mlp_chain_backward_workspace_floats": (ctypes.c_size_t, [_i32, _i32]),
    "dwg_mlp_chain_backward": (ctypes.c_int, [_i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp,
                                              _vp, _vp, _i32, _vp, _vp]),
    "dwg_adam_step": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "dwg_adam_step_dev": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp]),
    "dwg_adam_step_groups_dev": (ctypes.c_int, [_i32, _vp, _vp, _vp]),
    # include/dwg_nn.h
    "dwg_groupnorm_forward": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp]),
    "dwg_groupnorm_backward": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dwg_groupnorm_workspace_floats": (_sz, [_i32, _i32]),
    "dwg_layernorm_forward": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _f32, _vp, _vp]),
    "dwg_geglu_forward": (ctypes.c_int, [_i64, _i32, _vp, _vp, _vp]),
    "dwg_attention_forward": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                             _vp, _i64, _i64, _f32, _vp]),
    "dwg_attention_forward_dt": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                                _vp, _i64, _i64, _f32, _vp]),
    "dwg_attention_split_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "dwg_attention_forward_ws": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                                _vp, _i64, _i64, _f32, _vp, _sz, _vp]),
    "dwg_softmax_rows_forward": (ctypes.c_int, [_i32, _i32, _f32, _vp, _i64, _vp, _i64, _vp]),
    "dwg_softmax_rows_backward": (ctypes.c_int, [_i32, _i32, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "dwg_groupnorm_forward_dt": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp]),
    "dwg_groupnorm_backward_dt": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dwg_layernorm_forward_dt": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _f32, _vp, _vp]),
    "dwg_geglu_forward_dt": (ctypes.c_int, [_i32, _i64, _i32, _vp, _vp, _vp]),
    "dwg_softmax_rows_forward_dt": (ctypes.c_int, [_i32, _i32, _i32, _f32, _vp, _i64, _vp, _i64, _vp]),
    "dwg_softmax_rows_backward_dt": (ctypes.c_int, [_i32, _i32, _i32, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "dwg_add_dt": (ctypes.c_int, [_i32, _i64, _vp, _vp, _vp, _vp]),
    "dwg_cast_f32_to_dt": (ctypes.c_int, [_i32, _i64, _vp, _vp, _vp]),
    "dwg_mlp_wgrad_workspace_floats": (_sz, [_i32]),
    "dwg_mlp_wgrad": (ctypes.c_int, [_i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp]),
    "dwg_concat_channels": (ctypes.c_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dwg_add_bf16": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "dwg_interleave2x2": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_cast_f32_to_bf16": (ctypes.c_int, [_i64, _vp, _vp, _vp]),
    # include/dwg_sds.h
    "dwg_sds_posterior_sample": (ctypes.c_int, [_i32, ctypes.c_int64, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "dwg_sds_posterior_sample_backward": (ctypes.c_int, [_i32, ctypes.c_int64, _vp, _vp, ctypes.c_float, _vp, _vp, _vp]),
    "dwg_sds_add_noise": (ctypes.c_int, [_i32, ctypes.c_int64, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "dwg_sds_gradient": (ctypes.c_int, [_i32, ctypes.c_int64, _vp, _vp, _vp, _i32, _vp, ctypes.c_float, _i32, _i32, _vp, _vp, _vp]),
    # include/dwg_gaussian.h
    "dwg_gaussian_assemble_forward": (ctypes.c_int, [_i32, _i32, _vp, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_gaussian_assemble_forward_ld": (ctypes.c_int, [_i32, _i32, _vp, _vp, _f32, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_gaussian_assemble_backward_ld": (ctypes.c_int, [_i32, _i32, _f32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                                         _vp, _vp, _vp]),
    "dwg_copy_segments": (ctypes.c_int, [_i32, _vp, _f32, _f32, _vp]),
    "dwg_add_segments": (ctypes.c_int, [_i32, _vp, _vp]),
    "dwg_gaussian_assemble_backward": (ctypes.c_int, [_i32, _i32, _f32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                       _vp, _vp, _vp, _vp]),
    # include/dwg_meshbind.h
    "dwg_mesh_vertex_normals": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_meshbind_forward": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_meshbind_backward": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dwg_meshbind_backward_verts": (ctypes.c_int, [_i32, _i32] + [_vp] * 15 + [_vp]),
    "dwg_meshbind_backward_verts_gather": (ctypes.c_int, [_i32, _i32, _i32] + [_vp] * 18 + [_vp]),
    "dwg_mesh_vertex_normals_backward": (ctypes.c_int, [_i32, _i32] + [_vp] * 8 + [_vp]),
    # include/dwg_condition.h
    "dwg_condition_keypoints": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_float, _i32, _vp, _vp]),
    "dwg_condition_workspace_bytes": (_sz, [_i32, _i32]),
    "dwg_condition_draw": (ctypes.c_int, [_i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    # include/dwg_graph.h
    "dwg_graph_begin_capture": (ctypes.c_int, [_vp]),
    "dwg_graph_end_capture": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "dwg_graph_launch": (ctypes.c_int, [_vp, _vp]),
    "dwg_graph_destroy": (ctypes.c_int, [_vp]),
    "dwg_stream_create": (ctypes.c_int, [ctypes.POINTER(_vp)]),
    "dwg_stream_destroy": (ctypes.c_int, [_vp]),
    "dwg_stream_fork": (ctypes.c_int, [_vp, _vp]),
    # include/dwg_prof.h
    "dwg_prof_enable": (ctypes.c_int, [_i32]),
    "dwg_prof_query": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)]),
    "dwg_prof_dump": (_i64, [ctypes.c_char_p, _i64]),
    "dwg_prof_dump_symbols": (_i64, [ctypes.c_char_p, _i64]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdwg_hip.so not found at %s -- run `python dreamwaltz-g_amd/build.py` "
                "(there is no CPU fallback for the product path)" % LIB_PATH)
        # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64, and libdwg_hip.so (linked against /opt/rocm's
        # libamdhip64.so.7) must bind to THAT already-loaded runtime -- loaded the other way round the process ends up with two HIP
        # runtimes and every launch on a torch stream fails with DWG_E_LAUNCH.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with DWG error %d" % (what, rc))


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def prof_enable(on=True):
    check(lib().dwg_prof_enable(int(bool(on))), "dwg_prof_enable")


def prof_table():
    """{kernel name: (launches, total_ms)} recorded since prof_enable(True)."""
    L = lib()
    need = L.dwg_prof_dump(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    L.dwg_prof_dump(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        n, c, ms = line.split()
        out[n] = (int(c), float(ms))
    return out


def prof_symbols():
    """{kernel symbol (as rocprofv3 lists it): (launches, total_ms, algorithmic work)} since prof_enable(True)."""
    L = lib()
    need = L.dwg_prof_dump_symbols(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    L.dwg_prof_dump_symbols(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        n, c, ms, w = line.split("\t")
        out[n] = (int(c), float(ms), float(w))
    return out
