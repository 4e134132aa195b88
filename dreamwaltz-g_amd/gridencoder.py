"""MI355X-native grid encoder behind the reference's interfaces (boundary B2).

  GridEncoder / grid_encode      mirror /root/reference/core/nerf/gridencoder/grid.py:28-165 (same ctor args,
                                 same offset table, same autograd contract)
  grid_encode_forward/backward   mirror the pybind backend `_gridencoder` (src/bindings.cpp:5-9) so that the
                                 reference's own grid.py can bind to this module unchanged (dropin/_gridencoder.py)
The arithmetic is csrc/gridenc.hip through include/dwg_gridenc.h; no CPU fallback.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib

_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("dreamwaltz_g_amd grid encoder runs on the GPU only (HIP kernels)")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                        out_layout=0):
    """Backend-compatible entry (outputs [L,B,C] when out_layout == 0)."""
    _need_cuda(inputs)
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_grid_encode_forward(p(inputs), p(embeddings), p(offsets), p(outputs), B, D, C, L,
                                                  ctypes.c_float(S), H, p(dy_dx), gridtype, int(bool(align_corners)),
                                                  interp, out_layout, _st(inputs)), "dwg_grid_encode_forward")


_host_offsets = []          # [(offsets tensor kept alive, its version, ctypes array)]
_host_arrays = {}


def _host_array(values):
    """ctypes int32 array for a tuple of level offsets (cached by CONTENT)."""
    if values not in _host_arrays:
        _host_arrays[values] = (ctypes.c_int32 * len(values))(*values)
    return _host_arrays[values]


def _host_offsets_of(offsets):
    """HOST copy of the level offset table for callers that only hand over the device tensor (the `_gridencoder` backend seam).
    The cache entry holds the tensor itself -- its storage cannot be recycled for another table while the entry lives -- and its
    version counter; the array itself is shared by CONTENT."""
    for ref, ver, arr in _host_offsets:
        if ref is offsets and ref._version == ver:
            return arr
    arr = _host_array(tuple(int(v) for v in offsets.cpu().tolist()))
    _host_offsets[:] = [e for e in _host_offsets if e[0] is not offsets][-15:] + [(offsets, offsets._version, arr)]
    return arr


_xcd_scratch = {}
_xcd_checked = {}


def xcd_scratch_for(embeddings):
    """8 XCD-private copies of the table gradient (zero between calls), one allocation per (device, table size, stream): two
    backward passes in flight on different streams never share a scratch."""
    key = (embeddings.device, embeddings.numel(), torch.cuda.current_stream(embeddings.device).cuda_stream)
    if key not in _xcd_scratch:
        _xcd_scratch[key] = torch.zeros(8, embeddings.numel(), device=embeddings.device, dtype=torch.float32)
    return _xcd_scratch[key]


def xcd_path_ok(device) -> bool:
    """The XCD-private accumulation relies on a gfx950 property outside the HIP memory model: workgroup-scope float atomics to
    addresses shared by the workgroups of one XCD are performed in that XCD's L2, hence atomic among them (MI355X_MICROARCH:
    per-XCD L2, block b -> XCD b % 8).  Guarded twice: the device must be gfx950, and a one-off self-test per device compares
    the path against the agent-scope (memory-side) atomics on a random problem; a mismatch disables it for the process."""
    key = str(device)
    if key in _xcd_checked:
        return _xcd_checked[key]
    import os
    ok = os.environ.get("DWG_GRID_NO_XCD") != "1" and "gfx950" in torch.cuda.get_device_properties(device).gcnArchName
    if ok:
        _xcd_checked[key] = False               # no recursion while the self-test runs
        g = torch.Generator().manual_seed(0)
        enc_off = np.array([0, 4920, 20552], dtype=np.int32)
        B = 20000
        x = torch.rand(B, 3, generator=g).to(device)
        table = torch.zeros(int(enc_off[-1]), 2, device=device)
        grad = torch.randn(B, 4, generator=g).to(device)
        off = torch.from_numpy(enc_off).to(device)
        ho = _host_array(tuple(int(v) for v in enc_off))
        res = []
        for mode in ("device", "copies", "owner"):
            ge = torch.zeros_like(table)
            scratch = torch.zeros(8, table.numel(), device=device) if mode == "copies" else None
            cnt = torch.zeros(16, dtype=torch.int32, device=device) if mode == "owner" else None
            grid_encode_backward(grad, x, table, off, ge, B, 3, 2, 2, 1.0, 16, None, None, 1, False, 1, grad_layout=1,
                                 xcd_scratch=scratch, host_offsets=ho, xcd_counters=cnt)
            res.append(ge)
            if mode == "copies" and float(scratch.abs().max()) != 0.0:
                ok = False
            if mode == "owner" and not bool((cnt[8:] == 1).all()):
                ok = False                       # some XCD received no workgroup: its table lines were never written
        err = max(float((res[0] - r).abs().max() / res[0].abs().max().clamp_min(1e-20)) for r in res[1:])
        ok = ok and err < 1e-4
        if not ok:
            import warnings
            warnings.warn("dreamwaltz_g_amd grid encoder: XCD-private gradient accumulation failed its self-test (rel. err %.2e); "
                          "using device-scope atomics" % err)
    _xcd_checked[key] = ok
    return ok


_xcd_counters = {}


def xcd_counters_for(device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    if key not in _xcd_counters:
        _xcd_counters[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return _xcd_counters[key]


_INPLACE_OK = [True]


class table_grad_inplace:
    """`with table_grad_inplace(False):` -- forwards issued inside the block opt OUT of the in-place table gradient (see
    _grid_encode.backward): a multi-stream multi-view step, whose concurrent backwards must not read-add-write one gradient slice.  The
    decision is taken at FORWARD time and travels on the autograd ctx; nothing global is consulted by the backward, and leaving the block
    (normally or by an exception) restores the previous setting."""

    def __init__(self, ok: bool):
        self.ok = bool(ok)

    def __enter__(self):
        self.prev, _INPLACE_OK[0] = _INPLACE_OK[0], self.ok

    def __exit__(self, *exc):
        _INPLACE_OK[0] = self.prev

_SLAB_WS = {}     # (device, stream) -> byte workspace of the slab-binned backward (grown on demand; backwards on one stream are ordered,
                  # a workspace replaced by a bigger one is only released by the caching allocator in that stream's order)


def slab_path_ok(B, L, total_entries):
    """Limits of dwg_grid_encode_backward_slabs (it returns DWG_E_CAPACITY beyond them): one u32 LDS counter per 4096-entry slab within
    64 KiB, u32 record positions."""
    return (total_entries + 4095) // 4096 * 4 <= 64 * 1024 and B * L * 8 <= 0xffffffff


def slab_workspace_for(device, B, L, total_entries):
    need = int(_lib.lib().dwg_grid_backward_slabs_workspace_bytes(B, L, total_entries))
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _SLAB_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _SLAB_WS[key] = ws
    return ws


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp, grad_layout=0, xcd_scratch=None, host_offsets=None, xcd_counters=None,
                         slab_workspace=None, accumulate=False):
    _need_cuda(inputs)
    p = _lib.ptr
    ho = host_offsets if host_offsets is not None else _host_offsets_of(offsets)
    if slab_workspace is not None:
        fn = _lib.lib().dwg_grid_encode_backward_slabs_accumulate if accumulate else _lib.lib().dwg_grid_encode_backward_slabs
        _lib.check(fn(p(grad), p(inputs), p(embeddings), p(offsets), p(grad_embeddings), B, D,
                                                             C, L, ctypes.c_float(S), H, p(dy_dx), p(grad_inputs), gridtype,
                                                             int(bool(align_corners)), interp, grad_layout,
                                                             ctypes.cast(ho, ctypes.c_void_p), p(slab_workspace), slab_workspace.numel(),
                                                             _st(inputs)), "dwg_grid_encode_backward_slabs")
        return
    if xcd_counters is not None:
        _lib.check(_lib.lib().dwg_grid_encode_backward_owner(p(grad), p(inputs), p(embeddings), p(offsets), p(grad_embeddings), B, D,
                                                             C, L, ctypes.c_float(S), H, p(dy_dx), p(grad_inputs), gridtype,
                                                             int(bool(align_corners)), interp, grad_layout,
                                                             ctypes.cast(ho, ctypes.c_void_p), p(xcd_counters), _st(inputs)),
                   "dwg_grid_encode_backward_owner")
        return
    if xcd_scratch is not None:
        _lib.check(_lib.lib().dwg_grid_encode_backward_xcd(p(grad), p(inputs), p(embeddings), p(offsets), p(grad_embeddings), B, D,
                                                           C, L, ctypes.c_float(S), H, p(dy_dx), p(grad_inputs), gridtype,
                                                           int(bool(align_corners)), interp, grad_layout,
                                                           ctypes.cast(ho, ctypes.c_void_p), p(xcd_scratch), _st(inputs)),
                   "dwg_grid_encode_backward_xcd")
        return
    _lib.check(_lib.lib().dwg_grid_encode_backward(p(grad), p(inputs), p(embeddings), p(offsets), p(grad_embeddings), B, D,
                                                   C, L, ctypes.c_float(S), H, p(dy_dx), p(grad_inputs), gridtype,
                                                   int(bool(align_corners)), interp, grad_layout,
                                                   ctypes.cast(ho, ctypes.c_void_p), _st(inputs)),
               "dwg_grid_encode_backward")


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, host_offsets=None):
        """host_offsets: ctypes int32 array holding the same values as `offsets` (the GridEncoder module keeps one: it builds
        the table on the host).  Without it the backward falls back to a per-tensor cache keyed by the device address."""
        inputs = inputs.contiguous().float()
        embeddings = embeddings.contiguous().float()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=torch.float32)   # written directly as [B, L*C]
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=torch.float32) if calc_grad_inputs else None
        grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                            interpolation, out_layout=1)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.align_corners = align_corners
        ctx.host_offsets = host_offsets
        ctx.inplace_ok = _INPLACE_OK[0]
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.contiguous().float()
        grad_inputs = torch.empty_like(inputs) if dy_dx is not None else None
        # big batches: XCD-private accumulation of the table gradient (8 copies + one reduce pass beat memory-side atomics)
        import os
        # the slab-binned path from 2048 points (round 5; 16384 before): it is the DETERMINISTIC one (64-bit fixed-point sums), and a training
        # step's table gradient should reproduce bit for bit at every avatar size; below that the handful of float atomics of the
        # device-scope path cost less than the slab path's five dependent launches
        mode = os.environ.get("DWG_GRID_XCD_MODE", "slabs") if B >= int(os.environ.get("DWG_GRID_SLAB_MIN_POINTS", "2048")) else "device"
        if mode in ("owner", "copies") and not xcd_path_ok(inputs.device):
            mode = "device"
        if mode == "slabs" and not slab_path_ok(B, L, int(embeddings.shape[0])):
            mode = "device"
        scratch = xcd_scratch_for(embeddings) if mode == "copies" else None
        slab_ws = slab_workspace_for(inputs.device, B, L, int(embeddings.shape[0])) if mode == "slabs" else None
        # The table is a leaf Parameter whose .grad is its slice of the flat gradient buffer (optim.FlatBuffers marks such parameters:
        # `_dwg_flat`): the slab pass ADDS straight into it and autograd is handed None -- instead of a zeroed 50 MB temporary that autograd
        # then adds to .grad (a 50 MB fill and a 150 MB add per backward).  Opt-in PER PARAMETER: a frozen table, a table whose .grad was
        # rebound, a table of another optimizer or one autograd did not ask a gradient for gets the gradient returned the ordinary way
        # (FlatBuffers.owns_grad).  `torch.autograd.grad(out, [table])` on a flat-buffer table cannot be told from `.backward()` here: such
        # a caller wraps the forward in `table_grad_inplace(False)`.  DWG_GRID_GRAD_INPLACE=0: never in place.
        flat = getattr(embeddings, "_dwg_flat", None)
        in_place = (mode == "slabs" and flat is not None and ctx.needs_input_grad[1] and flat.owns_grad(embeddings)
                    and ctx.inplace_ok and os.environ.get("DWG_GRID_GRAD_INPLACE", "1") != "0")
        grad_embeddings = embeddings.grad if in_place else torch.zeros_like(embeddings)
        counters = xcd_counters_for(inputs.device) if (mode == "owner" and grad_embeddings.data_ptr() % 128 == 0) else None
        try:
            grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                                 gridtype, ctx.align_corners, interpolation, grad_layout=1, xcd_scratch=scratch,
                                 host_offsets=ctx.host_offsets, xcd_counters=counters, slab_workspace=slab_ws, accumulate=in_place)
        except Exception:
            if scratch is not None:
                scratch.zero_()        # a failed launch must not leave partial sums for the next call
            raise
        if in_place:
            flat.touch(embeddings)      # autograd never sees this gradient: record the parameter's participation for the optimizer
        return grad_inputs, (None if in_place else grad_embeddings), None, None, None, None, None, None, None, None


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, host_offsets=None):
    """Same positional signature as the reference's `grid_encode = _grid_encode.apply` (grid.py:96), plus the optional host copy."""
    return _grid_encode.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype,
                              align_corners, interpolation, host_offsets)


class GridEncoder(nn.Module):
    """Same constructor / buffers / parameter names as the reference's GridEncoder (grid.py:99-144)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.log2_hashmap_size, self.base_resolution = per_level_scale, log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id = gridtype, _gridtype_to_id[gridtype]
        self.interpolation, self.interp_id = interpolation, _interp_to_id[interpolation]
        self.align_corners = align_corners
        offsets, offset = [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self._host_offsets_py = tuple(int(o) for o in offsets)     # host copy for the library's LDS-path sizing
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._host_offsets_py = None       # `offsets` may have been replaced: re-read it once on the next forward

    def forward(self, inputs, bound=1, normalized=False):
        """`normalized`: the inputs already are (x + bound) / (2 bound) (avatar.animate normalises while it gathers the rows)."""
        if not normalized:
            inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        if self._host_offsets_py is None:
            self._host_offsets_py = tuple(int(o) for o in self.offsets.cpu().tolist())
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id,
                              _host_array(self._host_offsets_py))
        return outputs.view(prefix_shape + [self.output_dim])
