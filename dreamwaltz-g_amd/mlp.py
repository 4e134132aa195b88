"""Per-Gaussian MLPs of the avatar on the f32-exact MFMA GEMM (csrc/gemm.hip), with hand-written autograd.

Class / parameter names mirror the reference so its state_dicts load unchanged:
  MLP            /root/reference/core/nerf/nerf_model.py:12-33           (`net.{l}.weight/bias`, ReLU between layers)
  DeformNetwork  /root/reference/core/deformation/deform_model.py:61-143 (`layers.{i}`, `gaussian_warp`,
                 `gaussian_rotation`, `gaussian_scaling`; leaky-ReLU; xyz_input_ch given -> identity embedding;
                 D=4, W=64, no skips, is_6dof=False -- the configuration DreamWaltzG builds at avatar.py:1171-1174)
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from . import _lib, gemm


# DWG_MLP_BWD_PER_LAYER=1: the layer-by-layer backward of a chain (four launches per layer) instead of the fused kernel -- experiments only
PER_LAYER_BACKWARD = os.environ.get("DWG_MLP_BWD_PER_LAYER", "0") == "1"


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _inplace_allowed():
    from . import gridencoder
    return bool(gridencoder._INPLACE_OK[0]) and os.environ.get("DWG_MLP_GRAD_INPLACE", "1") != "0"


def _flat_slice(p, used, needed=True):
    """The FlatBuffers object whose gradient buffer `p.grad` is a slice of, if the backward may add into it directly: autograd asked for
    this input's gradient (`needed` = the matching ctx.needs_input_grad entry), `p` is a TRAINABLE leaf Parameter optim.FlatBuffers
    re-homed whose .grad still is that slice (FlatBuffers.owns_grad: address, shape and dtype -- a frozen parameter, or a .grad the user
    rebound, gets the ordinary returned gradient), and (weights) the tensor the kernels read IS the parameter (no dtype / layout copy)."""
    flat = getattr(p, "_dwg_flat", None) if p is not None else None
    if flat is None or not needed or not flat.owns_grad(p):
        return None
    if used is not None and (used.data_ptr() != p.data_ptr() or used.shape != p.shape):
        return None
    return flat


def _splitk(M):
    return max(1, min(256, (M + 1023) // 1024))


class _Linear(torch.autograd.Function):
    """y = act(x @ w[:, :Kx]^T + c), c = b + w[:, Kx:] @ extra  (extra: a vector broadcast to every row, e.g. the body pose
    that deform_model.py:113-115 expands and concatenates)."""

    @staticmethod
    def forward(ctx, x, w, b, extra, act):
        x = x.contiguous().float()
        Kx = x.shape[1]
        wc = w.contiguous().float()
        bias = b.float() if b is not None else torch.zeros(w.shape[0], device=x.device)
        if extra is not None:
            e = extra.reshape(1, -1).contiguous().float()
            # c = extra @ w[:, Kx:]^T + b   (1 x Ke) x (Ke x N)
            c = torch.empty(1, w.shape[0], device=x.device)
            gemm.gemm_raw(e, wc[:, Kx:], c, 1, w.shape[0], e.shape[1], (e.shape[1], 1), (wc.stride(0), 1), w.shape[0],
                          bias=bias.contiguous(), name="mlp_pose_bias")
            bias = c.reshape(-1)
        y = torch.empty(x.shape[0], w.shape[0], device=x.device)
        gemm.gemm_raw(x, wc, y, x.shape[0], w.shape[0], Kx, (Kx, 1), (wc.stride(0), 1), w.shape[0], bias=bias.contiguous(),
                      act=act, name="mlp_fwd")
        ctx.save_for_backward(x, wc, y, extra)
        ctx.act = act
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, extra = ctx.saved_tensors
        dx, dw, db = _linear_backward(x, w, y, extra, ctx.act, dy, ctx.needs_input_grad[0])
        return dx, dw, (db if ctx.has_b else None), None, None


def _linear_backward(x, w, y, extra, act, dy, need_dx, zeros=None):
    """Gradients of y = act(x @ w[:, :Kx]^T + b + w[:, Kx:] @ extra) given dy and the saved OUTPUT y -> (dx | None, dw, db).
    `zeros`: optional (colsum [N], dw like w) views of a buffer the caller zero-filled once for a whole chain."""
    M, Kx = x.shape
    N = w.shape[0]
    dy = dy.contiguous().float()
    dz = torch.empty_like(dy) if act else dy
    colsum = torch.zeros(N, device=x.device) if zeros is None else zeros[0]
    p = _lib.ptr
    _lib.check(_lib.lib().dwg_act_backward_colsum(M, N, gemm.ACT[act], p(dy), p(y) if act else None,
                                                  p(dz) if act else None, p(colsum), _st(x)), "dwg_act_backward_colsum")
    dw = torch.zeros_like(w) if zeros is None else zeros[1]
    # dW[:, :Kx] = dz^T x   (contraction over the M rows of two row-major operands)
    if N <= 64 and Kx <= 64:
        L = _lib.lib()
        ws = torch.empty(L.dwg_mlp_wgrad_workspace_floats(M), device=x.device, dtype=torch.float32)
        _lib.check(L.dwg_mlp_wgrad(M, N, Kx, p(dz), N, p(x), Kx, p(dw), w.shape[1], p(ws), _st(x)), "dwg_mlp_wgrad")
    else:
        gemm.gemm_raw(dz, x, dw, N, Kx, M, (1, N), (1, Kx), w.shape[1], splitk=_splitk(M), name="mlp_wgrad")
    if extra is not None:
        dw[:, Kx:] = colsum[:, None] * extra.reshape(1, -1)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        gemm.gemm_raw(dz, w, dx, M, Kx, N, (N, 1), (1, w.stride(0)), Kx, name="mlp_dgrad")
    return dx, dw, colsum


class _MlpChain(torch.autograd.Function):
    """The whole MLP in one forward launch (csrc/elementwise.hip k_mlp_chain, include/dwg_elementwise.h dwg_mlp_chain_forward) and one
    backward launch + reduce (k_mlp_chain_bwd, dwg_mlp_chain_backward; DWG_MLP_BWD_PER_LAYER=1: layer by layer with the kernels of _Linear).  args: x, extra (vector folded into the first bias, or None), acts (tuple of
    names, one per layer), then w_0, b_0, w_1, b_1, ..."""

    @staticmethod
    def forward(ctx, x, extra, acts, grad_on, *wb):
        L = _lib.lib()
        nl = len(acts)
        ws = [wb[2 * l].contiguous().float() for l in range(nl)]
        bs = [wb[2 * l + 1] for l in range(nl)]
        x = x.contiguous().float()
        M, Kx = x.shape
        dev = x.device
        # `extra` (the body pose every row is extended by) meets the trailing columns of w_0 inside the launch: a bias
        e = extra.reshape(-1).contiguous().float() if extra is not None else None
        biases = [None if b is None else b.contiguous().float() for b in bs]
        widths = [int(w.shape[0]) for w in ws]
        # `grad_on`: the caller's grad mode (inside forward() it is always off, and needs_input_grad reflects requires_grad of the inputs even
        # under no_grad / inference_mode): without it every inference frame wrote -- and the kernel kept -- the hidden activations of both
        # networks (460 MB per 300 k-Gaussian frame) for a backward that never comes
        keep = bool(grad_on) and any(ctx.needs_input_grad)
        hidden = [torch.empty(M, widths[l], device=dev) if keep else None for l in range(nl - 1)]
        out = torch.empty(M, widths[-1], device=dev)
        vp, i32 = ctypes.c_void_p * nl, ctypes.c_int32 * nl
        pv = lambda t: None if t is None else t.data_ptr()       # noqa: E731
        _lib.check(L.dwg_mlp_chain_forward(M, Kx, _lib.ptr(x), Kx, nl, vp(*[w.data_ptr() for w in ws]), i32(*[int(w.stride(0)) for w in ws]),
                                           vp(*[pv(b) for b in biases]), i32(*widths), i32(*[gemm.ACT[a] for a in acts]),
                                           vp(*[pv(h) for h in hidden] + [None]), _lib.ptr(out), widths[-1],
                                           _lib.ptr(e) if e is not None else None, int(e.numel()) if e is not None else 0, _st(x)),
                   "dwg_mlp_chain_forward")
        ctx.acts, ctx.nl = acts, nl
        ctx.has_b = [b is not None for b in bs]
        ctx.has_extra = extra is not None
        # the parameters themselves (not copies) and whether this forward may add into their flat-gradient slices in its backward: several
        # backwards running concurrently on their own streams (the views of a batched step) must not read-add-write one slice -- the switch the
        # grid encoder's in-place table gradient obeys (gridencoder.table_grad_inplace)
        ctx.params = [(wb[2 * l], wb[2 * l + 1]) for l in range(nl)]
        ctx.inplace_ok = _inplace_allowed()
        if keep:
            ctx.save_for_backward(x, out, *(ws + hidden + ([extra] if extra is not None else [])))
        return out

    @staticmethod
    def backward(ctx, dy):
        nl = ctx.nl
        saved = ctx.saved_tensors
        x, out = saved[0], saved[1]
        ws, hidden = list(saved[2:2 + nl]), list(saved[2 + nl:2 + nl + nl - 1])
        extra = saved[-1] if ctx.has_extra else None
        grads = [None] * (2 * nl)
        if not PER_LAYER_BACKWARD:
            # the whole chain in one launch + one reduce (csrc/elementwise.hip k_mlp_chain_bwd, include/dwg_elementwise.h dwg_mlp_chain_backward)
            L = _lib.lib()
            M, Kx = x.shape
            dev = x.device
            dy = dy.contiguous().float()
            need_dx = ctx.needs_input_grad[0]
            dx = torch.empty_like(x) if need_dx else None
            if M == 0:
                for l, w in enumerate(ws):
                    grads[2 * l] = torch.zeros_like(w)
                    grads[2 * l + 1] = torch.zeros(w.shape[0], device=dev) if ctx.has_b[l] else None
                return (dx, None, None, None) + tuple(grads)
            # A weight / bias that is a leaf Parameter whose .grad is its slice of a flat gradient buffer (optim.FlatBuffers marks those:
            # `_dwg_flat`) gets its gradient ADDED into that slice by the reduce kernel and autograd is handed None -- no temporary, no
            # AccumulateGrad `add_` launch per parameter (sixteen per step for the two networks).  Anything else (the concatenated heads of
            # the deformation network, frozen parameters, inputs autograd did not ask a gradient for, parameters of another optimizer or
            # with a rebound .grad) gets its gradient returned as usual.  NOT detectable here: `torch.autograd.grad(out, [weight])` on a
            # flat-buffer parameter -- that caller would receive None while the slice is added to; it must wrap the FORWARD in
            # `gridencoder.table_grad_inplace(False)` (what the concurrent multi-view backwards do, trainer.py).
            nig = ctx.needs_input_grad
            wflat = [_flat_slice(ctx.params[l][0], ws[l], nig[4 + 2 * l]) if ctx.inplace_ok else None for l in range(nl)]
            bflat = [_flat_slice(ctx.params[l][1], None, nig[5 + 2 * l]) if (ctx.inplace_ok and ctx.has_b[l]) else None for l in range(nl)]
            dws = [ctx.params[l][0].grad if wflat[l] is not None else torch.empty_like(ws[l]) for l in range(nl)]
            dbs = torch.empty(nl, 64, device=dev)
            dbp = [ctx.params[l][1].grad.data_ptr() if bflat[l] is not None else dbs[l].data_ptr() for l in range(nl)]
            acc = [(1 if wflat[l] is not None else 0) | (2 if bflat[l] is not None else 0) for l in range(nl)]
            e = extra.reshape(-1).contiguous().float() if extra is not None else None
            wsp = torch.empty(L.dwg_mlp_chain_backward_workspace_floats(M, nl), device=dev, dtype=torch.float32)
            vp, i32 = ctypes.c_void_p * nl, ctypes.c_int32 * nl
            _lib.check(L.dwg_mlp_chain_backward(
                M, Kx, _lib.ptr(x), Kx, nl, vp(*[w.data_ptr() for w in ws]), i32(*[int(w.stride(0)) for w in ws]),
                i32(*[int(w.shape[0]) for w in ws]), i32(*[gemm.ACT[a] for a in ctx.acts]), vp(*[h.data_ptr() for h in hidden] + [None]),
                _lib.ptr(out), int(out.shape[1]), _lib.ptr(dy), int(dy.shape[1]), _lib.ptr(dx) if need_dx else None, Kx,
                vp(*[d.data_ptr() for d in dws]), i32(*[int(d.stride(0)) for d in dws]), vp(*dbp), i32(*acc),
                _lib.ptr(e) if e is not None else None, int(e.numel()) if e is not None else 0, _lib.ptr(wsp), _st(x)), "dwg_mlp_chain_backward")
            for l, w in enumerate(ws):
                if wflat[l] is not None:
                    wflat[l].touch(ctx.params[l][0])        # autograd never sees this gradient: record the participation for the optimizer
                else:
                    grads[2 * l] = dws[l]
                if bflat[l] is not None:
                    bflat[l].touch(ctx.params[l][1])
                elif ctx.has_b[l]:
                    grads[2 * l + 1] = dbs[l, :w.shape[0]]
            return (dx, None, None, None) + tuple(grads)
        g = dy
        # one zero fill for every layer's bias-gradient accumulator and weight-gradient tile (16 fills per step before)
        sizes = [(int(w.shape[0]), int(w.numel())) for w in ws]
        flat = torch.zeros(sum(a + b for a, b in sizes), device=x.device)
        zs, o = [], 0
        for (a, b), w in zip(sizes, ws):
            zs.append((flat[o:o + a], flat[o + a:o + a + b].view_as(w))); o += a + b
        for l in range(nl - 1, -1, -1):
            xin = x if l == 0 else hidden[l - 1]
            y = out if l == nl - 1 else hidden[l]
            w = ws[l]
            wl = w if (l > 0 or extra is None) else w          # layer 0 keeps the pose columns (their gradient comes from colsum)
            dx, dw, db = _linear_backward(xin, wl, y, extra if l == 0 else None, ctx.acts[l], g, l > 0 or ctx.needs_input_grad[0], zeros=zs[l])
            grads[2 * l] = dw
            grads[2 * l + 1] = db if ctx.has_b[l] else None
            g = dx
        return (g if ctx.needs_input_grad[0] else None, None, None, None) + tuple(grads)


def mlp_chain(x, layers, acts, extra=None):
    """layers: [(weight, bias | None), ...]; acts: activation name per layer.  Falls back to the per-layer path for widths > 64."""
    if any(w.shape[0] > 64 for w, _ in layers) or x.shape[1] > 64 or x.shape[1] % 8 or any(w.shape[0] % 8 for w, _ in layers[:-1]) \
            or len(layers) > 6 or not x.is_cuda:
        h = x
        for l, ((w, b), a) in enumerate(zip(layers, acts)):
            h = linear(h, w, b, act=a, extra=extra if l == 0 else None)
        return h
    flat = []
    for w, b in layers:
        flat += [w, b]
    return _MlpChain.apply(x, extra, tuple(acts), torch.is_grad_enabled(), *flat)


def linear(x, w, b=None, act=None, extra=None):
    return _Linear.apply(x, w, b, extra, act)


class MLP(nn.Module):
    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        net = []
        for l in range(num_layers):
            net.append(nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=bias))
        self.net = nn.ModuleList(net)

    def forward(self, x):
        return mlp_chain(x, [(m.weight, m.bias) for m in self.net], ["relu"] * (self.num_layers - 1) + [None])


class _PackRows(torch.autograd.Function):
    """Several tensors stacked along dim 0 into ONE (weights) and their 1-D companions into another (biases), one launch (dwg_copy_segments)
    -- the three output heads of the deformation network as the single 64 -> 10 layer mlp_chain multiplies.  Backward: a piece that is a
    flat-buffer Parameter gets its rows of the packed gradient ADDED into its gradient slice by one launch for all such pieces
    (dwg_add_segments; autograd sees None and the participation is recorded), any other piece gets its rows back as a view."""

    @staticmethod
    def forward(ctx, n, *pieces):
        ws, bs = pieces[:n], pieces[n:]
        dev = ws[0].device
        W = torch.empty((sum(int(w.shape[0]) for w in ws),) + tuple(ws[0].shape[1:]), device=dev)
        B = torch.empty(sum(int(b.shape[0]) for b in bs), device=dev)
        segs, ow, ob = [], 0, 0
        keep = []
        for w in ws:
            wc = w.contiguous().float(); keep.append(wc)
            segs.append(_lib.SegmentC(W.data_ptr() + 4 * ow, wc.data_ptr(), wc.numel())); ow += wc.numel()
        for b in bs:
            bc = b.contiguous().float(); keep.append(bc)
            segs.append(_lib.SegmentC(B.data_ptr() + 4 * ob, bc.data_ptr(), bc.numel())); ob += bc.numel()
        arr = (_lib.SegmentC * len(segs))(*segs)
        _lib.check(_lib.lib().dwg_copy_segments(len(segs), ctypes.cast(arr, ctypes.c_void_p), 0.0, 0.0, _st(W)), "dwg_copy_segments")
        ctx.pieces, ctx.n, ctx.inplace_ok = pieces, n, _inplace_allowed()
        return W, B

    @staticmethod
    def backward(ctx, gW, gB):
        n, pieces = ctx.n, ctx.pieces
        out, segs, touched = [None] * len(pieces), [], []
        gW = None if gW is None else gW.contiguous().float()
        gB = None if gB is None else gB.contiguous().float()
        ow = ob = 0
        for i, p in enumerate(pieces):
            g, off = (gW, ow) if i < n else (gB, ob)
            cnt = p.numel()
            if g is not None and ctx.needs_input_grad[1 + i]:
                flat = _flat_slice(p, None, True) if ctx.inplace_ok else None
                if flat is not None:
                    segs.append(_lib.SegmentC(p.grad.data_ptr(), g.data_ptr() + 4 * off, cnt)); touched.append((flat, p))
                else:
                    out[i] = g.reshape(-1)[off:off + cnt].view(p.shape)
            if i < n:
                ow += cnt
            else:
                ob += cnt
        if segs:
            arr = (_lib.SegmentC * len(segs))(*segs)
            _lib.check(_lib.lib().dwg_add_segments(len(segs), ctypes.cast(arr, ctypes.c_void_p), _st(gW if gW is not None else gB)), "dwg_add_segments")
            for flat, p in touched:
                flat.touch(p)
        return (None,) + tuple(out)


class DeformNetwork(nn.Module):
    def __init__(self, xyz_input_ch=32, pose_input_ch=63, D=4, W=64, multires=10, residual=False, is_6dof=False):
        super().__init__()
        if residual or is_6dof or xyz_input_ch is None:
            raise NotImplementedError("only the configuration DreamWaltzG uses (avatar.py:1171-1174) is on the hot path")
        self.D, self.W = D, W
        self.input_ch = xyz_input_ch + pose_input_ch
        self.xyz_input_ch = xyz_input_ch
        self.layers = nn.ModuleList([nn.Linear(self.input_ch, W)] + [nn.Linear(W, W) for _ in range(D - 1)])
        self.gaussian_warp = nn.Linear(W, 3)
        self.gaussian_rotation = nn.Linear(W, 4)
        self.gaussian_scaling = nn.Linear(W, 3)
        self._head_cache = None

    def forward(self, x, body_pose, packed=False):
        """-> (warp, scaling, rotation) as the reference's DeformNetwork returns them, or -- `packed` -- the one [N, 10] tensor they are
        column blocks of (assemble.assemble_packed reads the blocks in place)."""
        # the three heads share one 64 -> 10 product (warp 3 | scaling 3 | rotation 4)
        heads = (self.gaussian_warp, self.gaussian_scaling, self.gaussian_rotation)
        if torch.is_grad_enabled() or (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            # (a captured frame re-runs the concatenation from the live parameters on every replay)
            if x.is_cuda:
                w, b = _PackRows.apply(3, *([h.weight for h in heads] + [h.bias for h in heads]))     # one launch; gradients straight into the flat slices
            else:
                w = torch.cat([h.weight for h in heads], 0)
                b = torch.cat([h.bias for h in heads], 0)
        else:
            # inference: the concatenated head is kept while the six tensors are the ones it was built from (address, version counter and the
            # optimizers' write epoch -- the fused Adam writes parameters through a raw pointer): two launches fewer per frame
            from . import optim
            key = (optim.PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for h in heads for t in (h.weight, h.bias))
            if self._head_cache is None or self._head_cache[0] != key:
                self._head_cache = (key, torch.cat([h.weight for h in heads], 0).detach(), torch.cat([h.bias for h in heads], 0).detach())
            w, b = self._head_cache[1], self._head_cache[2]
        o = mlp_chain(x, [(m.weight, m.bias) for m in self.layers] + [(w, b)], ["leaky_relu"] * self.D + [None], extra=body_pose)
        return o if packed else (o[:, 0:3], o[:, 3:6], o[:, 6:10])
