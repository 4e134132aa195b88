"""Optimizers of the SDS step behind the reference's seam.

The reference builds a dict of named optimizers (`avatar.get_optimizer(cfg)`, /root/reference/core/system/avatar.py:1590-1635):
  'avatar'     GaussianOptimizer (core/gaussian/gaussian_optimizer.py:49-141): Adam(lr=0, eps=1e-15) with groups positions / scales /
               quaternions and `update_learning_rate(spatial_scale, iteration)` (exponential position schedule x spatial_scale,
               scaling_lr x spatial_scale)
  'lbs'        torch.optim.Adam (default eps 1e-8) over _lbs_weights / _betas when learned
  'nerf'       Adam(betas=(0.9, 0.99), eps=1e-15): encoder lr 10 x nerf.lr, both MLPs nerf.lr
  'mesh_<part>' Adam(lr=0, eps=1e-15): bary_coords (position_lr_init), scales (scaling_lr)
and the trainer loops `optimizer.zero_grad()`, `optimizer.update_learning_rate(...)` (when present), `scaler.step(optimizer)` over
them (core/trainer.py:861-890).  Here every one of those objects is a VIEW of one flat fp32 parameter buffer with flat gradient /
first-moment / second-moment buffers next to it: the multi-view step all-reduces ONE tensor over RCCL, and each view's `step()` is
a fused Adam launch per learning-rate group (csrc/elementwise.hip k_adam through include/dwg_elementwise.h).
"""
import ctypes
import math
import struct
from typing import Dict, List, Optional

import torch

from . import _lib


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with an optional eased-in start: mirror of core/optim/optim_utils.py:4-38 (host-side
    float arithmetic; pinned against the reference's own function by tests/test_oracle_golden.py)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        else:
            delay_rate = 1.0
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return helper


class AdamSpec:
    """What one of the reference's optimizers is made of: param groups (+ Adam hyper-parameters); `gaussian` = the
    OptimizationParams of a GaussianOptimizer (adds update_learning_rate)."""

    def __init__(self, groups: List[dict], betas=(0.9, 0.999), eps=1e-8, gaussian: Optional[dict] = None):
        self.groups, self.betas, self.eps, self.gaussian = groups, betas, eps, gaussian


class FlatBuffers:
    """ONE flat fp32 buffer each for parameters, gradients and the two Adam moments (16-byte aligned slices).  The nn.Parameters
    are re-homed into the flat buffers (`.data` and `.grad` become views).

    Participation: with `.grad` always defined (a view of the flat gradient buffer) an optimizer cannot tell a parameter that took no part
    in this step's backward from one whose gradient is zero -- torch.optim.Adam can (`grad is None` after `zero_grad()`: the parameter, its
    moments and its step count stay as they are; gaussian_optimizer.py:93, trainer.py:861-890).  Every parameter therefore carries a
    post-accumulate hook that records that autograd wrote its slice this step (`touch` for kernels that add into the slice themselves: the
    grid table's in-place gradient), and FlatOptimizer.step() leaves un-touched groups alone.  Callers that fill the flat gradient by hand
    (no backward ran: nothing was recorded) get the plain "step everything" behaviour."""

    def __init__(self, params: List[torch.nn.Parameter], device):
        total, self.slices = 0, []
        for p in params:
            self.slices.append((total, p.numel()))
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, device=device)
        self.grad = torch.zeros(total, device=device)
        self.m = torch.zeros(total, device=device)
        self.v = torch.zeros(total, device=device)
        self.tracking = False           # some backward has recorded participation since the last zero_grad
        self._hooks = []
        for p, (off, n) in zip(params, self.slices):
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.grad[off:off + n].view_as(p.data)
            p._dwg_flat = self          # "my .grad is a slice of a flat gradient buffer": what lets a kernel add into it in place
            p._dwg_off = off            # ... namely the one that starts at element `off` (checked by `owns_grad` before any in-place add)
            p._dwg_touched = False
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self.touch))
        self.total = total

    def touch(self, p):
        p._dwg_touched = True
        self.tracking = True

    def owns_grad(self, p) -> bool:
        """True iff a kernel may ADD p's gradient into `p.grad` and report participation through `touch` instead of returning it to autograd:
        p is a trainable leaf this object re-homed and `p.grad` still IS its slice of the flat gradient buffer (same address, shape, dtype) --
        not a tensor a user rebound since, which the fused Adam would never read.  Frozen parameters (requires_grad False) are refused:
        autograd would have dropped their gradient and the optimizer must not see one."""
        g = p.grad
        return (getattr(p, "_dwg_flat", None) is self and p.is_leaf and p.requires_grad and g is not None and g.dtype == torch.float32
                and g.is_contiguous() and g.shape == p.shape and g.data_ptr() == self.grad.data_ptr() + 4 * int(getattr(p, "_dwg_off", -1)))

    def release(self, params):
        """Detach from the parameters (the buffers are being replaced: resize_flat_params)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in params:
            if getattr(p, "_dwg_flat", None) is self:
                p._dwg_flat = None


# Bumped by everything that writes parameters behind autograd's back (the fused Adam launches write the flat buffer through a raw pointer:
# no tensor version counter moves): caches of parameter-derived values key on it (avatar.DreamWaltzG._frozen_key).
PARAM_EPOCH = [0]


class FlatOptimizer:
    """One of the named optimizers, as a view [start, end) of the flat buffers: same surface the trainer uses on the reference's
    objects (param_groups, zero_grad, step, state_dict / load_state_dict, and update_learning_rate for 'avatar')."""

    def __init__(self, name, buf: FlatBuffers, spec: AdamSpec, ranges):
        self.name, self.buf, self.spec = name, buf, spec
        self.param_groups = []
        for g, (start, end) in zip(spec.groups, ranges):
            pg = dict(g)
            pg.update(betas=g.get('betas', spec.betas), eps=g.get('eps', spec.eps), start=start, end=end)
            self.param_groups.append(pg)
        self.start = min(r[0] for r in ranges)
        self.end = max(r[1] for r in ranges)
        self.t = 0
        self.current_iteration = 0
        self.grad_scale = 1.0        # 1 / world size for the multi-view step (mean of the all-reduced sum)
        if spec.gaussian is not None:
            ga = spec.gaussian
            self.num_iterations = ga["iterations"]
            self.default_scaling_lr = ga["scaling_lr"]
            self.position_sheduler_func = get_expon_lr_func(lr_init=ga["position_lr_init"], lr_final=ga["position_lr_final"],
                                                            lr_delay_mult=ga["position_lr_delay_mult"],
                                                            max_steps=ga["position_lr_max_steps"])
            self.update_learning_rate = self._update_learning_rate

    def _update_learning_rate(self, spatial_scale: float, iteration: Optional[int] = None):
        """GaussianOptimizer.update_learning_rate (gaussian_optimizer.py:130-141)."""
        if iteration is None:
            iteration = self.current_iteration
        lr = 0.
        for pg in self.param_groups:
            if pg.get('name') == "positions":
                lr = self.position_sheduler_func(iteration)
                pg['lr'] = lr * spatial_scale
            elif pg.get('name') == "scales":
                lr = self.default_scaling_lr
                pg['lr'] = lr * spatial_scale
        return lr

    def zero_grad(self, set_to_none: bool = False):
        self.buf.grad[self.start:self.end].zero_()
        self.buf.tracking = False
        for pg in self.param_groups:
            for p in pg.get('params', ()):
                p._dwg_touched = False

    def step(self, participation=None):
        """One Adam step of every param group.  Bias correction uses the group's OWN step count (torch.optim.Adam keeps one per parameter):
        a group that sat a step out -- its parameter was just replaced by the densifier, `grad is None` in the reference -- does not age.
        `participation`: one bool per group, decided by the caller (the multi-rank step agrees on it across the ranks, trainer._reduce_and_step)
        instead of this process's own record of which parameters its backward touched."""
        self.t += 1
        self.current_iteration += 1
        PARAM_EPOCH[0] += 1
        for gi, pg in enumerate(self.param_groups):
            if pg.pop("skip_once", False):
                # the densifier just replaced this group's parameter: torch.optim.Adam finds `grad is None` on the new Parameter and leaves
                # it (and its moments) alone for this step (gaussian_densifier.py:141-161 + trainer.py:876-890)
                continue
            if pg["end"] <= pg["start"]:
                continue
            if participation is not None:
                if not participation[gi]:
                    continue
            elif self.buf.tracking and pg.get('params') and not any(getattr(p, "_dwg_touched", False) for p in pg['params']):
                # no parameter of this group took part in this step's backward: torch.optim.Adam sees `grad is None` and leaves the
                # parameter, both moments and the step count untouched (no drift on stale momentum, no ageing of the bias correction)
                continue
            pg["t"] = pg.get("t", 0) + 1
            self._launch(pg)

    # -- the same step in two halves, for a captured graph of the whole avatar-side step (step_graph.GraphedTrainStep) ----------------
    def prepare_step(self, hyper_host: torch.Tensor, base: int):
        """HOST half of step(): step counts and this step's scalars of every group -> rows base, base + 1, ... of the (pinned) hyper table
        {lr / (1 - b1^t), sqrt(1 - b2^t), grad_scale, 0}.  Returns the number of rows used.  (Participation is not consulted: a captured
        step replays the same launches every time.)"""
        self.t += 1
        self.current_iteration += 1
        PARAM_EPOCH[0] += 1
        for k, pg in enumerate(self.param_groups):
            pg["t"] = pg.get("t", 0) + 1
            # the statements of dwg_adam_step (csrc/elementwise.hip): fp32 hyper-parameters, double arithmetic, one rounding -- a replay of the
            # captured step and an eager step then feed k_adam identical bits
            b1, b2, lr = (struct.unpack("f", struct.pack("f", float(x)))[0] for x in (pg["betas"][0], pg["betas"][1], pg["lr"]))
            hyper_host[base + k, 0] = lr / (1.0 - b1 ** pg["t"])
            hyper_host[base + k, 1] = math.sqrt(1.0 - b2 ** pg["t"])
            hyper_host[base + k, 2] = float(self.grad_scale)
        return len(self.param_groups)

    def launch_step(self, hyper_dev: torch.Tensor, base: int):
        """DEVICE half: one fused Adam launch per group reading its scalars from row base + k of the device hyper table (capturable)."""
        b = self.buf
        st = ctypes.c_void_p(torch.cuda.current_stream(b.flat.device).cuda_stream)
        for k, pg in enumerate(self.param_groups):
            n, o = pg["end"] - pg["start"], pg["start"] * 4
            if n <= 0:
                continue
            _lib.check(_lib.lib().dwg_adam_step_dev(n, ctypes.c_void_p(b.flat.data_ptr() + o), ctypes.c_void_p(b.grad.data_ptr() + o),
                                                    ctypes.c_void_p(b.m.data_ptr() + o), ctypes.c_void_p(b.v.data_ptr() + o),
                                                    ctypes.c_void_p(hyper_dev.data_ptr() + (base + k) * 16), float(pg["betas"][0]),
                                                    float(pg["betas"][1]), float(pg["eps"]), st), "dwg_adam_step_dev")
        return len(self.param_groups)

    def _launch(self, pg):
        """The fused Adam launch of one group over its slice of the flat buffers (csrc/elementwise.hip k_adam)."""
        b = self.buf
        n, o = pg["end"] - pg["start"], pg["start"] * 4
        st = ctypes.c_void_p(torch.cuda.current_stream(b.flat.device).cuda_stream)
        _lib.check(_lib.lib().dwg_adam_step(n, ctypes.c_void_p(b.flat.data_ptr() + o), ctypes.c_void_p(b.grad.data_ptr() + o),
                                            ctypes.c_void_p(b.m.data_ptr() + o), ctypes.c_void_p(b.v.data_ptr() + o), float(pg["lr"]),
                                            float(pg["betas"][0]), float(pg["betas"][1]), float(pg["eps"]), int(pg["t"]), float(self.grad_scale), st),
                   "dwg_adam_step")

    def state_dict(self):
        """Same information a torch Adam state_dict carries, flat: step count, per-group hyper-parameters and the two moments."""
        return {"t": self.t, "current_iteration": self.current_iteration,
                "param_groups": [{k: v for k, v in pg.items() if k != 'params'} for pg in self.param_groups],
                "exp_avg": self.buf.m[self.start:self.end].clone(), "exp_avg_sq": self.buf.v[self.start:self.end].clone()}

    def load_state_dict(self, sd):
        self.t, self.current_iteration = int(sd["t"]), int(sd.get("current_iteration", sd["t"]))
        for pg, s in zip(self.param_groups, sd["param_groups"]):
            pg["lr"] = s["lr"]
            pg["t"] = int(s.get("t", sd["t"]))
        self.buf.m[self.start:self.end].copy_(sd["exp_avg"]); self.buf.v[self.start:self.end].copy_(sd["exp_avg_sq"])


class FlatOptimizerDict(dict):
    """The dict `avatar.get_optimizer(cfg)` returns; `.buffers` is the shared flat storage (all-reduce operand)."""
    buffers: FlatBuffers

    def all_grads(self):
        return self.buffers.grad

    def set_grad_scale(self, s: float):
        for o in self.values():
            o.grad_scale = s

    def launch_steps(self, hyper_dev: torch.Tensor):
        """DEVICE half of every named optimizer's step in ONE launch (dwg_adam_step_groups_dev): group k of the dict's k-th optimizer reads
        the row `FlatOptimizer.prepare_step` filled for it (rows are handed out in dict order, as step_graph does).  Falls back to one launch
        per group beyond DWG_ADAM_MAX_GROUPS groups."""
        b = self.buffers
        groups, base = [], 0
        for o in self.values():
            for k, pg in enumerate(o.param_groups):
                n, off = pg["end"] - pg["start"], pg["start"] * 4
                if n > 0:
                    groups.append(_lib.AdamGroupC(b.flat.data_ptr() + off, b.grad.data_ptr() + off, b.m.data_ptr() + off, b.v.data_ptr() + off, n,
                                                  float(pg["betas"][0]), float(pg["betas"][1]), float(pg["eps"]), base + k))
            base += len(o.param_groups)
        if len(groups) > 16:
            base = 0
            for o in self.values():
                base += o.launch_step(hyper_dev, base)
            return base
        arr = (_lib.AdamGroupC * max(len(groups), 1))(*groups)
        st = ctypes.c_void_p(torch.cuda.current_stream(b.flat.device).cuda_stream)
        _lib.check(_lib.lib().dwg_adam_step_groups_dev(len(groups), ctypes.cast(arr, ctypes.c_void_p), ctypes.c_void_p(hyper_dev.data_ptr()), st),
                   "dwg_adam_step_groups_dev")
        return base

    def zero_grad(self):
        """`zero_grad()` of every named optimizer with ONE fill: their gradient ranges are disjoint pieces of the one flat buffer."""
        self.buffers.grad.zero_()
        self.buffers.tracking = False
        for p in self.params:
            p._dwg_touched = False


def _group_ranges(specs: Dict[str, AdamSpec], buf: FlatBuffers) -> Dict[str, list]:
    """[start, end) of every param group inside the flat buffers (16-byte aligned ends), in the order the parameters were laid out."""
    i, out = 0, {}
    for name, spec in specs.items():
        ranges = []
        for g in spec.groups:
            if not g['params']:
                ranges.append((buf.slices[i][0] if i < len(buf.slices) else buf.total,) * 2)       # an empty group owns nothing
                continue
            start = buf.slices[i][0]
            for _ in g['params']:
                off, n = buf.slices[i]
                i += 1
            ranges.append((start, (off + n + 3) // 4 * 4))
        out[name] = ranges
    return out


def build_flat_optimizers(specs: Dict[str, AdamSpec], device) -> FlatOptimizerDict:
    params = [p for spec in specs.values() for g in spec.groups for p in g['params']]
    buf = FlatBuffers(params, device)
    out = FlatOptimizerDict()
    out.buffers, out.specs, out.params = buf, specs, params
    for name, ranges in _group_ranges(specs, buf).items():
        out[name] = FlatOptimizer(name, buf, specs[name], ranges)
        out[name].owner = out                    # the dict this named optimizer is a view of (densifier: the buffers resize as a whole)
    return out


def resize_flat_params(opts: FlatOptimizerDict, new_values: Dict[torch.nn.Parameter, tuple]) -> Dict[torch.nn.Parameter, torch.nn.Parameter]:
    """Densification / pruning (gaussian_densifier.py:120-180): some parameters change their first dimension.  `new_values` maps a Parameter
    to (data, exp_avg, exp_avg_sq) of its new shape (moments None = zeros).  The flat parameter / gradient / moment buffers are laid out
    afresh: the untouched Parameters keep their identity and are re-homed (`.data` / `.grad` become views of the new buffers, moments
    carried over); a RESIZED parameter becomes a NEW nn.Parameter object, as in the reference (:131,156) -- a leaf whose `.data` changed
    shape keeps a stale gradient accumulator for as long as anybody holds an output of an earlier step.  Returns {old: new} for the
    resized ones (the caller re-binds its attributes); every named optimizer gets its new ranges.  The all-reduce operand of the
    multi-view step is the new `opts.buffers.grad`."""
    old = opts.buffers
    keep, renamed = {}, {}
    for p, (off, n) in zip(opts.params, old.slices):
        if p in new_values:
            data, m, v = new_values[p]
            data = data.detach().float().clone()
            q = torch.nn.Parameter(data, requires_grad=p.requires_grad)
            renamed[p] = q
            keep[q] = (data, torch.zeros_like(data) if m is None else m.detach().float(), torch.zeros_like(data) if v is None else v.detach().float(), None)
        else:
            keep[p] = (p.data.clone(), old.m[off:off + n].clone(), old.v[off:off + n].clone(), old.grad[off:off + n].clone())
    opts.params = [renamed.get(p, p) for p in opts.params]
    for spec in opts.specs.values():
        for g in spec.groups:
            g['params'] = [renamed.get(p, p) for p in g['params']]
    for o in opts.values():
        for pg in o.param_groups:
            if 'params' in pg:
                pg['params'] = [renamed.get(p, p) for p in pg['params']]
    old.release(list(keep.keys()))
    for p in opts.params:
        p.data = keep[p][0]
        p.grad = None
    touched = {p: bool(getattr(p, "_dwg_touched", False)) for p in opts.params}      # this step's participation survives the re-homing
    buf = FlatBuffers(opts.params, old.flat.device)
    buf.tracking = old.tracking
    for p in opts.params:
        p._dwg_touched = touched[p]
    for p, (off, n) in zip(opts.params, buf.slices):
        buf.m[off:off + n].copy_(keep[p][1].reshape(-1)); buf.v[off:off + n].copy_(keep[p][2].reshape(-1))
        if keep[p][3] is not None:
            buf.grad[off:off + n].copy_(keep[p][3])
    opts.buffers = buf
    resized = set(renamed.values())
    for name, ranges in _group_ranges(opts.specs, buf).items():
        o = opts[name]
        o.buf = buf
        for pg, g, (start, end) in zip(o.param_groups, o.spec.groups, ranges):
            pg["start"], pg["end"] = start, end
            if any(p in resized for p in g['params']):
                pg["skip_once"] = True
        o.start = min(r[0] for r in ranges)
        o.end = max(r[1] for r in ranges)
    return renamed
