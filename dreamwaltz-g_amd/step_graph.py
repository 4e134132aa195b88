"""A whole training step as ONE captured HIP graph, replayed per pose: zero_grad -> [condition image of the posed body] -> Scene.forward
(animate -> render) -> diffusion(...) -> backward (VAE, rasterizer, LBS, grid encoder, MLPs, mesh binding) -> fused Adam of every named
optimizer, i.e. the loop body of /root/reference/core/trainer.py:859-890.  Without guidance (BASELINE config c2) it is the avatar side alone;
with it (config c3) the VAE / ControlNet + UNet plans are captured INLINE (their own hipGraphs are switched off: one graph, no nesting) and
the call's three random draws (VAE posterior, timestep, noise: checklist Q12) are made eagerly into static tensors right before each replay,
from the same per-step seed the eager step uses -- the numbers are the eager step's.

Why: the c2 step is ~0.9 ms of kernels and its ~250 launches + autograd bookkeeping cost the host 2-2.5 ms -- the eager loop is host-bound
(DESIGN.md, round-3 verdict).  A replay costs one copy of the pose into static buffers, one 16-byte-per-group copy of the optimizer scalars
and one graph launch.

What is static in the graph: the Gaussian count, the rasterizer's pair capacity (frozen at `grow` x the largest count of the
warm-up steps, as player.GraphedAnimation does; a step that needs more pairs is TRUNCATED by the kernels and flags it -- `check()` reports
it, `recapture()` grows the capacity), the binning order of the rasterizer (results do not depend on it).  What moves per replay without
being captured: the pose (static device buffers, refreshed by an asynchronous copy before the launch), the optimizers' per-step scalars
(learning-rate schedule, bias corrections: `FlatOptimizer.prepare_step` writes them to a pinned table, one copy puts them where
`dwg_adam_step_dev` reads them) and -- round 5, `example_camera` -- the CAMERA: the reference samples a new one every step
(/root/reference/data/camera/__init__.py:124-165, core/trainer.py:840-860), so extrinsic / projection / c2w and the field of view live in the
same device block as the pose (the rasterizer reads the field of view through a pointer: dwg_raster_settings::tanfov) and `step(pose,
camera)` refreshes them with the pose's copy.  Without `example_camera` the camera is the fixed one of the benchmark configurations.  The densifier and multi-view / multi-rank steps are not captured (they change shapes / need the host).
"""
from typing import Dict

import torch

from .rasterizer import PairCapacity


class GraphedTrainStep:
    CAMERA_KEYS = ("extrinsic", "projection", "c2w", "tanfov_dev", "cond_intrinsics")

    def __init__(self, trainer, data: dict, example_pose: Dict[str, torch.Tensor], warmup_poses=None, grow: float = 4.0, capture_pose=None,
                 condition_fn=None, seed_fn=None, example_camera: Dict[str, torch.Tensor] = None, warmup_cameras=None, capture_camera=None):
        """`trainer`: an SDSTrainer whose `diffusion` makes no host round trips inside a call -- the no-guidance image loss of c2, or a
        ControlNetScoreDistillation (recognised by `draw_view_randoms`: its random draws are made here, outside the graph); `data`: the
        loader's dict of the (fixed) camera WITHOUT 'smpl_inputs'; `example_pose`: device tensors (float32), copied into the graph's static
        pose buffers.  `condition_fn(pose) -> [1,3,H,W]`: the loader's condition image of the posed body, drawn inside the graph;
        `seed_fn(step index) -> int`: the seed of the step's device-RNG stream (guidance only; None: the generator just runs on).  Building
        it takes REAL optimizer steps: one per warm-up pose, and one more -- on `capture_pose` (default: the last warm-up pose again) -- at
        the frozen pair capacity right before the capture.  `example_camera`: float32 tensors for (a subset of) CAMERA_KEYS, shaped like the
        loader's -- extrinsic / projection / c2w [1,4,4], tanfov_dev [2] = {tanfovx, tanfovy}, cond_intrinsics [3,3] -- plus optional host
        scalars 'radius' / 'tanfov' (the learning-rate schedule's spatial scale, trainer.py:713): the captured step then renders whatever
        camera `step(pose, camera)` was given (`warmup_cameras` / `capture_camera` pair with the warm-up / capture poses)."""
        self.trainer, self.grow = trainer, float(grow)
        self.device = next(iter(example_pose.values())).device
        if self.device.type != "cuda":
            raise RuntimeError("dreamwaltz_g_amd.step_graph runs on the GPU only (HIP kernels)")
        renderer = trainer.model.renderer
        if not renderer.async_pair_count:
            raise ValueError("GraphedTrainStep needs a renderer with async_pair_count=True (no host read-back inside the step)")
        if trainer.densifiers is not None or trainer.world != 1 or trainer.total_views != 1:
            raise NotImplementedError("a captured step is single-view, single-rank and without the densifier")
        # ONE device block holds everything the host refreshes per replay -- the optimizers' scalar table, then the pose tensors (16-byte
        # aligned views) -- so that a step costs one asynchronous copy from a pinned twin of the block, not one per tensor (nine before)
        opts = trainer.optimizers
        self._rows = sum(len(o.param_groups) for o in opts.values())
        if any(v.dtype != torch.float32 for v in example_pose.values()):
            raise TypeError("GraphedTrainStep: pose tensors must be float32")
        offs, o = {}, self._rows * 4
        for k, v in example_pose.items():
            offs[k] = o
            o += (v.numel() + 3) // 4 * 4
        cam = {k: v for k, v in (example_camera or {}).items() if k in self.CAMERA_KEYS}
        if any(v.dtype != torch.float32 for v in cam.values()):
            raise TypeError("GraphedTrainStep: camera tensors must be float32")
        coffs = {}
        for k, v in cam.items():
            coffs[k] = o
            o += (v.numel() + 3) // 4 * 4
        self._block_floats = o
        self._block_dev = torch.zeros(o, device=self.device)
        self.hyper_dev = self._block_dev[:self._rows * 4].view(self._rows, 4)
        self.pose = {k: self._block_dev[offs[k]:offs[k] + v.numel()].view(v.shape) for k, v in example_pose.items()}
        for k, v in example_pose.items():
            self.pose[k].copy_(v)
        self.data = dict(data); self.data["smpl_inputs"] = self.pose
        # the camera's tensors, when it moves: views of the same block, put where the renderer / the condition image read them
        self.camera = {k: self._block_dev[coffs[k]:coffs[k] + v.numel()].view(v.shape) for k, v in cam.items()}
        for k, v in cam.items():
            self.camera[k].copy_(v)
            self.data[k] = self.camera[k]
        self._cam_pinned = None
        self.condition_fn, self.seed_fn = condition_fn, seed_fn
        self.guided = hasattr(trainer.diffusion, "draw_view_randoms")
        self._rng, self._rand = None, None
        # The view-dependent prompt (trainer._select_text, /root/reference/core/trainer.py:941-955) is chosen ON THE HOST from the camera's
        # azimuth / elevation: a captured step would keep the capture-time embedding for every later camera.  It lives in a static device
        # buffer the captured call reads; `step()` copies the embedding of THIS camera's view into it before the replay.
        self._text_static = None
        if self.guided and trainer.view_prompt is not None and 'viewed' in trainer.text_embeds_dict:
            emb0, _ = trainer._select_text(self.data)
            self._text_static = emb0.detach().clone()
            self._text_index = None
        if self.guided:
            trainer.diffusion.set_use_graphs(False)            # the plans' kernels go into THIS graph, not into graphs of their own
            self._rng = torch.Generator(device=self.device)
            pn, t, n = trainer.diffusion.draw_view_randoms(self._rng, 1, trainer.max_step)
            self._rand = (pn.clone(), t.clone(), n.clone())     # static: refreshed eagerly before every replay
        # per-step host inputs go through NSLOT rotating pinned twins of the block; a slot is rewritten only after the replay that read it has
        # finished (nothing else throttles the host here: the eager loop is paced by the rasterizer's pair-count event, a replay is not)
        self._slots = [torch.zeros(self._block_floats).pin_memory() for _ in range(4)]
        self._pinned = [{k: b[offs[k]:offs[k] + v.numel()].view(v.shape) for k, v in example_pose.items()} for b in self._slots]
        self._cam_pinned = [{k: b[coffs[k]:coffs[k] + v.numel()].view(v.shape) for k, v in cam.items()} for b in self._slots]
        for b in self._cam_pinned:
            for k, v in cam.items():
                b[k].copy_(v)
        self._hyper_slots = [b[:self._rows * 4].view(self._rows, 4) for b in self._slots]
        self._slot = 0
        self._slot_events = [None] * 4
        self.hyper_host = self._hyper_slots[0]
        self.graph, self.loss, self.outputs = None, None, None
        self._side = torch.cuda.Stream(device=self.device)
        H, W = int(data["image_height"]), int(data["image_width"])
        self._hw = (H, W)
        self._spatial_scale = trainer.get_spatial_scale(self.data)           # the camera is static: one host read, not one per step
        # eager warm-up steps ON THE CAPTURE STREAM (autograd's AccumulateGrad nodes, lazy kernel attributes, the allocator, pair counts)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        most = 0
        with torch.cuda.stream(self._side):
            shared = renderer.pair_state(self.device, H, W)
            wposes = list(warmup_poses) if warmup_poses is not None else [example_pose] * 3
            wcams = list(warmup_cameras) if warmup_cameras is not None else [None] * len(wposes)
            for pose, wcam in zip(wposes, wcams):
                self._set_pose_now(pose)
                self._set_camera_now(wcam)
                self._eager_step()
                shared.resolve()
                most = max(most, shared.last_num_pairs)
            self._side.synchronize()
        own = PairCapacity()
        own.cap = max(shared.cap, int(most * self.grow), shared.min_pairs)
        own.frozen = True
        self._state = own
        if capture_pose is not None:
            with torch.cuda.stream(self._side):
                self._set_pose_now(capture_pose)
        if capture_camera is not None:
            with torch.cuda.stream(self._side):
                self._set_camera_now(capture_camera)
        self._capture()

    # -- pieces ------------------------------------------------------------------------------------------------------------
    def _set_pose_now(self, pose):
        for k, v in pose.items():
            self.pose[k].copy_(v, non_blocking=True)

    def _camera_scale(self, camera):
        """The learning-rate schedule's spatial scale of this step's camera (trainer.get_spatial_scale: radius x tanfov, host scalars)."""
        if camera is not None and self.trainer.cfg.render.spatial_scale is None and "radius" in camera and "tanfov" in camera:
            self._spatial_scale = float(camera["radius"].reshape(-1)[0]) * float(camera["tanfov"].reshape(-1)[0])

    def _set_camera_now(self, camera):
        if camera is None:
            return
        for k in self.camera:
            self.camera[k].copy_(camera[k].to(self.device, non_blocking=True).reshape(self.camera[k].shape))
        self._camera_scale(camera)
        self._select_view_text(camera)

    def _select_view_text(self, camera):
        """The prompt embedding of this camera's view -> the static buffer the captured call reads (one [1,77,d] device copy, only when the
        view class changes).  A moving camera of a guided step with text augmentation MUST carry 'azimuth' / 'elevation'."""
        if self._text_static is None:
            return
        if "azimuth" not in camera or "elevation" not in camera:
            raise KeyError("GraphedTrainStep: a guided step with view-dependent prompts needs the camera's 'azimuth' and 'elevation' "
                           "(the prompt is selected from them every step: trainer._select_text)")
        tr = self.trainer
        self.data["azimuth"], self.data["elevation"] = camera["azimuth"], camera["elevation"]
        idx = int(tr.view_prompt(azim=camera["azimuth"], elev=camera["elevation"]).item())
        if idx != self._text_index:
            self._text_static.copy_(tr.text_embeds_dict['viewed'][idx], non_blocking=True)
            self._text_index = idx

    def _host_prepare(self):
        """What the eager trainer does on the host around a step (trainer.py:861-870, 888-890): step index, learning-rate schedule, the
        optimizers' step counts and scalars -> the pinned table."""
        tr = self.trainer
        tr.train_step_index += 1
        scale = self._spatial_scale
        base = 0
        for o in tr.optimizers.values():
            if hasattr(o, "update_learning_rate"):
                o.update_learning_rate(iteration=tr.train_step_index, spatial_scale=scale)
            base += o.prepare_step(self.hyper_host, base)

    def _draw(self):
        """The step's random draws in the reference's order, from the step's own seed, into the static tensors (eager, before the replay)."""
        if not self.guided:
            return
        tr = self.trainer
        if self.seed_fn is not None:
            self._rng.manual_seed(int(self.seed_fn(tr.train_step_index)))
        pn, t, n = tr.diffusion.draw_view_randoms(self._rng, tr.train_step_index, tr.max_step)
        self._rand[0].copy_(pn); self._rand[1].copy_(t); self._rand[2].copy_(n)

    def _device_body(self):
        """The captured region."""
        tr = self.trainer
        if hasattr(tr.optimizers, "buffers") and hasattr(tr.optimizers, "zero_grad"):
            tr.optimizers.zero_grad()                   # one fill of the flat gradient buffer for every named optimizer
        else:
            for o in tr.optimizers.values():
                o.zero_grad()
        forced = {}
        if self.guided:
            forced = dict(posterior_noise=self._rand[0], timestep=self._rand[1], noise=self._rand[2])
            if self.condition_fn is not None:
                # (a moving camera: the condition image is drawn from the graph's own camera tensors)
                self.data["cond_images"] = self.condition_fn(self.pose, self.data) if self.camera else self.condition_fn(self.pose)
        tr._text_override = self._text_static           # (None: the trainer selects as in the eager step)
        try:
            loss, render_outputs, _, _ = tr.train_forward(self.data, **forced)
        finally:
            tr._text_override = None
        tr._backward(loss)
        if hasattr(tr.optimizers, "launch_steps"):
            tr.optimizers.launch_steps(self.hyper_dev)          # every group of every named optimizer: one fused Adam launch
        else:
            base = 0
            for o in tr.optimizers.values():
                base += o.launch_step(self.hyper_dev, base)
        return loss, render_outputs

    def _eager_step(self):
        torch.cuda.current_stream(self.device).synchronize()    # set-up path: the pinned table is free to rewrite
        self._host_prepare()
        self._draw()
        self.hyper_dev.copy_(self.hyper_host, non_blocking=True)
        return self._device_body()

    def _capture(self):
        tr, renderer = self.trainer, self.trainer.model.renderer
        H, W = self._hw
        state = self._state
        state.overflow, state.pending, state.frozen = False, False, True
        for entry in renderer._visit_orders.values():      # the periodic refresh of the binning order must not fall into the captured step
            entry[1] = 0
        # Capture has to run on a side stream, while the parameters' AccumulateGrad nodes were made by the warm-up steps on the current one:
        # autograd warns once per process about that mismatch.  Inside a capture the accumulation is ordered by the capture itself (one
        # stream), so the warning has nothing to report here.
        warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if warn_off is not None:
            warn_off(False)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            key = renderer.pair_state_key(self.device, H, W)
            shared = renderer._pair_states.get(key)
            renderer._pair_states[key] = state
            self._eager_step()                                  # once eagerly at the frozen capacity (allocator warm-up)
            self._side.synchronize()
            self._forget_pose_caches()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self._side):
                self.loss, self.outputs = self._device_body()
            self._forget_pose_caches()
            self._keep = [entry[0] for entry in renderer._visit_orders.values()]
            if shared is not None:
                renderer._pair_states[key] = shared
            else:
                renderer._pair_states.pop(key, None)
        torch.cuda.current_stream(self.device).wait_stream(self._side)

    def _forget_pose_caches(self):
        for m in self.trainer.model.modules():
            if getattr(m, "_last_forward", None) is not None:
                m._last_forward = None

    # -- per step ------------------------------------------------------------------------------------------------------------
    def step(self, pose_cpu: Dict[str, torch.Tensor], camera_cpu: Dict[str, torch.Tensor] = None):
        """One optimizer step for `pose_cpu` (host tensors) [seen by `camera_cpu`: host tensors for the keys of `example_camera`]: returns
        (loss, render outputs) -- static tensors, overwritten by the next step."""
        i = self._slot; self._slot = (self._slot + 1) % len(self._pinned)
        if self._slot_events[i] is not None:
            self._slot_events[i].synchronize()                  # the replay that read this slot four steps ago is done
        self.hyper_host = self._hyper_slots[i]
        if camera_cpu is not None:
            if not self.camera:
                raise ValueError("GraphedTrainStep.step: this step was captured with a fixed camera (no example_camera)")
            self._camera_scale(camera_cpu)                      # before the optimizers' scalars are prepared: the schedule reads it
            self._select_view_text(camera_cpu)
            for k, dst in self._cam_pinned[i].items():
                dst.copy_(camera_cpu[k].reshape(dst.shape))
        elif self.camera:                                       # the camera of the previous step stays: carry it into this slot
            prev = self._cam_pinned[(i - 1) % len(self._cam_pinned)]
            for k, dst in self._cam_pinned[i].items():
                dst.copy_(prev[k])
        self._host_prepare()
        self._draw()
        slot = self._pinned[i]
        if pose_cpu.keys() != slot.keys():
            raise KeyError("GraphedTrainStep.step: the pose must carry exactly the tensors of the example pose (%s), got %s"
                           % (sorted(slot.keys()), sorted(pose_cpu.keys())))
        for k, v in pose_cpu.items():
            slot[k].copy_(v)
        self._block_dev.copy_(self._slots[i], non_blocking=True)        # the scalar table and the pose: one copy
        self.graph.replay()
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.device))
        self._slot_events[i] = ev
        return self.loss, self.outputs

    def check(self) -> bool:
        """True if a step since the last call was truncated by the frozen pair capacity (one stream synchronisation)."""
        torch.cuda.current_stream(self.device).synchronize()
        st = self._state
        ovf = bool(st.truncated_host is not None and int(st.truncated_host[0]) != 0)      # counted on the device over every replay
        if st.host is not None:
            st.last_num_pairs, st.last_num_pairs_ref = int(st.host[0]), int(st.host[2])
        if ovf:
            st.truncated.zero_()
            st.truncated_host.zero_()
        return ovf

    def recapture(self, grow: float = 2.0):
        """After check() reported a truncated step: capture again at `grow` x the pair capacity WITHOUT taking a step.  Capturing needs one
        eager pass at the new capacity (allocator warm-up) and that pass is a whole optimizer step on whatever pose the static buffer holds:
        parameters, Adam moments, every group's step count / learning rate and the trainer's step index are put back afterwards, so the
        step sequence -- poses, seeds, bias corrections -- stays the eager run's (the truncated step itself is the caller's to repeat)."""
        tr = self.trainer
        buf = tr.optimizers.buffers
        snap = (buf.flat.clone(), buf.m.clone(), buf.v.clone(), tr.train_step_index,
                [(o.t, o.current_iteration, [(pg.get("t", 0), pg["lr"]) for pg in o.param_groups]) for o in tr.optimizers.values()])
        self._state.cap = int(self._state.cap * grow)
        self.graph = None
        self._capture()
        torch.cuda.current_stream(self.device).synchronize()
        buf.flat.copy_(snap[0]); buf.m.copy_(snap[1]); buf.v.copy_(snap[2])
        tr.train_step_index = snap[3]
        for o, (t, it, groups) in zip(tr.optimizers.values(), snap[4]):
            o.t, o.current_iteration = t, it
            for pg, (gt, lr) in zip(o.param_groups, groups):
                pg["t"], pg["lr"] = gt, lr
