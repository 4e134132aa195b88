"""Animation playback (BASELINE config c5: `Trainer.full_eval` / `evaluate`, /root/reference/core/trainer.py:1019-1150 -- per frame
`Scene.forward(data, smpl_observed_inputs=pose)` under inference mode) with the whole frame -- skeleton pass, LBS, grid encoder, MLPs,
mesh binding, rasterizer -- captured ONCE into a HIP graph and replayed per pose.

Why: at 300 k Gaussians / 1024^2 a frame is ~1.5 ms of kernels, and the ~60 launches + tensor bookkeeping of the eager path cost the host
about as much: the eager loop is host-bound.  A replay costs one copy of the pose into static buffers and one graph launch.

What is static in the graph: the camera, the Gaussian count and the pair-buffer capacity of the rasterizer (frozen at 1.5x the largest
count of the warm-up frames, at least the running capacity).  A frame that needs more pairs than that is TRUNCATED by the kernels and
flags it (in the graph's OWN pair state: eager frames through the same Scene keep the renderer's); `GraphedAnimation.check()` (one stream synchronisation) reports it, and `recapture()` grows the capacity.  Callers that cannot
tolerate a truncated frame call check() per frame -- that still skips all the per-frame host work."""
from typing import Dict, Iterable, Optional

import torch


class GraphedAnimation:
    def __init__(self, scene, data: dict, example_pose: Optional[Dict[str, torch.Tensor]], warmup_poses: Optional[Iterable[dict]] = None,
                 bg_mode: Optional[str] = None):
        """`example_pose` None: the canonical-pose frame (Scene.forward without observed pose: BASELINE config c1) -- no per-frame input at
        all, a replay is one graph launch."""
        self.scene, self.data, self.bg_mode = scene, data, bg_mode
        self.device = next(iter(example_pose.values())).device if example_pose is not None else torch.device(data['extrinsic'].device)
        if self.device.type != "cuda":
            raise RuntimeError("dreamwaltz_g_amd.player runs on the GPU only (HIP kernels)")
        if not scene.renderer.async_pair_count:
            raise ValueError("GraphedAnimation needs a renderer with async_pair_count=True (no host read-back inside the frame)")
        self.pose = {k: v.clone() for k, v in example_pose.items()} if example_pose is not None else None      # static inputs of the graph
        self.graph, self.outputs = None, None
        self._state = None
        self._capture(list(warmup_poses) if warmup_poses is not None else [example_pose] * (1 if example_pose is not None else 3))

    def _frame(self):
        with torch.inference_mode():
            return self.scene.forward(self.data, smpl_observed_inputs=self.pose, use_densifier=False, bg_mode=self.bg_mode)

    def _capture(self, warmup_poses, grow: float = 1.5, min_cap: int = 0):
        from .rasterizer import PairCapacity
        H, W = int(self.data["image_height"]), int(self.data["image_width"])
        shared = self.scene.renderer.pair_state(self.device, H, W)           # the renderer's own state: eager frames keep using it
        most = 0
        for pose in warmup_poses:                                             # eager frames: caches, lazy kernel attributes, pair counts
            self.set_pose(pose)
            self._frame()
            shared.resolve()
            most = max(most, shared.last_num_pairs)
        # The graph gets its OWN frozen PairCapacity (fixed capacity, no events, own pinned count / overflow words): eager frames rendered
        # through the same Scene while the graph is alive keep the renderer's state with overflow detection and head-room growth on.
        own = PairCapacity()
        own.cap = max(shared.cap, int(most * grow), shared.min_pairs, int(min_cap))
        own.frozen = True
        self._state, self._state_key = own, (self.scene.renderer, (H, W))
        self._capture_frozen()

    def _forget_pose_caches(self):
        # the skeleton pass remembers its last result per input tensors and versions (avatar.GeneralLinearBlendSkinning.forward): the
        # captured frame must RUN it, not find the eager frame's tensors
        for m in self.scene.modules():
            if getattr(m, "_last_forward", None) is not None:
                m._last_forward = None

    def _capture_frozen(self):
        """Capture one frame at the pair capacity the state holds now."""
        state = self._state
        state.overflow, state.pending, state.frozen = False, False, True
        renderer, (H, W) = self._state_key
        for entry in self.scene.renderer._visit_orders.values():               # the renderer's periodic refresh of the binning order must not
            entry[1] = 0                                                       # fall into the two frames below (it would be replayed per frame)
        # The captured kernels read the binning order through its DEVICE POINTER.  Eager frames through the same renderer replace the
        # renderer's entry every `reorder_every` calls; the graph keeps its own references so that the tensor it points at can never go
        # back to the allocator while the graph is alive (released by close() / recapture()).
        self._keep = None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            # the key the renderer computes for frames issued on THIS stream (it carries the stream once a multi-view step has switched the
            # renderer to per-stream states): asked of the renderer, not rebuilt here
            key = renderer.pair_state_key(self.device, H, W)
            shared = renderer._pair_states.get(key)
            renderer._pair_states[key] = state                                 # only while the two frames below are issued
            self._frame()                                                      # once eagerly at the frozen capacity (allocator warm-up)
            side.synchronize()
            self._forget_pose_caches()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.outputs = self._frame()
            self._forget_pose_caches()                                         # nothing outside the graph may alias its pool
        self._keep = [entry[0] for entry in self.scene.renderer._visit_orders.values()]
        if shared is not None:
            renderer._pair_states[key] = shared
        else:
            renderer._pair_states.pop(key, None)
        torch.cuda.current_stream(self.device).wait_stream(side)

    def set_pose(self, pose: Optional[Dict[str, torch.Tensor]]):
        if pose is None or self.pose is None:
            return
        for k, v in pose.items():
            self.pose[k].copy_(v, non_blocking=True)

    def replay(self, pose: Optional[Dict[str, torch.Tensor]] = None) -> dict:
        """One frame: the outputs dict of Scene.forward (static tensors, overwritten by the next replay)."""
        self.set_pose(pose)
        self.graph.replay()
        return self.outputs

    def check(self) -> bool:
        """True if the last replayed frame was complete (waits for the stream)."""
        torch.cuda.current_stream(self.device).synchronize()
        return int(self._state.host[1]) == 0

    @property
    def last_num_pairs(self):
        return int(self._state.host[0]), int(self._state.host[2])

    def recapture(self, warmup_poses):
        """After check() returned False: capture again with a capacity grown from fresh eager frames."""
        self.graph, self.outputs, self._keep = None, None, None
        self._capture(list(warmup_poses), grow=2.0, min_cap=self._state.cap * 2)

    def close(self):
        """Drop the graph, its frozen pair state and its references to the binning-order tensors."""
        self.graph, self.outputs, self._keep = None, None, None
        self._state = None
