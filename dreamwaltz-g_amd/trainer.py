"""The loop body of the reference's Trainer.train() for the 3DGS stage, on the HIP path (boundary B5):

  render(data, bg_mode)      /root/reference/core/trainer.py:680-709     -> Scene.forward
  get_spatial_scale(data)    trainer.py:711-716
  train_forward(data)        trainer.py:933-1017   render -> NCHW -> view-dependent text embedding -> diffusion(**sd_kwargs) -> loss
  train_step(data)           trainer.py:859-890    zero_grad + update_learning_rate on every optimizer; forward; backward [x views of
                                                   this rank, accumulated; ONE all-reduce of the flat gradient buffer]; step on every
                                                   optimizer

Method and dictionary-key names are the reference's.  What is added: the flat-buffer all-reduce of the multi-view step
(nothing in the reference to mirror) and the same-frame recovery of a pair-capacity overflow in the sync-free mode.
"""
import os
from typing import Any, Dict, Optional

import torch

from .text import TextAugmentation


class SDSTrainer:
    def __init__(self, cfg, model, diffusion, optimizers, text_embeds_dict: Optional[dict] = None, use_controlnet: bool = True,
                 dist=None, world: int = 1, max_step: Optional[int] = None, densifiers: Optional[dict] = None):
        self.cfg, self.model, self.diffusion, self.optimizers = cfg, model, diffusion, optimizers
        self.densifiers = densifiers                # {'avatar': GaussianDensifier} when cfg.render.use_densifier (trainer.py:600-603), else None
        self.text_embeds_dict = text_embeds_dict if text_embeds_dict is not None else {}
        self.view_prompt = TextAugmentation(cfg.guide.text, cfg.prompt) if cfg.prompt.text_augmentation else None
        self.use_controlnet = use_controlnet
        self.dist, self.world = dist, world
        self.train_step_index = 0
        self.max_step = cfg.optim.iters if max_step is None else max_step
        self.scaler = None                          # GradScaler(enabled=False) in the fp32 recipes: pass-through
        self.redone_frames = 0
        self._view_rng = None
        self._view_streams, self._views_warm = [], False
        self._unit = None
        self._cache_generation = getattr(getattr(model, "avatar", None), "cache_generation", None)
        self.past_checkpoints = []
        self.set_views(world)                       # one view per rank unless the caller says otherwise: mean of the summed gradients,
                                                    # folded into the Adam kernel
        # SURVEY 8e replica-consistency guard (nothing in the reference to mirror: it is single-GPU).  Replicas start from rank 0's
        # parameters, and every `replica_check_every` steps a 64-bit checksum of the flat parameter buffer is compared across the ranks
        # (one all-reduce of two int64 words): a drift -- a non-deterministic reduction order, a rank that missed an update -- raises on
        # every rank instead of training on silently diverged replicas.  DWG_REPLICA_CHECK_EVERY=0 turns the check off.
        self.replica_check_every = int(os.environ.get("DWG_REPLICA_CHECK_EVERY", "100"))
        self._allreduce_events, self.allreduce_ms_total, self.allreduce_steps = [], 0.0, 0      # exchange-step timing (world > 1)
        if self.world > 1:
            self.sync_replicas()

    # -- checkpoints (trainer.py:188-259): {'train_step', 'checkpoints', 'model': Scene.state_dict()[, 'optimizers', 'scaler']} ------
    def save_checkpoint(self, ckpt_dir, full: bool = False, max_keep_ckpts: int = 2):
        """File name step_{train_step:06d}.pth, rolling window of `max_keep_ckpts` files, as the reference writes them; `full` adds
        the optimizer states (here: step counts, learning rates and the flat Adam moments of every named optimizer)."""
        import os
        os.makedirs(ckpt_dir, exist_ok=True)
        state = {'train_step': self.train_step_index, 'checkpoints': self.past_checkpoints}
        if full:
            state['optimizers'] = [optimizer.state_dict() for optimizer in self.optimizers.values()]
            state['scaler'] = {}
        state['model'] = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        file_path = f"step_{self.train_step_index:06d}.pth"
        if len(self.past_checkpoints) == 0 or file_path != self.past_checkpoints[-1]:
            self.past_checkpoints.append(file_path)
        if len(self.past_checkpoints) > max_keep_ckpts:
            old = os.path.join(ckpt_dir, self.past_checkpoints.pop(0))
            if os.path.exists(old):
                os.unlink(old)
        torch.save(state, os.path.join(ckpt_dir, file_path))
        return os.path.join(ckpt_dir, file_path)

    def load_checkpoint(self, checkpoint, model_only: bool = False, resume: bool = True):
        """trainer.py:188-236.  NB the parameters live in the optimizers' flat buffer: a checkpoint with a DIFFERENT Gaussian count must
        be loaded into the model before `get_optimizer` is called (the reference rebuilds its optimizers the same way)."""
        d = torch.load(checkpoint, map_location=self.model.device, weights_only=False)
        if 'model' not in d:
            self.model.load_state_dict(d)
            return
        self.model.load_state_dict(d['model'], strict=False)
        self.past_checkpoints = d['checkpoints']
        if resume:
            self.train_step_index = d['train_step']
        if model_only:
            return
        if self.optimizers is not None and 'optimizers' in d:
            for optimizer, sd in zip(self.optimizers.values(), d['optimizers']):
                optimizer.load_state_dict(sd)

    def render(self, data, bg_mode=None):
        if self.cfg.prompt.scene != 'canonical' or self.cfg.render.always_animate:
            smpl_observed_inputs = data['smpl_inputs']
        else:
            smpl_observed_inputs = None
        use_densifier = self.model.training and self.cfg.render.use_densifier
        return self.model.forward(data=data, smpl_observed_inputs=smpl_observed_inputs, use_densifier=use_densifier, bg_mode=bg_mode)

    def get_spatial_scale(self, data):
        if self.cfg.render.spatial_scale is None:
            return data['radius'].mean().item() * data['tanfov'].mean().item()
        return self.cfg.render.spatial_scale

    def _select_text(self, data):
        """View-dependent prompt selection (trainer.py:944-955): -> (embedding [1,77,d] or None, prompt string)."""
        if getattr(self, "_text_override", None) is not None:
            # a captured step (step_graph.GraphedTrainStep): the embedding of the step's view sits in a static buffer the replay reads
            return self._text_override, self.cfg.guide.text
        if self.cfg.prompt.text_augmentation and 'viewed' in self.text_embeds_dict:
            view_index = self.view_prompt(azim=data['azimuth'], elev=data['elevation']).item()
            return self.text_embeds_dict['viewed'][view_index], self.view_prompt.texts[view_index]
        return self.text_embeds_dict.get('pos'), self.cfg.guide.text

    def train_forward_views(self, views, **forced):
        """The forward of `train_forward` for SEVERAL views in one guidance call (SURVEY section 7 "hard parts": multi-view batching per
        launch): every view is rendered by its own animate + rasterizer pass, the V images go through the VAE encoder as a batch of V and
        through ControlNet + UNet as a CFG batch of 2 V (small-M layers of the 8x8 .. 32x32 latent levels finally fill their MFMA tiles).
        Each view keeps its own prompt embedding, condition image, and -- from its `rng_seed` -- exactly the random draws its own
        single-view call would make (guidance.draw_view_randoms)."""
        images, texts, names, conds, draws, outs = [], [], [], [], [], []
        # The views' animate + rasterizer passes are independent chains of small, latency-bound launches: each runs on its OWN stream (and
        # so does its backward: autograd replays a node on its forward's stream), so that one view's tails and dependent launches overlap
        # with the other views' work -- "V views per launch" in effect, for the whole per-view chain and not only the rasterizer.  The
        # first batched step runs on one stream (caches of constant canonical-pose results are filled there).  DWG_VIEW_STREAMS=0: off.
        dev = getattr(self.model, "device", None)
        multi = (dev is not None and torch.device(dev).type == "cuda" and self._views_warm and os.environ.get("DWG_VIEW_STREAMS", "1") != "0")
        main = torch.cuda.current_stream(dev) if multi else None
        if multi:
            while len(self._view_streams) < len(views):
                self._view_streams.append(torch.cuda.Stream(device=dev))
            renderer = getattr(self.model, "renderer", None)
            if renderer is not None:
                renderer.per_stream_pair_states = True
            # The views' backward chains run on their forwards' side streams and all end in the SAME parameters' AccumulateGrad nodes (created
            # by whichever view touched the parameter first): autograd orders each accumulation behind its producer with an event wait -- the
            # synchronisation this design wants, one per (parameter, view), not an accident -- and warns about the stream mismatch once per
            # process.  The single-view step creates and consumes its nodes on one stream and never triggers it.
            warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if warn_off is not None and not getattr(self, "_accum_warn_off", False):
                warn_off(False)
                self._accum_warn_off = True
        from . import gridencoder as _ge
        seeded = [v.get('rng_seed') is not None for v in views]
        if any(seeded) and not all(seeded):
            raise ValueError("train_forward_views: `rng_seed` must be given for every view of a batched step or for none "
                             "(%d of %d views carry one)" % (sum(seeded), len(views)))
        with _ge.table_grad_inplace(not multi):     # concurrent backwards must not read-add-write the table's gradient slice in place
            self._render_views(views, multi, main, images, texts, names, conds, draws, outs, forced)
        if multi:
            for side in self._view_streams[:len(views)]:
                main.wait_stream(side)
        self._views_warm = True
        return self._guide_views(images, texts, names, conds, draws, outs, forced)

    def _views_on_one_chain(self, views):
        """The batched step rasterizes its V views on ONE launch chain, forward and backward (Scene.forward_views; round 6) when the views
        agree on image size and field of view and nothing needs the per-view rasterizer objects (densification statistics).
        DWG_VIEW_FRAMES=0: every view runs its own seven + three launches (the round-5 path)."""
        if os.environ.get("DWG_VIEW_FRAMES", "1") == "0" or not hasattr(self.model, "forward_views") or getattr(self.model, "avatars", None) is not None:
            return False
        if self.densifiers is not None or not all(torch.is_tensor(v.get('extrinsic')) and v['extrinsic'].is_cuda for v in views):
            return False
        cams = {(int(v['image_height']), int(v['image_width']), float(v['tanfov'][0]), float(v['tanfov_x'][0]) if 'tanfov_x' in v else None) for v in views}
        return len(cams) == 1

    def _render_views(self, views, multi, main, images, texts, names, conds, draws, outs, forced):
        chain = self._views_on_one_chain(views)
        if chain:
            animated = self.cfg.prompt.scene != 'canonical' or self.cfg.render.always_animate
            ros = self.model.forward_views(views, [v['smpl_inputs'] if animated else None for v in views],
                                           streams=self._view_streams[:len(views)] if multi else None)
        for i, data in enumerate(views):
            if chain:
                ro = ros[i]
                img = ro['image'].permute(0, 3, 1, 2)
            elif multi:
                side = self._view_streams[i]
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ro = self.render(data=data)
                    img = ro['image'].permute(0, 3, 1, 2).contiguous()
                img.record_stream(main)
            else:
                ro = self.render(data=data)
                img = ro['image'].permute(0, 3, 1, 2)
            outs.append(ro)
            images.append(img)
            emb, name = self._select_text(data)
            texts.append(emb); names.append(name)
            if self.use_controlnet:
                conds.append(data['cond_images'])
            seed = data.get('rng_seed')
            if seed is not None and 'noise' not in forced:
                if self._view_rng is None:
                    self._view_rng = torch.Generator(device=images[-1].device)
                self._view_rng.manual_seed(int(seed))
                draws.append(self.diffusion.draw_view_randoms(self._view_rng, self.train_step_index, self.max_step))

    def _guide_views(self, images, texts, names, conds, draws, outs, forced):
        sd_inputs = torch.cat(images, dim=0).contiguous()
        embeds = dict(self.text_embeds_dict)
        if texts[0] is not None:
            embeds['text'] = torch.cat(texts, dim=0)
        sd_kwargs = {'inputs': sd_inputs, 'text_embeds_dict': embeds, 'train_step': self.train_step_index, 'max_iteration': self.max_step,
                     'grad_viz': False, 'scaler': self.scaler}
        if self.use_controlnet:
            sd_kwargs['cond_inputs'] = torch.cat(conds, dim=0)
        if draws:
            sd_kwargs.update(posterior_noise=torch.cat([d[0] for d in draws]), timestep=torch.cat([d[1] for d in draws]),
                             noise=torch.cat([d[2] for d in draws]))
        sd_kwargs.update(forced)
        sd_outputs = self.diffusion(**sd_kwargs)
        total_loss = sd_outputs['diffusion_loss'] * self.cfg.guide.lambda_guidance
        for ro in outs:
            ro['regularizations'] = {}
        return total_loss, outs, sd_outputs, names

    def train_forward(self, data: Dict[str, Any], **forced):
        """`forced` (timestep=, noise=, posterior_noise=) pins the random draws for parity tests."""
        render_outputs = self.render(data=data)
        sd_inputs = render_outputs['image'].permute(0, 3, 1, 2).contiguous()
        emb, text = self._select_text(data)
        if emb is not None:
            self.text_embeds_dict['text'] = emb
        sd_kwargs = {'inputs': sd_inputs, 'text_embeds_dict': self.text_embeds_dict, 'train_step': self.train_step_index,
                     'max_iteration': self.max_step, 'grad_viz': False, 'scaler': self.scaler}
        if self.use_controlnet:
            sd_kwargs['cond_inputs'] = data['cond_images']
        sd_kwargs.update(forced)
        sd_outputs = self.diffusion(**sd_kwargs)
        # (lambda_guidance == 1 and no regularizer on this path: the two element-wise launches `loss * 1 + 0` are not made; the value is the same)
        lam = self.cfg.guide.lambda_guidance
        diffusion_loss = sd_outputs['diffusion_loss'] if lam == 1 else sd_outputs['diffusion_loss'] * lam
        render_outputs['regularizations'] = {}
        total_loss = diffusion_loss
        return total_loss, render_outputs, sd_outputs, text

    def _begin_step(self, spatial_scale):
        """zero_grad + update_learning_rate on every optimizer (trainer.py:861-870).  The reference does this between the forward and the
        backward of its one view; nothing in a forward reads a gradient, so doing it first is the same step -- and it is what lets a
        multi-view step accumulate several backwards into the one flat gradient buffer."""
        flat = hasattr(self.optimizers, "buffers") and hasattr(self.optimizers, "zero_grad")      # the package's own dict: one fill for all of them
        if flat:
            self.optimizers.zero_grad()
        for optimizer in self.optimizers.values():
            if not flat:
                optimizer.zero_grad()
            if hasattr(optimizer, 'update_learning_rate'):
                optimizer.update_learning_rate(iteration=self.train_step_index, spatial_scale=spatial_scale)

    def _forward_backward(self, data, **forced):
        loss, render_outputs, sd_outputs, text = self.train_forward(data, **forced)
        self._backward(loss)
        return loss, render_outputs, sd_outputs, text

    def _backward(self, loss):
        """loss.backward() -- the seed gradient of a one-element loss is the constant 1: kept per (device, shape), not refilled by a launch
        every step."""
        if loss.numel() != 1 or loss.dtype != torch.float32:
            return loss.backward()
        key = (loss.device, tuple(loss.shape))
        if self._unit is None or self._unit[0] != key:
            self._unit = (key, torch.ones(loss.shape, device=loss.device))
        loss.backward(gradient=self._unit[1])

    def _view(self, data, **forced):
        """Forward + backward of ONE view, accumulated into the flat gradient buffer.  Sync-free pair sizing: the rasterizer's backward
        has already waited for this frame's pair count; a frame whose pair workspace was too small contributed a ZERO gradient
        (rasterizer._RasterizeGaussians.backward) and is rendered again here (capacity has grown) BEFORE anything reaches the optimizers."""
        seed = data.get('rng_seed') if isinstance(data, dict) else None
        if seed is not None:
            # the view's own RNG stream (VAE posterior, timestep, noise): the same draws whichever rank renders it.  A private device
            # generator -- re-seeding the process-wide one costs the host a few hundred microseconds per step.
            if self._view_rng is None:
                dev = getattr(self.model, "device", None)
                self._view_rng = torch.Generator(device=dev if dev is not None else next(self.model.parameters()).device)
            forced = dict(forced, generator=self._view_rng)
            self._view_rng.manual_seed(int(seed))
        out = self._forward_backward(data, **forced)
        renderer = getattr(self.model, "renderer", None)
        while renderer is not None and renderer.consume_overflow():
            self.redone_frames += 1
            if seed is not None:
                self._view_rng.manual_seed(int(seed))
            out = self._forward_backward(data, **forced)
        return out

    def train_step(self, data, **forced):
        """One optimizer step.  `data`: the loader's dict for the reference's single-view step, or a LIST of such dicts -- the views THIS
        rank renders in a multi-view step (view v of V goes to rank v mod world; identical parameters everywhere).
        Every view's gradient is accumulated into the one flat buffer, then ONE all-reduce (RCCL over xGMI) when world > 1, then the
        fused Adam with 1 / V folded in (the mean over ALL views of the step, `set_views`)."""
        views = list(data) if isinstance(data, (list, tuple)) else [data]
        self._check_densifier(len(views))
        self._check_caches()
        self.train_step_index += 1
        self._begin_step(self.get_spatial_scale(views[0]))
        out = None
        if len(views) > 1 and getattr(self.diffusion, "views", 1) == len(views):
            # the guidance plans are built for this many views per call: ONE VAE / denoiser pass for all of them
            renderer = getattr(self.model, "renderer", None)
            while True:
                loss, outs, sd_outputs, names = self.train_forward_views(views, **forced)
                self._backward(loss)
                if renderer is None or not renderer.consume_overflow():
                    break
                self.redone_frames += 1         # a truncated frame contributed zeros, the others did not: start the step's gradient over
                self._begin_step(self.get_spatial_scale(views[0]))
            out = (loss, outs, sd_outputs, names)
        else:
            for view in views:
                out = self._view(view, **forced)
        if self.densifiers is not None:
            # trainer.py:879-886: between backward and the optimizer steps (single-view, single-rank steps only: checked before the forward)
            self.model.densify(densifiers=self.densifiers, render_outputs=out[1], spatial_scale=self.get_spatial_scale(views[0]),
                               train_step=self.train_step_index)
        if self.world > 1:
            self._reduce_and_step()
        else:
            for optimizer in self.optimizers.values():
                optimizer.step()
        if self.world > 1 and self.replica_check_every > 0 and self.train_step_index % self.replica_check_every == 0:
            self.check_replicas()
        return out

    def _reduce_and_step(self):
        """The exchange step of a multi-rank SDS step (nothing in the reference to mirror: SURVEY 2.2 finds no torch.distributed).
        The flat gradient buffer is reduced SLICE BY SLICE, one asynchronous all-reduce per named optimizer, smallest slice first -- the grid
        table's 50 MB go last -- and every optimizer steps as soon as ITS slice has arrived, so the fused Adam launches of the small groups
        run under the table's reduce instead of behind it.  The sums are the ones a single all-reduce of the whole buffer gives (slices are
        disjoint ranges of one allocation).  Participation -- which groups took part in this step's backward, optim.FlatBuffers -- is agreed
        on ACROSS the ranks first (MAX of a per-group flag): a group some rank touched steps everywhere, a group no rank touched keeps its
        moments and step count everywhere; a rank deciding alone would let the replicas diverge.  The reduce's duration (events on the
        current stream around issue .. last wait) is kept in `allreduce_ms`."""
        opts, buf = self.optimizers, self.optimizers.buffers
        order = sorted(opts.values(), key=lambda o: o.end - o.start)
        cuda = buf.grad.is_cuda
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        groups = [pg for o in order for pg in o.param_groups]
        flags = torch.tensor([1 if (not buf.tracking or not pg.get('params') or any(getattr(q, "_dwg_touched", False) for q in pg['params'])) else 0
                              for pg in groups], dtype=torch.int32, device=buf.grad.device)
        fwork = self.dist.all_reduce(flags, op=self.dist.ReduceOp.MAX, async_op=True)
        works = [self.dist.all_reduce(buf.grad[o.start:o.end], async_op=True) for o in order]
        fwork.wait()
        took_part = [bool(f) for f in flags.tolist()]
        k = 0
        for o, work in zip(order, works):
            work.wait()
            n = len(o.param_groups)
            o.step(participation=took_part[k:k + n])
            k += n
        if cuda:
            e1.record()
            self._allreduce_events.append((e0, e1))
            if len(self._allreduce_events) > 64:
                self._drain_allreduce_events()

    def _drain_allreduce_events(self):
        for e0, e1 in self._allreduce_events:
            e1.synchronize()
            self.allreduce_ms_total += e0.elapsed_time(e1); self.allreduce_steps += 1
        self._allreduce_events = []

    @property
    def allreduce_ms(self):
        """Mean duration of the exchange step (issue of the first slice's reduce .. last optimizer launched), ms per step; None: no step yet."""
        self._drain_allreduce_events()
        return self.allreduce_ms_total / self.allreduce_steps if self.allreduce_steps else None

    def _check_densifier(self, n_views):
        """The densifier is the reference's single-view, single-GPU feature (trainer.py:879-886): its statistics are per rendered frame, and
        replicas of a multi-GPU job would have to densify identically.  Refused BEFORE a step's forward / backward run."""
        if self.densifiers is not None and (n_views != 1 or self.world != 1 or self.total_views != 1):
            raise NotImplementedError("densification inside a multi-view / multi-GPU step (views this rank: %d, ranks: %d, views per step: %d)"
                                      % (n_views, self.world, self.total_views))

    def _check_caches(self):
        """The constant canonical-pose caches of the avatar are (re)filled by the first render after `invalidate_caches()` (checkpoint load,
        densification): that render must run on ONE stream -- the per-view side streams of a batched step have no dependency on each other."""
        gen = getattr(getattr(self.model, "avatar", None), "cache_generation", None)
        if gen != getattr(self, "_cache_generation", None):
            self._cache_generation = gen
            self._views_warm = False

    # -- replica consistency (world > 1) ------------------------------------------------------------------------------------------------
    def sync_replicas(self, src: int = 0):
        """Every rank takes rank `src`'s state: the flat parameter buffer and Adam moments, the optimizers' scalars (step counts per group,
        iteration, learning rates: what the bias corrections and the schedule are computed from), the trainer's step index and the model's
        buffers that live outside the flat buffer -- at construction, and after a checkpoint load on one rank."""
        buf = getattr(self.optimizers, "buffers", None)
        if self.world <= 1 or buf is None or self.dist is None:
            return
        for t in (buf.flat, buf.m, buf.v):
            self.dist.broadcast(t, src=src)
        scal = [float(self.train_step_index)]
        for o in self.optimizers.values():
            scal += [float(o.t), float(o.current_iteration)]
            for pg in o.param_groups:
                scal += [float(pg.get("t", 0)), float(pg["lr"])]
        st = torch.tensor(scal, dtype=torch.float64, device=buf.flat.device)
        self.dist.broadcast(st, src=src)
        vals = st.tolist()
        self.train_step_index = int(vals[0]); k = 1
        for o in self.optimizers.values():
            o.t, o.current_iteration = int(vals[k]), int(vals[k + 1]); k += 2
            for pg in o.param_groups:
                pg["t"], pg["lr"] = int(vals[k]), vals[k + 1]; k += 2
        if isinstance(self.model, torch.nn.Module):
            for b in self.model.buffers():
                if b.device != buf.flat.device or b.numel() == 0:
                    continue
                if not b.is_contiguous():
                    raise RuntimeError("sync_replicas: a non-contiguous model buffer of shape %s cannot be broadcast in place" % (tuple(b.shape),))
                self.dist.broadcast(b.view(torch.uint8) if b.dtype == torch.bool else b, src=src)
        # the broadcasts wrote parameters through `.data` views: no tensor version moved, so everything keyed on the parameters' state must be
        # told -- the optimizers' parameter epoch (inference caches of the MLP heads / the frozen avatar) and the avatar's canonical caches
        from . import optim as _optim
        _optim.PARAM_EPOCH[0] += 1
        inv = getattr(self.model, "invalidate_caches", None) or getattr(getattr(self.model, "avatar", None), "invalidate_caches", None)
        if inv is not None:
            inv()

    def replica_checksum(self) -> torch.Tensor:
        """64-bit checksum of the flat parameter buffer: the sum of its fp32 words read as int32, in int64 (exact, order-independent)."""
        return self.optimizers.buffers.flat.view(torch.int32).sum(dtype=torch.int64)

    def check_replicas(self):
        """all-reduce(MAX) of (c, -c): max and min of the ranks' checksums in one collective; raises RuntimeError on EVERY rank on drift."""
        c = self.replica_checksum().reshape(1)
        both = torch.cat([c, -c])
        self.dist.all_reduce(both, op=self.dist.ReduceOp.MAX)
        hi, lo = int(both[0]), -int(both[1])
        if hi != lo:
            raise RuntimeError("replica drift at step %d: parameter checksums differ across the %d ranks (max %d, min %d; this rank %d) -- "
                               "the replicas of a multi-GPU SDS job must apply bit-identical updates" % (self.train_step_index, self.world, hi, lo, int(c)))

    def set_views(self, total_views: int):
        """Views per step over ALL ranks (default: one per rank): the optimizers see the MEAN gradient over them."""
        self.total_views = int(total_views)
        self._check_densifier(1)
        if hasattr(self.optimizers, "set_grad_scale"):
            self.optimizers.set_grad_scale(1.0 / self.total_views)
