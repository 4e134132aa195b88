"""One SDS render-and-distill step on synthetic inputs: the harness bench.py and the
end-to-end tests drive.  Everything it calls is the product's reference-shaped API:

  DreamWaltzG.get_optimizer(cfg)  ->  Scene(cfg, avatar)  ->  SDSTrainer.train_step(data)
      = Scene.forward (animate: LBS + grid encoder + MLPs + mesh-bound Gaussians -> rasterize)  ->  ControlNetScoreDistillation
        (VAE encode in autograd -> ControlNet + UNet CFG batch 2 -> SpecifyGradient)  ->  backward  ->  [RCCL all-reduce of the flat
        gradient buffer when world > 1]  ->  fused Adam on the flat parameter buffer.

Multi-view (config c4): a step renders V views of the SAME avatar -- view v has its own camera (azimuth 45 deg * v), its
own pose and its own RNG stream (VAE posterior / timestep / noise), all functions of (v, step) only -- and view v is rendered by rank
v mod world.  Every rank accumulates its views into the flat gradient buffer, ONE flat fp32 all-reduce per step, the mean over the V
views folded into the fused Adam, identical updates on every rank.  The default (V = world) is one view per rank.
"""
import torch

from . import avatar as av, camera, configs, guidance as gd, scene as sc, synth, trainer as tr
from .condition import build_ray_casting_scene as cd_build
from .optim import get_expon_lr_func  # noqa: F401  (re-exported: tests pin it against the reference's function)


def build_synthetic_avatar(n_gaussians, device, seed=0, learn_hand_betas=False):
    """c3's avatar: 90 % free Gaussians in a body-sized box + 10 % mesh-bound on two vertex clusters ("hands")."""
    M = (n_gaussians // 10) // 6 * 6          # mesh-bound Gaussians, 6 per triangle
    N = n_gaussians - M
    body = synth.synthetic_body(seed=seed)
    body = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in body.items()}
    glbs = av.GeneralLinearBlendSkinning(body).to(device)
    g = synth.random_gaussians(N, seed=seed)
    gen = torch.Generator().manual_seed(seed + 11)
    logits = torch.full((N, 55), -1e9)
    logits.scatter_(1, torch.randint(0, 55, (N, 4), generator=gen), torch.randn(N, 4, generator=gen))
    lbs_w = torch.softmax(logits, dim=1)
    cnl = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
               right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
    cnl = {k: v.to(device) for k, v in cnl.items()}
    mesh = None
    if M > 0:
        # "hands": two local vertex clusters, triangles = a vertex and its two nearest neighbours (cm-sized faces, as on the
        # SMPL-X hand/face meshes the reference binds to)
        Vp, Fp = min(1200, body["v_template"].shape[0]), M // 6
        vt = body["v_template"].cpu()
        centres = torch.tensor([[0.35, 0.2, 0.0], [-0.35, 0.2, 0.0]])
        d2 = ((vt[:, None, :] - centres[None]) ** 2).sum(-1)                      # [V, 2]
        vi = torch.cat([torch.topk(d2[:, 0], Vp // 2, largest=False).indices, torch.topk(d2[:, 1], Vp // 2, largest=False).indices])
        sub = vt[vi]
        nn = torch.topk(torch.cdist(sub, sub), 3, largest=False).indices          # self + 2 nearest
        a = torch.randint(0, Vp, (Fp,), generator=gen)
        mesh = {"hands": av.MeshBindingGaussianModel(sub, nn[a], vi)}
    with torch.random.fork_rng(devices=[]):      # encoder / MLP initialisation: a function of `seed` only (every rank, every instance)
        torch.manual_seed(seed + 21)
        avatar = av.DreamWaltzG(glbs, g["positions"], g["scales"], g["quaternions"], lbs_w, cnl, mesh, learn_hand_betas=learn_hand_betas)
    return avatar.to(device), N, M


class SDSStep:
    def __init__(self, n_gaussians=100000, res=512, device="cuda", rank=0, world=1, guidance=True, dist=None, seed=0, cfg=None,
                 avatar=None, guidance_obj=None, async_pair_count=True, iters=10000, gpu_condition=True, views=None, dtype="f32x",
                 batch_views=False):
        self.device, self.rank, self.world, self.dist, self.res = torch.device(device), rank, world, dist, res
        self.views = int(views) if views is not None else world           # views per step over ALL ranks
        if self.views < world:
            raise ValueError("a multi-view step needs at least one view per rank (views=%d, world=%d)" % (self.views, world))
        self.my_views = list(range(rank, self.views, world))              # view v -> rank v mod world
        self.cfg = cfg if cfg is not None else configs.TrainConfig()
        self.cfg.device = str(self.device)
        self.cfg.render.bg_color = (0.5, 0.5, 0.5)           # the GS recipes (train_w_expr.sh:68,81,94)
        self.cfg.optim.iters = iters
        if avatar is None:
            avatar, self.N, self.M = build_synthetic_avatar(n_gaussians, self.device, seed, self.cfg.render.learn_hand_betas)
        else:
            self.N, self.M = avatar._n_points, avatar._n_points_on_mesh
        self.G = self.N + self.M
        self.avatar = avatar
        self.optimizers = avatar.get_optimizer(self.cfg)                 # dict of views on ONE flat buffer
        self.scene = sc.Scene(self.cfg, avatar, async_pair_count=async_pair_count).to(self.device)
        self.scene.train()
        self.view_data = {v: dict(camera.make_camera(radius=2.0, azimuth=45.0 * v, elevation=80.0, fovy=55.0, height=res, width=res,
                                                     device=self.device)) for v in self.my_views}
        self.data = self.view_data[self.my_views[0]]
        self.guidance = guidance_obj
        tg = torch.Generator().manual_seed(seed + 5)
        if guidance and self.guidance is None:
            # batch_views: ONE VAE / denoiser pass per step for all the views of this rank (plans built for that batch)
            self.guidance = gd.ControlNetScoreDistillation(self.device, image_hw=512, seed=seed, cfg=self.cfg.guide, dtype=dtype,
                                                           views=len(self.my_views) if batch_views else 1)
        if self.guidance is not None:
            cd = self.guidance.unet_cfg.cross_dim
            hw = self.guidance.image_hw
            self.text = {"neg": torch.randn(1, 77, cd, generator=tg).to(self.device), "pos": torch.randn(1, 77, cd, generator=tg).to(self.device),
                         "viewed": [torch.randn(1, 77, cd, generator=tg).to(self.device) for _ in range(14)]}
            fixed_cond = (torch.randint(0, 256, (1, 3, hw, hw), generator=tg).float() / 255.0).to(self.device)
            for d in self.view_data.values():
                d["cond_images"] = fixed_cond
            self.condition = self._build_condition(hw, seed) if gpu_condition else None
            diffusion = self.guidance
        else:
            self.text = {}
            wimg = torch.randn(1, res, res, 3, generator=torch.Generator().manual_seed(seed + 6)).to(self.device)
            diffusion = _ImageLoss(wimg)
        densifiers = None
        if self.cfg.render.use_densifier:                                # trainer.py:600-603 (off in every shipped recipe)
            densifiers = {'avatar': avatar.get_densifier(cfg=self.cfg, optimizer=self.optimizers)}
        self.trainer = tr.SDSTrainer(self.cfg, self.scene, diffusion, self.optimizers, self.text, use_controlnet=self.guidance is not None,
                                     dist=dist, world=world, max_step=iters, densifiers=densifiers)
        self.trainer.set_views(self.views)
        self.step_idx = 0
        # optional: step index -> the loader's camera dict of that step (camera.make_camera(..., device="cpu")) -- the reference samples a new
        # camera every step (data/camera/__init__.py:124-165); None: the fixed camera of the benchmark configurations
        self.camera_fn = None

    # kept names (bench.py / tools)
    @property
    def opt(self):
        return self.optimizers

    @property
    def renderer(self):
        return self.scene.renderer

    def capture_graphs(self):
        if self.guidance is not None:
            self.guidance.capture_graphs()

    def set_use_graphs(self, on):
        if self.guidance is not None:
            self.guidance.set_use_graphs(on)

    # -- the data layer's per-step condition image (smpl_condition.py:271-320), on the GPU -----------------------------------
    def _build_condition(self, hw, seed):
        """The reference's loader runs a full SMPL-X forward and draws the OpenPose skeleton of its 128 keypoints with occlusion
        culling against the 20 908-triangle body every step.  Synthetic stand-ins for the licensed model's topology: triangles
        between neighbours of a spatial ordering of the template vertices, keypoints = the 55 posed joints + 73 posed vertices."""
        from . import condition as cd
        lbs = self.avatar.lbs_model
        V = int(lbs.v_template.shape[0])
        g = torch.Generator().manual_seed(seed + 31)
        order = torch.argsort(lbs.v_template[:, 1].cpu() * 7.0 + lbs.v_template[:, 0].cpu())
        k = torch.randint(0, V - 2, (20908,), generator=g)
        tri = torch.stack([order[k], order[k + 1], order[k + 2]], dim=1).to(torch.int32).to(self.device)
        pick = torch.randint(0, V, (cd.N_KEYPOINTS - 55,), generator=g).to(self.device)
        f = hw / (2.0 * float(self.data["tanfov"][0]))
        intr = torch.tensor([[f, 0.0, hw / 2.0], [0.0, f, hw / 2.0], [0.0, 0.0, 1.0]], device=self.device)
        return dict(gen=cd.SMPL2Condition(self.cfg.prompt), triangles=tri, pick=pick, intrinsics=intr, hw=hw,
                    all_vertices=torch.arange(V, device=self.device, dtype=torch.int32))

    def condition_image(self, smpl_inputs, data=None):
        """[1,3,H,W] in [0,1]: posed body (one skeleton pass + all 10 475 vertices) -> keypoints -> culling -> OpenPose drawing."""
        c = self.condition
        data = self.data if data is None else data
        lbs = self.avatar.lbs_model
        with torch.no_grad():
            _, _, tr = lbs(**smpl_inputs)
            verts = lbs.transform_vertices(tr, c["all_vertices"], lbs.v_template)
            A = tr.A
            J = lbs._joints(tr)
            joints = (A[:, :3, :3] * J[:, None, :]).sum(-1) + A[:, :3, 3]         # A carries the global translation (no library GEMM for 55 3x3 products)
            keypoints = torch.cat([joints, verts[c["pick"]]], dim=0)
            scene = cd_build(verts, c["triangles"])
            intr = data["cond_intrinsics"] if data.get("cond_intrinsics") is not None else c["intrinsics"]      # a moving camera brings its own
            return c["gen"].export_pose_chw(keypoints, scene, extrinsic=data["extrinsic"][0], intrinsics=intr, width=c["hw"], height=c["hw"])

    def camera_tensors(self, cam: dict) -> dict:
        """What a step needs of the loader's camera dict `cam` (host tensors, camera.make_camera): the three matrices, the field of view as
        {tanfovx, tanfovy} for the device-resident camera of a captured step, the condition image's pinhole intrinsics, and the two host
        scalars of the learning-rate schedule's spatial scale."""
        tfy = float(cam["tanfov"][0]); tfx = float(cam["tanfov_x"][0]) if "tanfov_x" in cam else tfy
        out = {"extrinsic": cam["extrinsic"].float(), "projection": cam["projection"].float(), "c2w": cam["c2w"].float(),
               "tanfov_dev": torch.tensor([tfx, tfy]), "radius": cam["radius"], "tanfov": cam["tanfov"]}
        for k in ("azimuth", "elevation"):          # host scalars of the view-dependent prompt selection (trainer._select_text)
            if k in cam:
                out[k] = cam[k]
        if getattr(self, "condition", None) is not None:
            hw = self.condition["hw"]
            fx, fy = hw / (2.0 * tfx), hw / (2.0 * tfy)
            out["cond_intrinsics"] = torch.tensor([[fx, 0.0, hw / 2.0], [0.0, fy, hw / 2.0], [0.0, 0.0, 1.0]])
        return out

    def _apply_camera(self, d: dict, cam: dict):
        """The eager step's per-step camera: the loader's dict entries replaced in the view's data (host scalars stay on the host, as the
        reference reads them with .item(): gaussian_renderer.py:28, trainer.py:713)."""
        ct = self.camera_tensors(cam)
        for k in ("extrinsic", "projection", "c2w"):
            d[k] = ct[k].to(self.device)
        for k in ("tanfov", "radius", "azimuth", "elevation"):
            d[k] = cam[k]
        if "tanfov_x" in cam:                       # a non-square field of view: the renderer reads it next to tanfov (renderer.py)
            d["tanfov_x"] = cam["tanfov_x"]
        else:
            d.pop("tanfov_x", None)
        if "cond_intrinsics" in ct:
            d["cond_intrinsics"] = ct["cond_intrinsics"].to(self.device)

    def _upload_pose(self, cpu_inputs):
        """The per-step host input (the 165-float pose) goes up through pinned staging buffers with an asynchronous copy: a plain
        `.to(device)` from pageable memory blocks the host until everything queued on the stream has finished, i.e. it drains the
        pipeline every step and exposes the host's enqueue time of the next forward (measured: ~1 ms per step).  Four rotating slots;
        the host is never more than one step ahead (the rasterizer's pair-count event), so a slot is free when it comes round again."""
        if self.device.type != "cuda":
            return {k: v.to(self.device) for k, v in cpu_inputs.items()}
        nslot = 4 * len(self.my_views)
        slots = self.__dict__.setdefault("_pose_slots", [None] * nslot)
        i = self.__dict__["_pose_slot_next"] = (self.__dict__.get("_pose_slot_next", -1) + 1) % nslot
        if slots[i] is None:
            slots[i] = {k: torch.empty_like(v).pin_memory() for k, v in cpu_inputs.items()}
        out = {}
        for k, v in cpu_inputs.items():
            slots[i][k].copy_(v)
            out[k] = slots[i][k].to(self.device, non_blocking=True)
        return out

    def run(self, **forced):
        views = []
        for v in self.my_views:
            d = self.view_data[v]
            d["smpl_inputs"] = self._upload_pose(synth.random_smpl_inputs(seed=1000 * v + self.step_idx, device="cpu"))
            if self.camera_fn is not None:
                self._apply_camera(d, self.camera_fn(self.step_idx))
            if self.guidance is not None:
                d["rng_seed"] = (1234 + v) * 1000003 + self.step_idx      # the view's own device-RNG stream (Q12 draw order inside it)
            if getattr(self, "condition", None) is not None:
                d["cond_images"] = self.condition_image(d["smpl_inputs"], d)
            views.append(d)
        out = self.trainer.train_step(views if len(views) > 1 else views[0], **forced)
        self.step_idx += 1
        return out

    def graphed(self, warmup=4):
        """The whole step as ONE captured HIP graph (step_graph.GraphedTrainStep): single-view steps without guidance (config c2).  Returns
        an object whose `.step()` takes the place of `run()`: same pose sequence, same updates."""
        from . import step_graph
        if len(self.my_views) != 1 or self.world != 1:
            raise NotImplementedError("only the single-view, single-rank step is captured as a whole")
        d = self.view_data[self.my_views[0]]
        v = self.my_views[0]
        poses = [{k: t.to(self.device) for k, t in synth.random_smpl_inputs(seed=1000 * v + self.step_idx + i, device="cpu").items()}
                 for i in range(warmup + 1)]
        data = {k: t for k, t in d.items() if k != "smpl_inputs"}
        cond_fn = (lambda pose, data=None: self.condition_image(pose, d if data is None else data)) if getattr(self, "condition", None) is not None else None
        # the view's device-RNG stream of step k (0-based) is seeded as run() seeds it: the trainer's index is k + 1 when the draws are made
        seed_fn = (lambda idx: (1234 + v) * 1000003 + (idx - 1)) if self.guidance is not None else None
        if self.guidance is not None and cond_fn is None:
            data["cond_images"] = d["cond_images"]
        cam_kw, cams = {}, None
        if self.camera_fn is not None:
            # a camera per step, in device memory: the captured step follows it (step_graph.GraphedTrainStep.step(pose, camera))
            cams = [self.camera_tensors(self.camera_fn(self.step_idx + i)) for i in range(warmup + 1)]
            dev = lambda c: {k: (t.to(self.device) if k in step_graph.GraphedTrainStep.CAMERA_KEYS else t) for k, t in c.items()}   # noqa: E731
            cam_kw = dict(example_camera=dev(cams[0]), warmup_cameras=[dev(c) for c in cams[:warmup]], capture_camera=dev(cams[warmup]))
        g = step_graph.GraphedTrainStep(self.trainer, data, poses[0], warmup_poses=poses[:warmup], capture_pose=poses[warmup],
                                        condition_fn=cond_fn, seed_fn=seed_fn, **cam_kw)
        self.step_idx += warmup + 1          # the warm-up steps and the capture's eager step were real optimizer steps
        step = self

        class _Runner:
            graph = g

            def step(self_inner):
                cam = step.camera_tensors(step.camera_fn(step.step_idx)) if step.camera_fn is not None else None
                out = g.step(synth.random_smpl_inputs(seed=1000 * v + step.step_idx, device="cpu"), cam)
                step.step_idx += 1
                return out
        return _Runner()

    @property
    def num_pairs(self):
        """(pairs after exact culling, reference tile-pair count K of SURVEY 8d) of the last rendered frame."""
        r = self.scene.renderer.last_rasterizer
        return r.last_num_pairs if r is not None else (0, 0)

    # -- reporting -----------------------------------------------------------------------------------------------
    def describe(self):
        wl = ("full SDS step: " + ("OpenPose condition image of the posed body (skeleton pass over all vertices, 128 keypoints, culling "
                                   "against 20908 triangles, drawing) -> " if getattr(self, "condition", None) is not None else "") +
              "animate(LBS+grid-encoder+MLPs, %d unconstrained + %d mesh-bound Gaussians) -> raster %dx%d fwd+bwd "
              "-> VAE-encode fwd+dgrad -> ControlNet+UNet SD-1.5 CFG batch 2 @64x64 latents -> Adam" % (self.N, self.M, self.res, self.res)
              ) if self.guidance is not None else (
            "sub-path only (NOT the headline workload): animate + raster %dx%d fwd+bwd + Adam, no diffusion" % (self.res, self.res))
        gdt = self.guidance.dtype_name if self.guidance is not None else "f32"
        prec = {"bf16": "denoiser+VAE bf16 storage / fp32 accumulate (MFMA bf16)", "f32": "denoiser+VAE fp32 storage and arithmetic (exact-f32 MFMA)",
                "f16": "denoiser+VAE fp16 storage / fp32 accumulate (MFMA f16)",
                "f32x": "denoiser+VAE fp32-grade split precision: every value hi + 2^-11 lo fp16 halves (22 significand bits, 4 bytes), every product "
                        "three f16 MFMAs with fp32 accumulate (eps 3e-6 / SDS gradient 1e-5 vs the fp32 CPU oracle: the reference's fp32 results)"}[gdt]
        return {"dtype": gdt,
                "config": {"workload": wl, "gaussians": self.G, "resolution": self.res, "views_per_step": self.views,
                           "views_per_step_per_gpu": len(self.my_views),
                           "views_per_guidance_call": getattr(self.guidance, "views", 1) if self.guidance is not None else None,
                           "weights": "seeded random init of the SD-1.5 / ControlNet / VAE architecture",
                           "precision": prec + "; LBS, encoder, MLPs, rasterizer fp32",
                           "parallelism": "dp%d (view v on GPU v mod %d, flat-gradient all-reduce)" % (self.world, self.world)}}

    def flops_by_kernel(self):
        tot = {}
        if self.guidance is None:
            return tot
        for plan in self.guidance.plans():
            for k, v in plan.flops.items():
                tot[k] = tot.get(k, 0.0) + v
        return tot


class _ImageLoss:
    """Stand-in for the diffusion object in the no-guidance sub-path (config c2: 'no guidance, grad-check'): loss = sum(image * W)
    with a fixed random W, same call signature / result keys as the guidance object."""

    def __init__(self, wimg):
        self.wimg = wimg

    def __call__(self, inputs, text_embeds_dict=None, train_step=0, max_iteration=1, **kwargs):
        loss = (inputs.permute(0, 2, 3, 1) * self.wimg).sum().reshape(1)
        return {"diffusion_loss": loss, "timestep": torch.zeros(1, dtype=torch.long), "gradients": None, "latents": None,
                "sources": None, "targets": None}
