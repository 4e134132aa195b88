"""One SDS render-and-distill step, end to end on the HIP kernels -- the loop body of Trainer.train()
(/root/reference/core/trainer.py:840-896: train_forward :933-1017 -> backward :876 -> optimizer steps :888-890) for the
default 3DGS stage, on the synthetic inputs of SURVEY.md section 8(d) config c3:

  animate (LBS + grid encoder + MLPs + mesh-bound Gaussians)  ->  rasterize 512^2  ->  VAE encode (in autograd)
  -> ControlNet + UNet CFG pass (batch 2, guidance 50)  ->  SpecifyGradient backward through VAE / rasterizer / avatar
  -> [RCCL all-reduce of the flat gradient buffer when world > 1]  ->  fused Adam on the flat parameter buffer.

Multi-GPU (config c4): rank r renders view r (azimuth 45 deg * r) of the SAME avatar (identical seeds for parameters,
distinct seeds for pose/noise), one flat fp32 all-reduce per step, identical Adam updates on every rank.
"""
import ctypes
import math

import torch

from . import _lib, avatar as av, camera, guidance as gd, renderer as rd, rasterizer, synth


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


class FlatAdam:
    """All trainable parameters live in ONE flat fp32 buffer (16-byte aligned slices, grouped by learning rate), gradients
    in a second one: a single all-reduce and one fused Adam launch per group (include/dwg_elementwise.h)."""

    def __init__(self, groups, device, betas=(0.9, 0.999), eps=1e-15):
        self.groups = []
        total = 0
        for g in groups:
            start = total
            for p in g["params"]:
                total += (p.numel() + 3) // 4 * 4
            self.groups.append(dict(lr=g["lr"], betas=g.get("betas", betas), start=start, end=total, name=g.get("name"),
                                    schedule=g.get("schedule"), base_lr=g.get("base_lr", g["lr"])))
        self.flat = torch.zeros(total, device=device)
        self.grad = torch.zeros(total, device=device)
        self.m = torch.zeros(total, device=device)
        self.v = torch.zeros(total, device=device)
        off = 0
        for g in groups:
            for p in g["params"]:
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p.data)
                p.grad = self.grad[off:off + n].view_as(p.data)
                off += (n + 3) // 4 * 4
        self.eps = eps
        self.t = 0

    def zero_grad(self):
        self.grad.zero_()

    def update_learning_rate(self, spatial_scale, iteration=None):
        """GaussianOptimizer.update_learning_rate (gaussian_optimizer.py:130-141): the 'positions' group follows its exponential
        schedule times spatial_scale, the 'scales' group is its base rate times spatial_scale; every other group is constant."""
        it = self.t if iteration is None else iteration
        lr = 0.0
        for g in self.groups:
            if g["name"] == "positions" and g["schedule"] is not None:
                lr = g["schedule"](it)
                g["lr"] = lr * spatial_scale
            elif g["name"] == "scales":
                lr = g["base_lr"]
                g["lr"] = lr * spatial_scale
        return lr

    def step(self, grad_scale=1.0):
        self.t += 1
        L = _lib.lib()
        st = ctypes.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
        for g in self.groups:
            n = g["end"] - g["start"]
            o = g["start"] * 4
            _lib.check(L.dwg_adam_step(n, ctypes.c_void_p(self.flat.data_ptr() + o), ctypes.c_void_p(self.grad.data_ptr() + o),
                                       ctypes.c_void_p(self.m.data_ptr() + o), ctypes.c_void_p(self.v.data_ptr() + o), g["lr"],
                                       g["betas"][0], g["betas"][1], self.eps, self.t, grad_scale, st), "dwg_adam_step")


class SDSStep:
    def __init__(self, n_gaussians=100000, res=512, device="cuda", rank=0, world=1, guidance=True, dist=None, seed=0):
        self.device, self.rank, self.world, self.dist, self.res = torch.device(device), rank, world, dist, res
        self.G = n_gaussians
        M = (n_gaussians // 10) // 6 * 6          # mesh-bound Gaussians (hands/face), 6 per triangle
        N = n_gaussians - M
        self.N, self.M = N, M
        body = synth.synthetic_body(seed=seed)
        body = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in body.items()}
        glbs = av.GeneralLinearBlendSkinning(body).to(self.device)
        g = synth.random_gaussians(N, seed=seed)
        gen = torch.Generator().manual_seed(seed + 11)
        logits = torch.full((N, 55), -1e9)
        logits.scatter_(1, torch.randint(0, 55, (N, 4), generator=gen), torch.randn(N, 4, generator=gen))
        lbs_w = torch.softmax(logits, dim=1)
        cnl = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
                   right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
        cnl = {k: v.to(self.device) for k, v in cnl.items()}
        mesh = None
        if M > 0:
            # "hands": two local vertex clusters, triangles = a vertex and its two nearest neighbours (cm-sized faces, as on the
            # SMPL-X hand/face meshes the reference binds to)
            Vp, Fp = 1200, M // 6
            vt = body["v_template"].cpu()
            centres = torch.tensor([[0.35, 0.2, 0.0], [-0.35, 0.2, 0.0]])
            d2 = ((vt[:, None, :] - centres[None]) ** 2).sum(-1)                      # [V, 2]
            vi = torch.cat([torch.topk(d2[:, 0], Vp // 2, largest=False).indices, torch.topk(d2[:, 1], Vp // 2, largest=False).indices])
            sub = vt[vi]
            nn = torch.topk(torch.cdist(sub, sub), 3, largest=False).indices          # self + 2 nearest
            a = torch.randint(0, Vp, (Fp,), generator=gen)
            tri = nn[a]
            mesh = {"hands": av.MeshBindingGaussianModel(sub, tri, vi)}
        self.avatar = av.DreamWaltzG(glbs, g["positions"], g["scales"], g["quaternions"], lbs_w, cnl, mesh).to(self.device)
        self.renderer = rd.GaussianRenderer(bg_color=(0.5, 0.5, 0.5))
        cam = camera.make_camera(radius=2.0, azimuth=45.0 * rank, elevation=80.0, fovy=55.0, height=res, width=res, device=self.device)
        cam["tanfov_host"] = float(cam["tanfov"][0])
        self.cam = cam
        self.guidance = None
        if guidance:
            self.guidance = gd.ControlNetScoreDistillation(self.device, image_hw=512, seed=seed)
            tg = torch.Generator().manual_seed(seed + 5)
            self.text = {"neg": torch.randn(1, 77, 768, generator=tg).to(self.device), "text": torch.randn(1, 77, 768, generator=tg).to(self.device)}
            self.cond = (torch.randint(0, 256, (1, 3, 512, 512), generator=tg).float() / 255.0).to(self.device)
        else:
            self.wimg = torch.randn(1, res, res, 3, generator=torch.Generator().manual_seed(seed + 6)).to(self.device)
        a = self.avatar
        spatial = 2.0 * float(cam["tanfov"][0])      # spatial_scale = radius * tanfov (trainer.py:711-716)
        iters = 10000                                # cfg.optim.iters; position_lr_max_steps = 2 * iters (avatar.py:1594-1601)
        groups = [
            dict(params=[a._positions], lr=1.6e-4 * spatial, name="positions",
                 schedule=get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=2 * iters)),
            dict(params=[a._scales], lr=2.5e-3 * spatial, name="scales", base_lr=2.5e-3),
            dict(params=[a._quaternions], lr=1e-3, name="quaternions"),
            dict(params=[a.nerf_encoder.embeddings], lr=1e-2, betas=(0.9, 0.99)),
            dict(params=list(a.nerf_opacity_and_color_net.parameters()) + list(a.nerf_scale_and_quaternion_net.parameters()),
                 lr=1e-3, betas=(0.9, 0.99)),
        ]
        for gm in a.mesh_binding_gaussians.values():
            groups.append(dict(params=[gm._bary_coords, gm._scales], lr=1e-3))
        self.opt = FlatAdam(groups, self.device)
        self.spatial_scale = spatial
        self.step_idx = 0
        self.num_pairs = 0
        rasterizer.ASYNC[0] = True       # training loop: no host sync for the pair count (checked one frame late)
        torch.manual_seed(1234 + rank)

    def capture_graphs(self):
        if self.guidance is not None:
            self.guidance.capture_graphs()

    def set_use_graphs(self, on):
        if self.guidance is not None:
            self.guidance.set_use_graphs(on)

    def run(self):
        self.opt.zero_grad()
        self.opt.update_learning_rate(self.spatial_scale, self.step_idx)      # trainer.py:861-866 (host-side floats)
        pose = synth.random_smpl_inputs(seed=1000 * self.rank + self.step_idx, device=self.device)
        gaussians = self.avatar.animate(pose)
        out = self.renderer.render(self.cam, gaussians)
        self.num_pairs = rasterizer.LAST_NUM_PAIRS[0]
        if self.guidance is not None:
            image = out["image"].permute(0, 3, 1, 2)
            res = self.guidance(image, self.text, cond_inputs=self.cond)
            loss = res["diffusion_loss"] * 1.0
        else:
            loss = (out["image"] * self.wimg).sum()
        loss.backward()
        if self.world > 1:
            self.dist.all_reduce(self.opt.grad)      # RCCL over xGMI: one flat fp32 buffer
        self.opt.step(grad_scale=1.0 / self.world)
        self.step_idx += 1

    # -- reporting -----------------------------------------------------------------------------------------------
    def describe(self):
        wl = ("full SDS step: animate(LBS+grid-encoder+MLPs, %d unconstrained + %d mesh-bound Gaussians) -> raster %dx%d fwd+bwd "
              "-> VAE-encode fwd+dgrad -> ControlNet+UNet SD-1.5 CFG batch 2 @64x64 latents -> Adam" % (self.N, self.M, self.res, self.res)
              ) if self.guidance is not None else (
            "sub-path only (NOT the headline workload): animate + raster %dx%d fwd+bwd + Adam, no diffusion" % (self.res, self.res))
        return {"dtype": "bf16" if self.guidance is not None else "f32",
                "config": {"workload": wl, "gaussians": self.G, "resolution": self.res, "views_per_step_per_gpu": 1,
                           "weights": "seeded random init of the SD-1.5 / ControlNet / VAE architecture",
                           "precision": "denoiser+VAE bf16 storage / fp32 accumulate; LBS, encoder, MLPs, rasterizer fp32",
                           "parallelism": "dp%d (one view per GPU, flat-gradient all-reduce)" % self.world}}

    def flops_by_kernel(self):
        tot = {}
        if self.guidance is None:
            return tot
        for plan in (self.guidance.denoiser.plan, self.guidance.vae.fwd, self.guidance.vae.bwd):
            for k, v in plan.flops.items():
                tot[k] = tot.get(k, 0.0) + v
        return tot

    def roofline(self, prof, hbm_peak_gbs, bf16_peak_tflops, symbols=None):
        """Roofline entry for the dominant kernel = the kernel SYMBOL (as rocprofv3 --kernel-trace names it) with the
        largest total time in the profiled region.  `symbols` = _lib.prof_symbols(): per-symbol launches, summed HIP-event
        duration and summed algorithmic flops (2*M*N*K per launch), so achieved = flops per launch / average launch
        duration, directly comparable with the AverageNs column of profiles/*_kernel_stats.csv."""
        if not prof:
            return None
        G, K, P = self.G, self.num_pairs, self.res * self.res
        out = {}
        if symbols:
            name, (count, total_ms, work) = max(symbols.items(), key=lambda kv: kv[1][1])
            avg_ms = total_ms / max(1, count)
            if work > 0:
                ach = work / count / (avg_ms * 1e-3) / 1e12
                out = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": bf16_peak_tflops, "unit": "TFLOP/s",
                       "frac": ach / bf16_peak_tflops, "traffic": None, "avg_launch_ms": avg_ms, "launches": count,
                       "flops_per_launch": work / count}
            else:
                out = {"kernel": name, "bound": "hbm", "achieved": None, "peak": hbm_peak_gbs, "unit": "GB/s", "frac": None,
                       "traffic": None, "avg_launch_ms": avg_ms, "launches": count}
            # every MFMA kernel symbol, same formula (the conv / linear / attention products of the denoiser and the VAE)
            out["mfma_kernels"] = {k: {"launches": c, "avg_launch_ms": ms / c, "tflops": w / (ms * 1e-3) / 1e12,
                                       "frac": w / (ms * 1e-3) / 1e12 / bf16_peak_tflops}
                                   for k, (c, ms, w) in sorted(symbols.items(), key=lambda kv: -kv[1][1]) if w > 0 and ms > 0}
            tw = sum(w for (_, _, w) in symbols.values()); tms = sum(ms for (_, ms, w) in symbols.values() if w > 0)
            if tms > 0:
                out["mfma_all"] = {"tflops": tw / (tms * 1e-3) / 1e12, "frac": tw / (tms * 1e-3) / 1e12 / bf16_peak_tflops,
                                   "flops_per_step": tw / self._steps_timed(prof)}
        # the rasterizer's own HBM roofline is part of the headline metric: always report it next to the dominant kernel
        rf = sum(prof[k][1] for k in prof if k.startswith("raster_") and not k.endswith("_bwd"))  # noqa
        rb = sum(prof[k][1] for k in prof if k.startswith("raster_") and k.endswith("_bwd"))
        n = max(1, prof.get("raster_render_fwd", (1, 0))[0])
        if rf > 0:
            out["raster_forward"] = {"bytes": 56 * G + 44 * K + 20 * P, "pairs": K, "ms": rf / n,
                                     "achieved_GBps": (56 * G + 44 * K + 20 * P) / (rf / n * 1e-3) / 1e9,
                                     "frac_of_hbm_peak": (56 * G + 44 * K + 20 * P) / (rf / n * 1e-3) / 1e9 / hbm_peak_gbs}
        if rb > 0:
            out["raster_backward"] = {"bytes": 80 * K + 20 * P + 152 * G, "ms": rb / n,
                                      "achieved_GBps": (80 * K + 20 * P + 152 * G) / (rb / n * 1e-3) / 1e9,
                                      "frac_of_hbm_peak": (80 * K + 20 * P + 152 * G) / (rb / n * 1e-3) / 1e9 / hbm_peak_gbs}
        return out

    def _steps_timed(self, prof):
        return max(1, prof.get("raster_render_fwd", (1, 0))[0])
