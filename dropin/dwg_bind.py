"""dwg_bind -- binds the reference's OWN main.py / Trainer to the HIP path at import time, without editing a reference file.

Enable:   PYTHONPATH=<repo>/dropin:<repo>  DWG_BIND=1  python main.py --stage gs ...
(`dropin/sitecustomize.py` calls `install()` when DWG_BIND=1; or call `dwg_bind.install()` yourself before importing `core.*`.)

What gets bound, all through attributes the reference resolves AT CALL TIME (`from core.system.avatar import build_gaussian_avatar` etc.
sit inside Trainer methods: /root/reference/core/trainer.py:446-453,529-530), so replacing the module attribute is enough:

  B1  diff_gaussian_rasterization           dropin/diff_gaussian_rasterization/  (found through PYTHONPATH; nothing to patch)
  B2  core.nerf.gridencoder backend         dropin/_gridencoder.py               (found through PYTHONPATH before the JIT build)
  B3  core.system.avatar.build_gaussian_avatar (avatar.py:1642-1714)  -> the reference builds ITS avatar (point cloud, nearest triangles,
                                             inverse LBS, LBS weights ...), then `DreamWaltzG.from_reference(ref)` adopts every Parameter
                                             and buffer by name; non-DreamWaltzG gs_types are returned untouched (reference path)
  B5  core.system.scene.build_scene (scene.py:224-245)                -> dreamwaltz_g_amd.scene.Scene around that avatar (same forward /
                                             state_dict / avatar.get_optimizer surface the Trainer uses: trainer.py:578-604,680-709,859-890)
  B4  core.guidance.controlnet.ControlNetScoreDistillation (controlnet.py:75-114) -> the reference constructs its object as always
                                             (diffusers pipeline, text encoder, schedulers); after __init__ the two hot methods of THAT object,
                                             `_predict` (controlnet.py:83-114) and `encode_images` (vae.py:34-40), are bound to the HIP plans
                                             built from the loaded modules' state_dict()s.  Everything else (get_text_embeds, __call__,
                                             calc_gradients, tp_scheduler, pipe, decode_latents, isinstance checks) is the reference's own.

Environment: DWG_BIND_DTYPE = f32x | f32 | f16 | bf16  storage type of the denoiser / VAE plans.  Unset: the precision the reference loaded its
                                                     pipeline in -- torch.float32 (its default, core/guidance/basic.py:233) -> f32x (fp32-grade
                                                     split precision on the 16-bit MFMA pipe), torch.float16 (`--guide.dtype fp16`,
                                                     basic.py:24-27) -> f16.  bf16 narrows the user's precision: only on request
             DWG_BIND_KEEP_MODULES = 1               keep the diffusers UNet / ControlNet on the GPU (default: moved to the CPU once their
                                                     weights live in the plans -- the text encoder and the VAE decoder stay where they were)
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_PATCHED = "__dwg_bound__"


def _pkg():
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    import dwg_import  # noqa: F401
    import dreamwaltz_g_amd
    return dreamwaltz_g_amd


# --------------------------------------------------------------------------------------------------------------------------------------
# B3: avatar
# --------------------------------------------------------------------------------------------------------------------------------------
def bind_avatar(ref_avatar):
    """reference DreamWaltzG -> HIP-backed DreamWaltzG with the same tensors; anything else is returned as it is."""
    if type(ref_avatar).__name__ != "DreamWaltzG" or getattr(ref_avatar, _PATCHED, False):
        return ref_avatar
    _pkg()
    from dreamwaltz_g_amd.avatar import DreamWaltzG
    av = DreamWaltzG.from_reference(ref_avatar)
    setattr(av, _PATCHED, True)
    return av


def _patch_avatar_module(mod):
    orig = mod.build_gaussian_avatar
    if getattr(orig, _PATCHED, False):
        return

    def build_gaussian_avatar(*args, **kwargs):
        return bind_avatar(orig(*args, **kwargs))
    build_gaussian_avatar.__doc__ = orig.__doc__
    setattr(build_gaussian_avatar, _PATCHED, True)
    build_gaussian_avatar.__wrapped__ = orig
    mod.build_gaussian_avatar = build_gaussian_avatar


# --------------------------------------------------------------------------------------------------------------------------------------
# B5: scene
# --------------------------------------------------------------------------------------------------------------------------------------
def _patch_scene_module(mod):
    orig = mod.build_scene
    if getattr(orig, _PATCHED, False):
        return

    def build_scene(cfg, avatar):
        items = avatar if isinstance(avatar, (list, tuple)) else [avatar]
        if not all(getattr(a, _PATCHED, False) for a in items):
            return orig(cfg=cfg, avatar=avatar)            # not ours (another gs_type): the reference's scene
        _pkg()
        from dreamwaltz_g_amd.scene import Scene
        import torch
        r = cfg.render
        if r.use_mlp_background or r.use_video_background or r.use_gs_background:
            raise NotImplementedError("learned / video / Gaussian backgrounds are outside the bound hot path (scene.py:226-237)")
        # exact pair sizing through the 16-byte read-back per frame, like the reference's CUDA extension: the reference's loop body does
        # not know about re-rendering a truncated frame
        return Scene(cfg, avatar, background=None, async_pair_count=False).to(torch.device(cfg.device))
    setattr(build_scene, _PATCHED, True)
    build_scene.__wrapped__ = orig
    mod.build_scene = build_scene


# --------------------------------------------------------------------------------------------------------------------------------------
# B4: guidance
# --------------------------------------------------------------------------------------------------------------------------------------
def plan_dtype_for(ref, dtype=None):
    """Storage type of the HIP plans for a constructed reference guidance object: the explicit argument, else DWG_BIND_DTYPE, else the
    precision the reference itself loaded its pipeline in (core/guidance/basic.py:233 `torch_dtype=self.torch_dtype`): torch.float32 (the
    default of every shipped recipe) -> "f32x", whose results are the fp32 ones (eps 3e-6 vs the fp32 oracle); torch.float16
    (`--guide.dtype fp16`, basic.py:24-27) -> "f16".  The binding never narrows the user's arithmetic on its own: bf16 plans (eps 1.5 % off)
    only through the argument / DWG_BIND_DTYPE=bf16."""
    import torch
    return dtype or os.environ.get("DWG_BIND_DTYPE") or ("f16" if getattr(ref, "torch_dtype", None) is torch.float16 else "f32x")


def bind_guidance(ref, dtype=None, keep_modules=None):
    """Binds `_predict` and `encode_images` of a constructed reference ControlNetScoreDistillation to HIP plans fed from the state_dict()s
    of its loaded diffusers modules (pipe.unet, controlnet, pipe.vae).  Returns `ref` (the same object)."""
    if getattr(ref, _PATCHED, False):
        return ref
    _pkg()
    import torch
    from dreamwaltz_g_amd import guidance as gd, sd15
    dtype = plan_dtype_for(ref, dtype)
    unet, cnet, vae = ref.pipe.unet, ref.controlnet, ref.pipe.vae
    if type(cnet).__name__ == "MultiControlNetModel":
        raise NotImplementedError("MultiControlNetModel (several condition types at once)")
    ucfg, vcfg = sd15.unet_config_from(getattr(unet, "config", None)), sd15.vae_config_from(getattr(vae, "config", None))
    f32 = lambda sd: {k: v.detach().float().cpu() for k, v in sd.items()}       # noqa: E731
    vsd = {k: v for k, v in f32(vae.state_dict()).items() if k.startswith(("encoder.", "quant_conv."))}
    hip = gd.ControlNetScoreDistillation(ref.device, unet_cfg=ucfg, vae_cfg=vcfg, unet_sd=f32(unet.state_dict()), controlnet_sd=f32(cnet.state_dict()),
                                         vae_sd=vsd, image_hw=int(ref.default_image_size), cfg=ref.cfg, dtype=dtype)
    if os.environ.get("DWG_BIND_EAGER") != "1":
        stream_ok = torch.cuda.current_stream(ref.device).cuda_stream != 0
        if stream_ok:
            hip.capture_graphs()

    # f32x range safety (round 5): the split-precision plans saturate at +-65504 and thin out below 6.1e-5 where the reference's fp32
    # pipeline (core/guidance/basic.py:233) does neither.  The stored activations of a call are scanned after EVERY one of the first
    # DWG_BIND_RANGE_CHECK_FIRST calls (default 20: a run that saturates does so at its first high-noise timesteps, and must not train
    # 200 steps on clipped values before anything is said) and every DWG_BIND_RANGE_CHECK_EVERY calls after that (default 200; 0: never);
    # a hit is reported ONCE per layer set with the layers' names and the documented way out.  DWG_BIND_RANGE_STRICT=1 raises instead.
    check_every = int(os.environ.get("DWG_BIND_RANGE_CHECK_EVERY", "200"))
    check_first = int(os.environ.get("DWG_BIND_RANGE_CHECK_FIRST", "20")) if check_every > 0 else 0
    strict = os.environ.get("DWG_BIND_RANGE_STRICT") == "1"
    state = {"calls": 0, "warned": set(), "checks": 0}

    def _range_check():
        rep = hip.range_report()
        ref.hip_range_report = rep
        state["checks"] += 1
        ref.hip_range_checks = state["checks"]
        if rep is None or rep["ok"]:
            return
        if strict:
            raise FloatingPointError("dwg_bind: the f32x plans saturated / produced non-finite values at call %d (DWG_BIND_RANGE_STRICT=1); "
                                     "rerun with DWG_BIND_DTYPE=f32" % state["calls"])
        layers = tuple(sorted({"%s:%s" % (n, d["layer"]) for n in ("denoiser", "vae_forward", "vae_backward") for d in rep[n]["worst"]
                               if d["saturated"] or d["nonfinite"]}))
        if layers not in state["warned"]:
            state["warned"].add(layers)
            import warnings
            warnings.warn("dwg_bind: the f32x (split fp16) plans SATURATED at +-65504 or produced non-finite values in %s -- the reference's fp32 "
                          "pipeline would not have; rerun with DWG_BIND_DTYPE=f32 (exact-f32 MFMA plans, ~2.5x slower)" % (", ".join(layers) or "the weights"),
                          RuntimeWarning, stacklevel=3)

    def _predict(self, latents_model_input, text_embeddings, cond_inputs):
        hip.timestep = self.timestep                       # controlnet.py:83-114 reads self.timestep
        out = hip._predict(latents_model_input, text_embeddings, cond_inputs).to(latents_model_input.dtype)
        state["calls"] += 1
        if check_every > 0 and hip.dtype_name == "f32x" and (state["calls"] <= check_first or state["calls"] % check_every == 0):
            _range_check()
        return out

    def encode_images(self, images):
        if not isinstance(images, torch.Tensor):           # PIL inputs (visualisation only): the reference's own path
            return type(self).encode_images(self, images)
        return hip.encode_images(images)                   # normalise (2x-1) + encoder + posterior sample + scaling factor, differentiable

    ref._predict = types.MethodType(_predict, ref)
    ref.encode_images = types.MethodType(encode_images, ref)
    ref.hip = hip
    keep = keep_modules if keep_modules is not None else os.environ.get("DWG_BIND_KEEP_MODULES") == "1"
    if not keep:
        for m in (unet, cnet):                             # their weights now live in the plans
            if hasattr(m, "to"):
                m.to("cpu")
    setattr(ref, _PATCHED, True)
    return ref


def _patch_guidance_module(mod):
    cls = mod.ControlNetScoreDistillation
    if getattr(cls.__init__, _PATCHED, False):
        return
    orig_init = cls.__init__

    def __init__(self, *args, **kwargs):
        orig_init(self, *args, **kwargs)
        if type(self) is cls:                               # not the SDXL subclass family
            bind_guidance(self)
    setattr(__init__, _PATCHED, True)
    __init__.__wrapped__ = orig_init
    cls.__init__ = __init__


# --------------------------------------------------------------------------------------------------------------------------------------
# post-import hooks
# --------------------------------------------------------------------------------------------------------------------------------------
HOOKS = {"core.system.avatar": _patch_avatar_module, "core.system.scene": _patch_scene_module,
         "core.guidance.controlnet": _patch_guidance_module}


class _HookLoader(importlib.abc.Loader):
    def __init__(self, inner, hook):
        self.inner, self.hook = inner, hook

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        self.hook(module)

    def __getattr__(self, name):
        return getattr(self.inner, name)


class _HookFinder(importlib.abc.MetaPathFinder):
    def __init__(self):
        self.busy = set()

    def find_spec(self, fullname, path, target=None):
        if fullname not in HOOKS or fullname in self.busy:
            return None
        self.busy.add(fullname)
        try:
            spec = None
            for finder in sys.meta_path:
                if finder is self or not hasattr(finder, "find_spec"):
                    continue
                spec = finder.find_spec(fullname, path, target)
                if spec is not None:
                    break
        finally:
            self.busy.discard(fullname)
        if spec is None or spec.loader is None:
            return None
        spec.loader = _HookLoader(spec.loader, HOOKS[fullname])
        return spec


def install():
    """Idempotent.  Modules of HOOKS that are already imported are patched right away, the others right after their import."""
    for d in (_HERE, _ROOT):
        if d not in sys.path:
            sys.path.insert(0, d)
    if not any(isinstance(f, _HookFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _HookFinder())
    for name, hook in HOOKS.items():
        if name in sys.modules:
            hook(sys.modules[name])
    return True


def uninstall():
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _HookFinder)]
    for name in HOOKS:
        mod = sys.modules.get(name)
        if mod is None:
            continue
        for attr in ("build_gaussian_avatar", "build_scene"):
            f = getattr(mod, attr, None)
            if f is not None and getattr(f, _PATCHED, False):
                setattr(mod, attr, f.__wrapped__)
        cls = getattr(mod, "ControlNetScoreDistillation", None)
        if cls is not None and getattr(cls.__init__, _PATCHED, False):
            cls.__init__ = cls.__init__.__wrapped__
