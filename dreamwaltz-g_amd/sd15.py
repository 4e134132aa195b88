"""SD-1.5 UNet + ControlNet + VAE-encoder graphs on the HIP kernels (boundary B4).

The reference obtains these networks from `diffusers` (UNet2DConditionModel, ControlNetModel, AutoencoderKL.encoder; call
sites /root/reference/core/guidance/controlnet.py:98-114, vae.py:34-40, basic.py:147,176,208).  diffusers is a third-party
package that is not installed here; the layer graph below follows its published SD-1.5 architecture [3P-memory, SURVEY G9]
and uses diffusers' state_dict key names, so a real checkpoint converts with `convert_state_dict()`.

Execution model: every network is compiled ONCE into a static plan -- a flat list of pre-built C-ABI calls over
pre-allocated NHWC bf16 buffers (shapes never change during SDS) -- so a step is a tight loop of library calls on one HIP
stream with no allocation, no shape logic and no host synchronisation; the plan is hipGraph-capturable as is.
"""
import ctypes
import os
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, gemm, xfmt

BF16 = torch.bfloat16

# Storage element type of a plan's activations and weights (accumulation is fp32 in all of them):
#   "bf16"  default: MFMA bf16 (v_mfma_f32_32x32x16_bf16), direct-to-LDS tiles, flash attention
#   "f32"   the precision the reference runs the 3DGS stage in (/root/reference/configs/__init__.py:236,241; `--optim.fp16` is only passed
#           to the NeRF stages of scripts/train_w_expr.sh): exact-f32 MFMA (v_mfma_f32_32x32x2_f32, peak 157 TFLOP/s), attention as
#           QK^T -> row softmax -> PV.  It is the full-precision reference the bf16 plans are measured against ON THE GPU.
#   "f16"   the reference's `--guide.dtype fp16` (core/guidance/basic.py:24-27,233) / autocast storage type (configs/__init__.py:462): the same kernels as "bf16" compiled for _Float16 operands
#           (csrc/gemm_f16.hip, attention_f16.hip: v_mfma_f32_32x32x16_f16 -- same MFMA rate), 10 mantissa bits instead of 7, range 65504.
#   "f32x"  SPLIT PRECISION -- the reference's fp32 results at the 16-bit MFMA rate: every activation and weight is held as hi + 2^-11 lo fp16
#           halves (32 bytes per 8 channels, csrc/dwg_xfmt.h; torch.int32-typed tensors here, xfmt.py), every product is three
#           v_mfma_f32_32x32x16_f16 with fp32 accumulation (csrc/gemm_x.hip, attention_x.hip), norm / softmax / element-wise layers decode and
#           re-split in registers.  22 significand bits end to end: eps and SDS gradients agree with the fp32 oracle like the exact "f32"
#           plans do (tests/test_sd15_f32x_gpu.py), at 2.5-4x their speed.
TORCH_DTYPE = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16, "f32x": xfmt.DTYPE}
DT_CODE = {"f32": 0, "bf16": 1, "f16": 2, "f32x": 3}          # DWG_DTYPE_* (include/dwg_types.h)


def dtype_name(dtype) -> str:
    names = {"bf16": "bf16", "bfloat16": "bf16", torch.bfloat16: "bf16", "f32": "f32", "fp32": "f32", "float32": "f32", torch.float32: "f32",
             "f16": "f16", "fp16": "f16", "float16": "f16", "half": "f16", torch.float16: "f16", "f32x": "f32x", "fp32x": "f32x"}
    if dtype not in names:
        raise ValueError("plan dtype must be one of bf16 / f32 / f16 / f32x, got %r" % (dtype,))
    return names[dtype]


# ----------------------------------------------------------------------------------------------------------------------
# configuration + parameter inventory (diffusers key names / shapes)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: int = 8
    cross_dim: int = 768
    groups: int = 32
    attn_blocks: Tuple[bool, ...] = (True, True, True, False)
    cond_channels: Tuple[int, ...] = (16, 32, 96, 256)      # ControlNet conditioning embedding
    cond_in_channels: int = 3

    @property
    def temb_dim(self):
        return self.block_out_channels[0] * 4


@dataclass
class VAEConfig:
    in_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    groups: int = 32
    scaling_factor: float = 0.18215


def _cfg_get(config, key, default):
    if config is None:
        return default
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


def unet_config_from(config) -> "UNetConfig":
    """UNetConfig from a diffusers `UNet2DConditionModel.config` (attribute or dict access; None -> the SD-1.5 defaults).  Anything that is
    not the SD-1.x layer graph this file builds (one transformer block per attention site, heads = attention_head_dim, GEGLU, no
    class / addition embeddings) is rejected."""
    d = UNetConfig()
    if config is None:
        return d
    boc = tuple(int(c) for c in _cfg_get(config, "block_out_channels", d.block_out_channels))
    down = tuple(_cfg_get(config, "down_block_types", ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)))
    heads = _cfg_get(config, "attention_head_dim", d.heads)
    if isinstance(heads, (list, tuple)):
        if len(set(heads)) != 1:
            raise NotImplementedError("per-level attention_head_dim %r (SDXL-style UNets are outside this path)" % (heads,))
        heads = heads[0]
    for key, want in (("transformer_layers_per_block", 1), ("class_embed_type", None), ("addition_embed_type", None), ("use_linear_projection", False),
                      ("only_cross_attention", False), ("dual_cross_attention", False)):
        got = _cfg_get(config, key, want)
        if got != want:
            raise NotImplementedError("UNet config %s=%r (SD-1.x layer graph only)" % (key, got))
    return UNetConfig(in_channels=int(_cfg_get(config, "in_channels", d.in_channels)), out_channels=int(_cfg_get(config, "out_channels", d.out_channels)),
                      block_out_channels=boc, layers_per_block=int(_cfg_get(config, "layers_per_block", d.layers_per_block)), heads=int(heads),
                      cross_dim=int(_cfg_get(config, "cross_attention_dim", d.cross_dim)), groups=int(_cfg_get(config, "norm_num_groups", d.groups)),
                      attn_blocks=tuple(t.startswith("CrossAttn") for t in down))


def vae_config_from(config) -> "VAEConfig":
    """VAEConfig from a diffusers `AutoencoderKL.config` (None -> SD-1.5's)."""
    d = VAEConfig()
    if config is None:
        return d
    return VAEConfig(in_channels=int(_cfg_get(config, "in_channels", d.in_channels)),
                     block_out_channels=tuple(int(c) for c in _cfg_get(config, "block_out_channels", d.block_out_channels)),
                     layers_per_block=int(_cfg_get(config, "layers_per_block", d.layers_per_block)),
                     latent_channels=int(_cfg_get(config, "latent_channels", d.latent_channels)), groups=int(_cfg_get(config, "norm_num_groups", d.groups)),
                     scaling_factor=float(_cfg_get(config, "scaling_factor", d.scaling_factor)))


def _resnet_shapes(sh, pre, cin, cout, temb):
    sh[pre + ".norm1.weight"] = (cin,); sh[pre + ".norm1.bias"] = (cin,)
    sh[pre + ".conv1.weight"] = (cout, cin, 3, 3); sh[pre + ".conv1.bias"] = (cout,)
    if temb:
        sh[pre + ".time_emb_proj.weight"] = (cout, temb); sh[pre + ".time_emb_proj.bias"] = (cout,)
    sh[pre + ".norm2.weight"] = (cout,); sh[pre + ".norm2.bias"] = (cout,)
    sh[pre + ".conv2.weight"] = (cout, cout, 3, 3); sh[pre + ".conv2.bias"] = (cout,)
    if cin != cout:
        sh[pre + ".conv_shortcut.weight"] = (cout, cin, 1, 1); sh[pre + ".conv_shortcut.bias"] = (cout,)


def _transformer_shapes(sh, pre, c, cross):
    sh[pre + ".norm.weight"] = (c,); sh[pre + ".norm.bias"] = (c,)
    sh[pre + ".proj_in.weight"] = (c, c, 1, 1); sh[pre + ".proj_in.bias"] = (c,)
    t = pre + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        sh["%s.%s.weight" % (t, n)] = (c,); sh["%s.%s.bias" % (t, n)] = (c,)
    for a, kd in (("attn1", c), ("attn2", cross)):
        sh["%s.%s.to_q.weight" % (t, a)] = (c, c)
        sh["%s.%s.to_k.weight" % (t, a)] = (c, kd)
        sh["%s.%s.to_v.weight" % (t, a)] = (c, kd)
        sh["%s.%s.to_out.0.weight" % (t, a)] = (c, c); sh["%s.%s.to_out.0.bias" % (t, a)] = (c,)
    sh[t + ".ff.net.0.proj.weight"] = (8 * c, c); sh[t + ".ff.net.0.proj.bias"] = (8 * c,)
    sh[t + ".ff.net.2.weight"] = (c, 4 * c); sh[t + ".ff.net.2.bias"] = (c,)
    sh[pre + ".proj_out.weight"] = (c, c, 1, 1); sh[pre + ".proj_out.bias"] = (c,)


def _encoder_shapes(sh, cfg: UNetConfig):
    """conv_in, time embedding, down blocks, mid block: shared by UNet and ControlNet."""
    boc = cfg.block_out_channels
    sh["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3); sh["conv_in.bias"] = (boc[0],)
    sh["time_embedding.linear_1.weight"] = (cfg.temb_dim, boc[0]); sh["time_embedding.linear_1.bias"] = (cfg.temb_dim,)
    sh["time_embedding.linear_2.weight"] = (cfg.temb_dim, cfg.temb_dim); sh["time_embedding.linear_2.bias"] = (cfg.temb_dim,)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet_shapes(sh, "down_blocks.%d.resnets.%d" % (i, j), cin, cout, cfg.temb_dim)
            if cfg.attn_blocks[i]:
                _transformer_shapes(sh, "down_blocks.%d.attentions.%d" % (i, j), cout, cfg.cross_dim)
            cin = cout
        if i != len(boc) - 1:
            sh["down_blocks.%d.downsamplers.0.conv.weight" % i] = (cout, cout, 3, 3)
            sh["down_blocks.%d.downsamplers.0.conv.bias" % i] = (cout,)
    c = boc[-1]
    _resnet_shapes(sh, "mid_block.resnets.0", c, c, cfg.temb_dim)
    _transformer_shapes(sh, "mid_block.attentions.0", c, cfg.cross_dim)
    _resnet_shapes(sh, "mid_block.resnets.1", c, c, cfg.temb_dim)


def _skip_channels(cfg: UNetConfig) -> List[int]:
    boc = cfg.block_out_channels
    ch = [boc[0]]
    for i, cout in enumerate(boc):
        ch += [cout] * cfg.layers_per_block
        if i != len(boc) - 1:
            ch.append(cout)
    return ch


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    sh = OrderedDict()
    _encoder_shapes(sh, cfg)
    boc = cfg.block_out_channels
    skips = _skip_channels(cfg)
    rev = list(reversed(boc))
    prev = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            skip = skips.pop()
            _resnet_shapes(sh, "up_blocks.%d.resnets.%d" % (i, j), prev + skip, cout, cfg.temb_dim)
            if list(reversed(cfg.attn_blocks))[i]:
                _transformer_shapes(sh, "up_blocks.%d.attentions.%d" % (i, j), cout, cfg.cross_dim)
            prev = cout
        if i != len(rev) - 1:
            sh["up_blocks.%d.upsamplers.0.conv.weight" % i] = (cout, cout, 3, 3)
            sh["up_blocks.%d.upsamplers.0.conv.bias" % i] = (cout,)
    sh["conv_norm_out.weight"] = (boc[0],); sh["conv_norm_out.bias"] = (boc[0],)
    sh["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3); sh["conv_out.bias"] = (cfg.out_channels,)
    return sh


def controlnet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    sh = OrderedDict()
    _encoder_shapes(sh, cfg)
    cc = cfg.cond_channels
    e = "controlnet_cond_embedding"
    sh[e + ".conv_in.weight"] = (cc[0], cfg.cond_in_channels, 3, 3); sh[e + ".conv_in.bias"] = (cc[0],)
    k = 0
    for i in range(len(cc) - 1):
        sh["%s.blocks.%d.weight" % (e, k)] = (cc[i], cc[i], 3, 3); sh["%s.blocks.%d.bias" % (e, k)] = (cc[i],); k += 1
        sh["%s.blocks.%d.weight" % (e, k)] = (cc[i + 1], cc[i], 3, 3); sh["%s.blocks.%d.bias" % (e, k)] = (cc[i + 1],); k += 1
    sh[e + ".conv_out.weight"] = (cfg.block_out_channels[0], cc[-1], 3, 3); sh[e + ".conv_out.bias"] = (cfg.block_out_channels[0],)
    for k, c in enumerate(_skip_channels(cfg)):
        sh["controlnet_down_blocks.%d.weight" % k] = (c, c, 1, 1); sh["controlnet_down_blocks.%d.bias" % k] = (c,)
    c = cfg.block_out_channels[-1]
    sh["controlnet_mid_block.weight"] = (c, c, 1, 1); sh["controlnet_mid_block.bias"] = (c,)
    return sh


def vae_encoder_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    sh = OrderedDict()
    boc = cfg.block_out_channels
    sh["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3); sh["encoder.conv_in.bias"] = (boc[0],)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet_shapes(sh, "encoder.down_blocks.%d.resnets.%d" % (i, j), cin, cout, 0)
            cin = cout
        if i != len(boc) - 1:
            sh["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i] = (cout, cout, 3, 3)
            sh["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (cout,)
    c = boc[-1]
    _resnet_shapes(sh, "encoder.mid_block.resnets.0", c, c, 0)
    a = "encoder.mid_block.attentions.0"
    sh[a + ".group_norm.weight"] = (c,); sh[a + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh["%s.%s.weight" % (a, n)] = (c, c); sh["%s.%s.bias" % (a, n)] = (c,)
    _resnet_shapes(sh, "encoder.mid_block.resnets.1", c, c, 0)
    sh["encoder.conv_norm_out.weight"] = (c,); sh["encoder.conv_norm_out.bias"] = (c,)
    sh["encoder.conv_out.weight"] = (2 * cfg.latent_channels, c, 3, 3); sh["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    sh["quant_conv.weight"] = (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1)
    sh["quant_conv.bias"] = (2 * cfg.latent_channels,)
    return sh


def random_state_dict(shapes, seed=0, gain=1.0) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights of the given architecture (no checkpoints on the GPU box: BASELINE `data: synthetic`)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shp in shapes.items():
        if "norm" in name and name.endswith("weight") and len(shp) == 1:
            sd[name] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("bias"):
            sd[name] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            sd[name] = torch.randn(shp, generator=g) * (gain / math.sqrt(fan_in))
    return sd


# ----------------------------------------------------------------------------------------------------------------------
# plan machinery
# ----------------------------------------------------------------------------------------------------------------------
def _pad8(c):
    return (c + 7) // 8 * 8


class Plan:
    """Flat list of prebuilt library calls; run() is the only per-step Python work."""

    def __init__(self, device, dtype="bf16"):
        self.device = device
        self.dtype_name = dtype_name(dtype)
        self.dtype, self.dt = TORCH_DTYPE[self.dtype_name], DT_CODE[self.dtype_name]
        self.esize = 4 if self.dtype_name in ("f32", "f32x") else 2
        self.is_x = self.dtype_name == "f32x"
        self.ops = []
        self.keep = []          # descriptors / tensors kept alive
        self.tags = []
        self.graph = None
        self.use_graph = True
        self.has_py_ops = False
        self.flops = {}         # algorithmic flops per kernel label (2*M*N*K*batch), for the roofline report
        self._lib = _lib.lib()
        self.branch = 0         # ops are tagged with the branch (stream) they run on; 0 = the caller's stream
        self._side = {}         # branch id -> library-owned side stream
        self.scope = ""         # the layer being built (set by the Builder): names the buffers created inside it (range_report)
        self.labels = []        # one per kept buffer

    # -- independent branches (run concurrently on side streams; parallel paths of the captured graph) -------------------
    def fork(self, b):
        """Branch b starts here: it is ordered after everything recorded on branch 0 so far."""
        self.ops.append(("fork", 0, b))

    def join(self, b):
        """Branch 0 continues only after everything recorded on branch b."""
        self.ops.append(("fork", b, 0))

    def on_branch(self, b):
        plan = self

        class _Ctx:
            def __enter__(self):
                self.prev, plan.branch = plan.branch, b

            def __exit__(self, *exc):
                plan.branch = self.prev
        return _Ctx()

    def _stream_of(self, b, main):
        if b == 0:
            return main
        if b not in self._side:
            h = ctypes.c_void_p()
            _lib.check(self._lib.dwg_stream_create(ctypes.byref(h)), "dwg_stream_create")
            self._side[b] = h
        return self._side[b]

    def _issue(self, main):
        for op in self.ops:
            if op[0] == "fork":
                rc = self._lib.dwg_stream_fork(self._stream_of(op[1], main), self._stream_of(op[2], main))
            else:
                rc = op[1](self._stream_of(op[0], main))
            if rc:
                raise RuntimeError("plan op failed with DWG error %s" % rc)

    def buf(self, *shape, dtype=None, zero=False):
        import os
        dtype = self.dtype if dtype is None else dtype
        zero = zero or os.environ.get("DWG_PLAN_ZERO") == "1"
        t = (torch.zeros if zero else torch.empty)(*shape, device=self.device, dtype=dtype)
        self.keep.append(t)
        self.labels.append((len(self.keep) - 1, self.scope or "op%d" % len(self.ops)))
        self.tags.append((len(self.ops), tuple(shape)))
        return t

    def range_report(self, top=8):
        """Range telemetry of an f32x plan (round 5; verdict round 4, item 3): every activation buffer the plan owns is scanned AFTER a run
        (cold path: one small launch per buffer, one read-back) for values the format cannot hold with fp32-grade precision -- hi halves at
        +-65504 (the split saturates instead of overflowing, csrc/dwg_xfmt.h), non-zero values below fp16's normal range (|x| < 6.1e-5: fewer
        than 22 significand bits), non-finite values.  Every layer output has its own buffer, so a hit names its layer.  The reference runs
        this stage in fp32 (/root/reference/configs/__init__.py:236,241): `saturated` > 0 means the results left its range and
        DWG_BIND_DTYPE=f32 (exact-f32 MFMA plans) is the fallback.  Other plan types: None."""
        if not self.is_x:
            return None
        rows = [(i, lab) for (i, lab) in self.labels if self.keep[i].dtype == xfmt.DTYPE and self.keep[i].numel() % 8 == 0 and self.keep[i].numel() > 0
                and self.keep[i].is_contiguous()]
        if not rows:
            return {"tensors": 0, "elements": 0, "saturated": 0, "subnormal": 0, "nonfinite": 0, "max_abs": 0.0, "worst": []}
        cnt = torch.zeros(len(rows), 5, dtype=torch.int64, device=self.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for r, (i, _) in enumerate(rows):
            t = self.keep[i]
            _lib.check(self._lib.dwg_xfmt_range_scan(t.numel(), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(cnt[r].data_ptr()), st),
                       "dwg_xfmt_range_scan")
        c = cnt.cpu()
        amax = c[:, 3].to(torch.int32).view(torch.float32)        # fp32 bits of max |x| (non-negative: fits int32)
        per = [{"layer": lab, "shape": tuple(self.keep[i].shape), "saturated": int(c[r, 0]), "subnormal": int(c[r, 1]), "nonfinite": int(c[r, 2]),
                "max_abs": float(amax[r])} for r, (i, lab) in enumerate(rows)]
        worst = sorted(per, key=lambda d: (-(d["saturated"] + d["nonfinite"]), -d["max_abs"]))[:top]
        return {"tensors": len(rows), "elements": int(c[:, 4].sum()), "saturated": int(c[:, 0].sum()), "subnormal": int(c[:, 1].sum()),
                "nonfinite": int(c[:, 2].sum()), "max_abs": float(amax.max()), "worst": worst}

    # -- host <-> plan buffers: fp32 values in, the plan's storage type out (and back) ------------------------------------------
    def store(self, dst, src, stage=None):
        """dst (a whole, contiguous activation buffer of this plan) <- fp32 values.  `src` has dst's logical shape, or fewer channels: the rest
        of `stage` (a zero-initialised fp32 tensor of dst's shape, kept by the caller) pads it -- an f32x buffer is written in whole
        8-channel groups by ONE pack launch (dwg_xfmt_pack) on the current stream."""
        if not self.is_x:
            (dst if src.shape == dst.shape else dst[..., :src.shape[-1]]).copy_(src)
            return
        if stage is not None:
            stage[..., :src.shape[-1]].copy_(src)
            src = stage
        elif src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        assert src.shape == dst.shape and dst.is_contiguous(), (src.shape, dst.shape)
        _lib.check(self._lib.dwg_xfmt_pack(src.numel(), ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "dwg_xfmt_pack")

    def load(self, t):
        """fp32 copy of a whole, contiguous activation buffer of this plan."""
        if not (self.is_x and t.dtype == xfmt.DTYPE):
            return t.float()
        out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
        _lib.check(self._lib.dwg_xfmt_unpack(t.numel(), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "dwg_xfmt_unpack")
        return out

    def stage_like(self, buf):
        """Zeroed fp32 staging tensor for `store` into an f32x buffer from fewer channels (None for the other plan types)."""
        return torch.zeros(buf.shape, device=self.device, dtype=torch.float32) if self.is_x else None

    def add_gemm(self, desc, allow_split=True):
        if allow_split and desc.batch1 * desc.batch2 == 1 and desc.splitk <= 1 and not desc.accumulate:
            desc.splitk = 0     # library picks a split-K factor for shapes that cannot fill the chip
            need = self._lib.dwg_gemm_workspace_bytes(ctypes.byref(desc))
            if need > 0:
                # this call site's own workspace.  DWG_SPLITK_FUSED=1: with a zeroed counter header -- the last slice of a tile reduces inside the
                # GEMM kernel (bit-identical; measured SLOWER on MI355X, 29.1 vs 25.0 ms per step: DESIGN.md "Measured (round 6)")
                import os
                fused = os.environ.get("DWG_SPLITK_FUSED") == "1"
                ws = self.buf(int(need) // 4, dtype=torch.float32, zero=fused)
                desc.workspace, desc.workspace_bytes, desc.workspace_counters = ws.data_ptr(), int(need), int(fused)
            else:
                desc.splitk = 1
        self.keep.append(desc)
        label = desc.name.decode() if desc.name else ("conv_igemm" if desc.conv_enabled else "gemm")
        self.flops[label] = self.flops.get(label, 0.0) + 2.0 * desc.M * desc.N * desc.K * desc.batch1 * desc.batch2
        fn, ref = self._lib.dwg_gemm, ctypes.byref(desc)
        self.ops.append((self.branch, lambda s, fn=fn, ref=ref: fn(ref, s)))

    def add_call(self, fn, *args):
        self.ops.append((self.branch, lambda s, fn=fn, args=args: fn(*args, s)))

    def add_py(self, f):
        self.has_py_ops = True
        self.ops.append((self.branch, lambda s, f=f: (f(), 0)[1]))

    def run_eager(self):
        self._issue(ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def capture(self):
        """Record the whole plan into one hipGraph (launch-bound inner loop -> a single hipGraphLaunch per step).  The plan
        contains only dwg_* launches on caller-owned buffers, so the capture needs nothing from the PyTorch allocator."""
        if self.graph is not None:
            return
        assert not self.has_py_ops, "plans with Python ops cannot be captured"
        _lib.prof_enable(False)
        self.run_eager()                    # warm-up (lazy kernel attribute initialisation) outside the capture
        torch.cuda.synchronize(self.device)
        self._cap_stream = torch.cuda.Stream(device=self.device)
        s = ctypes.c_void_p(self._cap_stream.cuda_stream)
        _lib.check(self._lib.dwg_graph_begin_capture(s), "dwg_graph_begin_capture")
        try:
            self._issue(s)
        finally:
            h = ctypes.c_void_p()
            rc = self._lib.dwg_graph_end_capture(s, ctypes.byref(h))
        _lib.check(rc, "dwg_graph_end_capture")
        self.graph = h

    def run(self):
        if self.graph is not None and self.use_graph:
            cur = torch.cuda.current_stream(self.device).cuda_stream
            if cur == 0:
                # hipGraphLaunch into the legacy default stream is NOT ordered against work already queued there (ROCm 7.2:
                # wrong results unless the device is idle).  Callers that want graph replay run under a real stream
                # (torch.cuda.set_stream(torch.cuda.Stream())); on the default stream the plan silently stays eager.
                self.run_eager()
                return
            _lib.check(self._lib.dwg_graph_launch(self.graph, ctypes.c_void_p(cur)), "dwg_graph_launch")
        else:
            self.run_eager()

    def run_debug(self):
        """Runs op by op and reports the first op after which any plan buffer holds a non-finite value (DWG_PLAN_ZERO=1)."""
        s = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        bufs = [t for t in self.keep if torch.is_tensor(t) and t.is_floating_point()]
        for i, op in enumerate(self.ops):
            rc = 0 if op[0] == "fork" else op[1](s)
            torch.cuda.synchronize()
            bad = [tuple(t.shape) for t in bufs if not torch.isfinite(t.float()).all()]
            if rc or bad:
                return i, rc, bad
        return None


class Weights:
    """Kernel-layout copies of a diffusers-format state_dict: conv [Cout,KH,KW,Cin(pad 8)] bf16, linear [out,in] bf16,
    biases / norm affine fp32."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, dtype="bf16"):
        self.device = device
        self.sd = sd
        self.cache = {}
        self.wdtype = TORCH_DTYPE[dtype_name(dtype)]         # storage type of the conv / linear weights (biases, norm affine: fp32)
        self.is_x = dtype_name(dtype) == "f32x"

    def _to_dev(self, w):
        """fp32 kernel-layout weight (contraction axis innermost) -> the plan's storage type on the device."""
        if self.is_x:
            # range telemetry of the weights (xfmt.pack clamps at +-65504 without a trace): counted once, at plan build
            a = w.abs()
            r = self.__dict__.setdefault("range", {"elements": 0, "saturated": 0, "subnormal": 0, "max_abs": 0.0})
            r["elements"] += a.numel(); r["saturated"] += int((a >= xfmt.X_MAX).sum()); r["subnormal"] += int(((a < 6.1035e-5) & (a > 0)).sum())
            r["max_abs"] = max(r["max_abs"], float(a.max()) if a.numel() else 0.0)
            return xfmt.pack(w.to(self.device)).contiguous()
        return w.to(self.device, self.wdtype).contiguous()

    def conv(self, name, flip_for_dgrad=False):
        key = (name, flip_for_dgrad)
        if key not in self.cache:
            w = self.sd[name + ".weight"].float()
            if flip_for_dgrad:   # input-gradient of a conv = conv with spatially flipped, channel-transposed weights
                w = w.flip(2, 3).permute(1, 0, 2, 3)
            cout, cin = w.shape[0], w.shape[1]
            wp = torch.zeros(_pad8(cout) if flip_for_dgrad else cout, w.shape[2], w.shape[3], _pad8(cin))
            wp[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
            self.cache[key] = self._to_dev(wp)
        return self.cache[key]

    def conv_dgrad_s2(self, name, py, px):
        """Tap subset of a 3x3 stride-2 convolution's weights that reaches the input pixels of parity (py, px), laid out as the
        weights [Cin_x, KH', KW', Cout_y] of the stride-1 convolution over the incoming gradient that produces them:
        dx[2i+py, 2j+px] = sum_{ty,tx} W[ty,tx] . dy[i - pad + ty, j - pad + tx];  even parity: taps ky = (2, 0) with pad 1,
        odd parity: tap ky = 1 with pad 0 (the forward reads x[2i+ky, 2j+kx])."""
        key = ("dgrad_s2", name, py, px)
        if key not in self.cache:
            w = self.sd[name + ".weight"].float()                        # [Cout_y, Cin_x, 3, 3]
            kmap = {0: [2, 0], 1: [1]}
            kys, kxs = kmap[py], kmap[px]
            cy, cx = w.shape[0], w.shape[1]
            wp = torch.zeros(_pad8(cx), len(kys), len(kxs), _pad8(cy))
            for ty, ky in enumerate(kys):
                for tx, kx in enumerate(kxs):
                    wp[:cx, ty, tx, :cy] = w[:, :, ky, kx].t()
            self.cache[key] = self._to_dev(wp)
        return self.cache[key]

    def lin(self, *names):
        key = ("lin",) + names
        if key not in self.cache:
            self.cache[key] = self._to_dev(torch.cat([self.sd[n + ".weight"].float().reshape(self.sd[n + ".weight"].shape[0], -1)
                                                      for n in names], 0))
        return self.cache[key]

    def lin_t(self, name):
        """The transposed copy [in, out] of a linear weight: the backward product dy @ W as a K-contiguous GEMM (B(n, k) = W^T[n][k])."""
        key = ("lin_t", name)
        if key not in self.cache:
            w = self.sd[name + ".weight"].float()
            self.cache[key] = self._to_dev(w.reshape(w.shape[0], -1).t().contiguous())
        return self.cache[key]

    def lin_geglu(self, name):
        """GEGLU projection [8C, C] (rows: hidden | gate) re-ordered in 32-row blocks [hidden_q | gate_q] for the fused
        GEGLU-pair epilogue of dwg_gemm; returns (weight bf16, bias fp32) in that order."""
        key = ("geglu", name)
        if key not in self.cache:
            w = self.sd[name + ".weight"].float(); b = self.sd[name + ".bias"].float()
            F = w.shape[0] // 2
            idx = torch.arange(F).view(-1, 32)
            perm = torch.cat([idx, idx + F], dim=1).reshape(-1)          # [q*64 + 0..31] = hidden, [q*64 + 32..63] = gate
            self.cache[key] = (self._to_dev(w[perm]), b[perm].to(self.device).contiguous())
        return self.cache[key]

    def f32(self, name):
        if name not in self.cache:
            self.cache[name] = self.sd[name].float().to(self.device).contiguous()
        return self.cache[name]

    def bias_cat(self, *names):
        key = ("bias",) + names
        if key not in self.cache:
            self.cache[key] = torch.cat([self.sd[n].float() for n in names], 0).to(self.device).contiguous()
        return self.cache[key]


class Builder:
    def __init__(self, plan: Plan, w: Weights, groups=32, prefix="net"):
        self.p, self.w, self.groups, self.prefix = plan, w, groups, prefix
        self.L = _lib.lib()

    # -- contractions -------------------------------------------------------------------------------------------
    def _conv_tag(self, tag, KH, Ho, stride, dil, KW=None):
        if tag:
            return tag
        KW = KH if KW is None else KW
        base = "conv3x3" if KH == 3 else ("conv1x1" if KH * KW == 1 else "conv%dx%d" % (KH, KW))
        return "%s%s_%s_r%d" % (base, "s2" if stride == 2 else ("T" if dil > 1 else ""), self.prefix, Ho)

    def conv(self, x, name, stride=1, pad=1, act=None, residual=None, upsample=1, bias_img=None, out_dtype=None, out_hw=None,
             pad_tl=None, r_batch_bcast=False, weight=None, bias=True, in_dilation=1, tag=None):
        """x [B,H,W,C] NHWC bf16. bias_img: (tensor [B, ld] fp32, ld) per-image channel bias replacing the conv bias."""
        B, H, W, C = x.shape
        self.p.scope = name
        wt = self.w.conv(name) if weight is None else weight
        Cout, KH, KW, Cin = wt.shape
        assert Cin == C, (name, Cin, C)
        Hv, Wv = ((H - 1) * in_dilation + 1) * upsample, ((W - 1) * in_dilation + 1) * upsample
        pt, pl = (pad, pad) if pad_tl is None else pad_tl
        if out_hw is None:
            Ho, Wo = (Hv + 2 * pad - KH) // stride + 1, (Wv + 2 * pad - KW) // stride + 1
        else:
            Ho, Wo = out_hw
        y = self.p.buf(B, Ho, Wo, Cout, dtype=out_dtype)
        b = None
        if bias_img is None and bias:
            b = self.w.f32(name + ".bias")
            if b.numel() < Cout:   # dgrad weights are channel padded
                b = None
        conv = (C, H, W, Ho, Wo, KH, KW, stride, pt, pl, in_dilation)
        K = KH * KW * C
        if r_batch_bcast:   # residual is [Vr,Ho,Wo,Cout], image b of the batch adds residual b % Vr (the ControlNet hint of view b % Vr is
            # shared by the CFG entries of that view): a batched GEMM, one image per batch entry, batch = (B / Vr) x Vr
            Vr = int(residual.shape[0])
            assert B % Vr == 0, (B, Vr)
            d = gemm.gemm_raw(x, wt, y, Ho * Wo, Cout, K, (0, 1), (K, 1), Cout, bias=b, residual=residual, ldr=Cout, act=act,
                              batch=(B // Vr, Vr), a_batch=(Vr * H * W * C, H * W * C), c_batch=(Vr * Ho * Wo * Cout, Ho * Wo * Cout),
                              r_batch=(0, Ho * Wo * Cout), conv=conv,
                              conv_upsample=upsample, name=self._conv_tag(tag, KH, Ho, stride, in_dilation, KW), run=False)
        else:
            kw = {}
            if bias_img is not None:
                kw = dict(bias=bias_img[0], bias_row_div=Ho * Wo, bias_ld=bias_img[1])
            else:
                kw = dict(bias=b)
            d = gemm.gemm_raw(x, wt, y, B * Ho * Wo, Cout, K, (0, 1), (K, 1), Cout, residual=residual,
                              ldr=Cout if residual is not None else 0, act=act, conv=conv, conv_upsample=upsample,
                              name=self._conv_tag(tag, KH, Ho, stride, in_dilation, KW), run=False, **kw)
        self.p.add_gemm(d)
        return y

    def linear(self, x, wt, bias=None, act=None, residual=None, out_dtype=None, tag="linear"):
        K = x.shape[-1]
        M = x.numel() // K
        N = wt.shape[0]
        Nout = N // 2 if act == "geglu_pair" else N
        y = self.p.buf(*x.shape[:-1], Nout, dtype=out_dtype)
        d = gemm.gemm_raw(x, wt, y, M, N, K, (K, 1), (wt.stride(0), 1), Nout, bias=bias, residual=residual,
                          ldr=Nout if residual is not None else 0, act=act, name=tag, run=False)
        self.p.add_gemm(d)
        return y

    # -- norms / activations ------------------------------------------------------------------------------------
    def _gn_ws(self, B):
        """One partial-sum scratch per plan: GroupNorm calls are sequential on the plan's stream."""
        need = int(self.L.dwg_groupnorm_workspace_floats(B, self.groups))
        pool = self.p.__dict__.setdefault("_gn_workspace", {})      # one per branch: branches run concurrently
        ws = pool.get(self.p.branch)
        if ws is None or ws.numel() < need:
            ws = self.p.buf(need, dtype=torch.float32)
            pool[self.p.branch] = ws
        return ws

    def groupnorm(self, x, name, eps, silu, keep_stats=False):
        self.p.scope = name
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        y = self.p.buf(*x.shape)
        stats = self.p.buf(B, self.groups, 2, dtype=torch.float32)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_groupnorm_forward_dt, self.p.dt, B, HW, C, self.groups, pp(x), pp(self.w.f32(name + ".weight")),
                        pp(self.w.f32(name + ".bias")), eps, int(silu), pp(y), pp(stats), pp(self._gn_ws(B)))
        return (y, stats) if keep_stats else y

    def groupnorm_bwd(self, x, dy, stats, name, eps, silu, residual=None):
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        dx = self.p.buf(*x.shape)
        scratch = self.p.buf(B, self.groups, 2, dtype=torch.float32)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_groupnorm_backward_dt, self.p.dt, B, HW, C, self.groups, pp(x), pp(dy), pp(stats),
                        pp(self.w.f32(name + ".weight")), pp(self.w.f32(name + ".bias")), eps, int(silu), pp(dx), pp(scratch),
                        pp(self._gn_ws(B)), pp(residual) if residual is not None else None)
        return dx

    def layernorm(self, x, name):
        self.p.scope = name
        C = x.shape[-1]
        M = x.numel() // C
        y = self.p.buf(*x.shape)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_layernorm_forward_dt, self.p.dt, M, C, pp(x), pp(self.w.f32(name + ".weight")), pp(self.w.f32(name + ".bias")),
                        1e-5, pp(y))
        return y

    def geglu(self, x):
        F2 = x.shape[-1]
        M = x.numel() // F2
        y = self.p.buf(*x.shape[:-1], F2 // 2)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_geglu_forward_dt, self.p.dt, M, F2 // 2, pp(x), pp(y))
        return y

    def _pooled(self, tag, numel, dtype):
        """Scratch shared by the sequential ops of one branch (score / probability matrices of the unfused attention)."""
        pool = self.p.__dict__.setdefault("_scratch_pool", {})
        key = (tag, self.p.branch, dtype)
        t = pool.get(key)
        if t is None or t.numel() < numel:
            t = pool[key] = self.p.buf(numel, dtype=dtype)
        return t[:numel]

    def attention(self, q, k, v, heads):
        B, Nq, HD = q.shape
        Nk, d = k.shape[1], HD // heads
        o = self.p.buf(B, Nq, HD)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        if self.p.dtype_name == "f32":
            # full-precision plans: S = Q K^T (batched over image x head) -> row softmax -> O = P V on the exact-f32 MFMA GEMM
            S = self._pooled("attn_S", B * heads * Nq * Nk, torch.float32)
            P = self._pooled("attn_P", B * heads * Nq * Nk, self.p.dtype)
            self.p.add_gemm(gemm.gemm_raw(q, k, S, Nq, Nk, d, (q.stride(1), 1), (k.stride(1), 1), Nk, batch=(B, heads),
                                          a_batch=(q.stride(0), d), b_batch=(k.stride(0), d), c_batch=(heads * Nq * Nk, Nq * Nk),
                                          name="attn_qk", run=False))
            self.p.add_call(self.L.dwg_softmax_rows_forward_dt, self.p.dt, B * heads * Nq, Nk, float(d) ** -0.5, pp(S), Nk, pp(P), Nk)
            self.p.add_gemm(gemm.gemm_raw(P, v, o, Nq, d, Nk, (Nk, 1), (1, v.stride(1)), HD, batch=(B, heads),
                                          a_batch=(heads * Nq * Nk, Nq * Nk), b_batch=(v.stride(0), d), c_batch=(o.stride(0), d),
                                          name="attn_pv", run=False))
            return o
        # small query counts (self-attention of the 32x32 / 16x16 levels): the keys are split over workgroups and merged by a second launch
        # (include/dwg_nn.h dwg_attention_forward_ws) -- the workspace belongs to this call site
        need = int(self.L.dwg_attention_split_workspace_bytes(self.p.dt, B, heads, Nq, Nk, d))
        if need > 0:
            ws = self.p.buf((need + 3) // 4, dtype=torch.float32)
            self.p.add_call(self.L.dwg_attention_forward_ws, self.p.dt, B, heads, Nq, Nk, d, pp(q), q.stride(1), q.stride(0), pp(k), k.stride(1),
                            k.stride(0), pp(v), v.stride(1), v.stride(0), pp(o), o.stride(1), o.stride(0), float(d) ** -0.5, pp(ws), need)
            return o
        self.p.add_call(self.L.dwg_attention_forward_dt, self.p.dt, B, heads, Nq, Nk, d, pp(q), q.stride(1), q.stride(0), pp(k), k.stride(1),
                        k.stride(0), pp(v), v.stride(1), v.stride(0), pp(o), o.stride(1), o.stride(0), float(d) ** -0.5)
        return o

    def cat(self, a, b):
        Ca, Cb = a.shape[-1], b.shape[-1]
        y = self.p.buf(*a.shape[:-1], Ca + Cb)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        k2 = self.p.esize // 2                      # the concat kernel moves 16-byte pieces of 2-byte elements: fp32 channels count twice
        self.p.add_call(self.L.dwg_concat_channels, a.numel() // Ca, Ca * k2, Cb * k2, pp(a), pp(b), pp(y))
        return y

    def add(self, a, b):
        y = self.p.buf(*a.shape)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_add_dt, self.p.dt, a.numel(), pp(a), pp(b), pp(y))
        return y

    def conv_dgrad_s2(self, dy, name):
        """Input gradient of `F.pad(x, (0,1,0,1)); conv(3x3, stride 2)` (diffusers Downsample2D, padding 0): four stride-1
        convolutions of dy, one per output parity class, then one interleave pass."""
        B, Ho, Wo, _ = dy.shape
        subs = []
        for py in (0, 1):
            for px in (0, 1):
                wt = self.w.conv_dgrad_s2(name, py, px)
                subs.append(self.conv(dy, None, weight=wt, bias=False, stride=1, pad_tl=(wt.shape[1] - 1, wt.shape[2] - 1),
                                      out_hw=(Ho, Wo)))
        C = subs[0].shape[-1]
        out = self.p.buf(B, 2 * Ho, 2 * Wo, C)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_interleave2x2, B, Ho, Wo, C * (self.p.esize // 2), pp(subs[0]), pp(subs[1]), pp(subs[2]), pp(subs[3]), pp(out))
        return out

    def transpose(self, x):
        """[B, R, C] -> [B, C, R] (2-byte element and f32x plans): operands of the VAE attention made K-contiguous for the direct-to-LDS GEMM kernels."""
        B, R, C = x.shape
        y = self.p.buf(B, C, R)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_transpose_dt, self.p.dt, B, R, C, pp(x), x.stride(1), x.stride(0), pp(y), R, C * R)
        return y

    def cast_bf16(self, x):
        """fp32 accumulator buffer -> the plan's activation type (a no-op for the fp32 plans)."""
        if self.p.dtype_name == "f32":
            return x
        y = self.p.buf(*x.shape)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self.p.add_call(self.L.dwg_cast_f32_to_dt, self.p.dt, x.numel(), pp(x), pp(y))
        return y

    # -- blocks ----------------------------------------------------------------------------------------------------
    def resnet(self, x, pre, temb_bias, eps=1e-5):
        self.p.scope = pre
        C = x.shape[-1]
        n1 = self.groupnorm(x, pre + ".norm1", eps, True)
        h1 = self.conv(n1, pre + ".conv1", bias_img=temb_bias)
        n2 = self.groupnorm(h1, pre + ".norm2", eps, True)
        Cout = h1.shape[-1]
        sc = x if C == Cout else self.conv(x, pre + ".conv_shortcut", pad=0)
        return self.conv(n2, pre + ".conv2", residual=sc)

    def text_kv(self, text, t):
        """[k | v] of block `t`'s cross-attention.  The text k / v projections of ALL transformer blocks of this network read the same
        [B, 77, 768] input and nothing else: they run as ONE GEMM against the row-concatenated weights when the first block asks (16
        launches of ~12 us each in the UNet, 7 in the ControlNet -> 1 + 1); a block's k and v are column slices of its output."""
        batches = self.__dict__.setdefault("_text_kv", {})
        key = text.data_ptr()
        base = getattr(text, "_base", None)
        if key not in batches and base is not None and base.data_ptr() in batches and batches[base.data_ptr()] is not None \
                and base.dim() == text.dim() and base.shape[1:] == text.shape[1:] and text.is_contiguous():
            # a batch-row slice of a text whose projections exist already (the halves of a split decoder; a prelude that ran on the whole
            # batch): the same rows of that output
            r0 = (text.data_ptr() - base.data_ptr()) // (base.stride(0) * base.element_size())
            out, offs, _ = batches[base.data_ptr()]
            batches[key] = (out[r0:r0 + text.shape[0]], offs, text)
        if key not in batches:
            if os.environ.get("DWG_TEXT_KV_PER_BLOCK") == "1":           # experiment switch: the round-1 schedule
                batches[key] = None
            else:
                suffix = ".attn2.to_k.weight"
                blocks = sorted(n[:-len(suffix)] for n in self.w.sd if n.endswith(suffix) and self.w.sd[n].shape[1] == text.shape[-1])
                names = []
                for b_ in blocks:
                    names += [b_ + ".attn2.to_k", b_ + ".attn2.to_v"]
                out = self.linear(text, self.w.lin(*names), tag="attn_kv")
                offs, o = {}, 0
                for b_ in blocks:
                    c = int(self.w.sd[b_ + suffix].shape[0])
                    offs[b_] = (o, c); o += 2 * c
                batches[key] = (out, offs, text)
        ent = batches[key]
        if ent is None or t not in ent[1]:
            return self.linear(text, self.w.lin(t + ".attn2.to_k", t + ".attn2.to_v"), tag="attn_kv")
        o, c = ent[1][t]
        return ent[0][..., o:o + 2 * c]

    def transformer(self, x, pre, text, heads):
        self.p.scope = pre
        B, H, W, C = x.shape
        n = self.groupnorm(x, pre + ".norm", 1e-6, False)
        h = self.conv(n, pre + ".proj_in", pad=0).view(B, H * W, C)
        t = pre + ".transformer_blocks.0"
        l1 = self.layernorm(h, t + ".norm1")
        qkv = self.linear(l1, self.w.lin(t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"), tag="attn_qkv")
        a = self.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
        h = self.linear(a, self.w.lin(t + ".attn1.to_out.0"), bias=self.w.f32(t + ".attn1.to_out.0.bias"), residual=h, tag="attn_out")
        l2 = self.layernorm(h, t + ".norm2")
        q = self.linear(l2, self.w.lin(t + ".attn2.to_q"), tag="attn_q")
        kv = self.text_kv(text, t)
        a = self.attention(q, kv[..., :C], kv[..., C:], heads)
        h = self.linear(a, self.w.lin(t + ".attn2.to_out.0"), bias=self.w.f32(t + ".attn2.to_out.0.bias"), residual=h, tag="attn_out")
        l3 = self.layernorm(h, t + ".norm3")
        wg, bg = self.w.lin_geglu(t + ".ff.net.0.proj")
        g = self.linear(l3, wg, bias=bg, act="geglu_pair", tag="ff_in")       # GEGLU fused into the projection's epilogue
        h = self.linear(g, self.w.lin(t + ".ff.net.2"), bias=self.w.f32(t + ".ff.net.2.bias"), residual=h, tag="ff_out")
        return self.conv(h.view(B, H, W, C), pre + ".proj_out", pad=0, residual=x)


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of t * exp(-ln(1e4) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class _TimeEmbed:
    """time_embedding MLP + every resnet's time_emb_proj as ONE batched projection:
    temb_bias[r] = conv1.bias + time_emb_proj(silu(emb))  -> consumed by conv1's epilogue as a per-image channel bias."""

    def __init__(self, b: Builder, w: Weights, cfg: UNetConfig, B, resnet_names):
        self.tin = b.p.buf(B, cfg.block_out_channels[0])
        e1 = b.linear(self.tin, w.lin("time_embedding.linear_1"), bias=w.f32("time_embedding.linear_1.bias"), act="silu", tag="temb")
        emb = b.linear(e1, w.lin("time_embedding.linear_2"), bias=w.f32("time_embedding.linear_2.bias"), act="silu", tag="temb")
        names = [n + ".time_emb_proj" for n in resnet_names]
        wt = w.lin(*names)
        bias = w.bias_cat(*[n + ".time_emb_proj.bias" for n in resnet_names]) + w.bias_cat(*[n + ".conv1.bias" for n in resnet_names])
        b.p.keep.append(bias)
        self.all = b.linear(emb, wt, bias=bias, out_dtype=torch.float32, tag="temb_proj")     # [B, sum C]
        self.offsets, off = {}, 0
        for n in resnet_names:
            c = w.sd[n + ".conv1.bias"].numel()
            self.offsets[n] = (off, c); off += c
        self.ld = off

    def bias_for(self, name):
        off, c = self.offsets[name]
        return (self.all[:, off:off + c], self.ld)


def _encoder_resnet_names(cfg: UNetConfig):
    names = []
    for i in range(len(cfg.block_out_channels)):
        for j in range(cfg.layers_per_block):
            names.append("down_blocks.%d.resnets.%d" % (i, j))
    return names + ["mid_block.resnets.0", "mid_block.resnets.1"]


def _build_encoder(b: Builder, cfg: UNetConfig, x, temb: _TimeEmbed, text, hint=None):
    """conv_in (+ ControlNet hint) -> down blocks -> mid block.  Returns (skips, mid)."""
    h = b.conv(x, "conv_in", residual=hint, r_batch_bcast=hint is not None)
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            pre = "down_blocks.%d.resnets.%d" % (i, j)
            h = b.resnet(h, pre, temb.bias_for(pre))
            if cfg.attn_blocks[i]:
                h = b.transformer(h, "down_blocks.%d.attentions.%d" % (i, j), text, cfg.heads)
            skips.append(h)
        if i != nb - 1:
            h = b.conv(h, "down_blocks.%d.downsamplers.0.conv" % i, stride=2)
            skips.append(h)
    h = b.resnet(h, "mid_block.resnets.0", temb.bias_for("mid_block.resnets.0"))
    h = b.transformer(h, "mid_block.attentions.0", text, cfg.heads)
    h = b.resnet(h, "mid_block.resnets.1", temb.bias_for("mid_block.resnets.1"))
    return skips, h


class DenoiserPlan:
    """ControlNet-conditioned UNet forward for a CFG batch: eps = UNet(x, t, text, ControlNet(x, t, text, cond)).

    Inputs (static buffers, overwritten before run()):  latents [B,h,w,8] (4 used), t_emb [B,320] x2, text [B,77,768],
    cond [V,8h,8w,8] (3 used; V = `views`, B = 2 V ordered [negative of view 0..V-1 | text of view 0..V-1]: the condition image of a
    view is identical for its CFG entries -- controlnet.py:60-72 repeats it -- so its embedding is computed once per view and
    broadcast).  Output: eps [B,h,w,4] fp32.  `weights`: (UNet, ControlNet) `Weights` of another plan of the same dtype to share."""

    def __init__(self, cfg: UNetConfig, unet_sd, cn_sd, device, batch=2, latent_hw=64, text_len=77, dtype="bf16", views=1, weights=None):
        self.cfg, self.device, self.B, self.hw, self.views = cfg, device, batch, latent_hw, int(views)
        self._temb_table = None
        self.temb_rows = 1001                 # timesteps the embedding table covers: 0 .. the scheduler's num_train_timesteps (guidance sets it)
        assert batch % self.views == 0, (batch, views)
        self.plan = Plan(device, dtype)
        p = self.plan
        # The PRELUDE (round 6): everything of the call that does not depend on the latents -- both time embeddings and their per-resnet
        # projections, the ControlNet's hint embedding of the condition image (eight convolutions down from 8h x 8w), the text k / v projections of
        # every cross-attention of both networks -- is a plan of its own.  `run()` replays it in front of the main plan; a caller that knows the
        # timestep, the text and the condition image before the latents exist (guidance.__call__: they exist before the VAE encoder runs) starts
        # it on a side stream with `prefetch()` and it runs UNDER the VAE encoder, whose launches are matrix-bound.
        self.pre = Plan(device, dtype)
        self._pre_event, self._pre_ready, self._pre_stream = None, False, None
        wu, wc = weights if weights is not None else (Weights(unet_sd, device, dtype), Weights(cn_sd, device, dtype))
        self.weights = (wu, wc)     # kernel-layout weight tensors must outlive the plan that points at them
        bu, bc = Builder(p, wu, cfg.groups, "unet"), Builder(p, wc, cfg.groups, "cnet")
        pu, pc = Builder(self.pre, wu, cfg.groups, "unet"), Builder(self.pre, wc, cfg.groups, "cnet")
        B, hw = batch, latent_hw
        self.latents = p.buf(B, hw, hw, _pad8(cfg.in_channels), zero=True)
        self.text = p.buf(B, text_len, cfg.cross_dim)
        self.cond_scale = 2 ** (len(cfg.cond_channels) - 1)      # one stride-2 convolution per embedding level: 8 for SD-1.5
        self.cond = p.buf(self.views, hw * self.cond_scale, hw * self.cond_scale, _pad8(cfg.cond_in_channels), zero=True)
        self._lat_stage, self._cond_stage = p.stage_like(self.latents), p.stage_like(self.cond)
        # ---- UNet encoder (does not depend on the ControlNet)
        up_names = []
        rev_attn = list(reversed(cfg.attn_blocks))
        for i in range(len(cfg.block_out_channels)):
            for j in range(cfg.layers_per_block + 1):
                up_names.append("up_blocks.%d.resnets.%d" % (i, j))
        # ---- ControlNet body on branch 1: hint embedding (batch 1) + its own encoder.  It only shares INPUTS with the UNet
        # encoder, and below 64x64 neither fills the chip alone, so the two run as parallel paths (side stream / graph branch).
        import os
        par = os.environ.get("DWG_SERIAL_DENOISER") != "1"
        # DWG_DENOISER_SPLIT: 0 (default) = every chain on the whole CFG batch; 1 = the DECODER split by batch half -- it is ONE chain, and its
        # memory-bound layers (GroupNorm, LayerNorm, split-K reduces) otherwise have nothing to run under: the halves are two independent
        # chains (side stream / graph branch) with half the rows per launch; 2 = the encoders split by half as well (four chains).
        # Measured (DESIGN.md "Measured (round 6)"): 1 is 0.9 % faster per step (25.58 -> 25.35 ms) at 18 % more launches, each of them a less
        # efficient half-batch launch (kernel time summed over the step 26.8 -> 28.9 ms) -- wall time is bought with chip time, so it stays opt-in
        split = int(os.environ.get("DWG_DENOISER_SPLIT", "0")) if (par and B % 2 == 0) else 0
        if split >= 2 and self.views != 1:
            split = 1                                   # (the four-chain form shares ONE view's hint embedding between the halves)
        nb = len(cfg.block_out_channels)

        class _Rows:
            """A _TimeEmbed seen through a slice of the batch rows."""
            def __init__(self, temb, rows):
                self.temb, self.rows = temb, rows

            def bias_for(self, name):
                tb, ld = self.temb.bias_for(name)
                return (tb[self.rows], ld)

        def decoder(h, skips, rows, first=0, last=None):
            """up blocks first .. last - 1 on the batch rows `rows` (+ the output convolution when the last block is included)"""
            last = nb if last is None else last
            skips = list(skips)
            tu = _Rows(self.temb_u, rows)
            for i in range(first, last):
                for j in range(cfg.layers_per_block + 1):
                    pre = "up_blocks.%d.resnets.%d" % (i, j)
                    h = bu.resnet(bu.cat(h, skips.pop()), pre, tu.bias_for(pre))
                    if rev_attn[i]:
                        h = bu.transformer(h, "up_blocks.%d.attentions.%d" % (i, j), self.text[rows], cfg.heads)
                if i != nb - 1:
                    h = bu.conv(h, "up_blocks.%d.upsamplers.0.conv" % i, upsample=2)
            if last < nb:
                return h, skips
            n = bu.groupnorm(h, "conv_norm_out", 1e-5, True)
            return bu.conv(n, "conv_out", out_dtype=torch.float32)

        def zero_convs(cskips, cmid, skips, mid):
            """the ControlNet's zero convolutions, whose epilogue adds the UNet skip they feed"""
            return ([bc.conv(cs, "controlnet_down_blocks.%d" % k, pad=0, residual=skips[k]) for k, cs in enumerate(cskips)],
                    bc.conv(cmid, "controlnet_mid_block", pad=0, residual=mid))

        def hint_embedding(b_):
            e = "controlnet_cond_embedding"
            hnt = b_.conv(self.cond, e + ".conv_in", act="silu")
            for k in range(2 * (len(cfg.cond_channels) - 1)):
                hnt = b_.conv(hnt, "%s.blocks.%d" % (e, k), stride=2 if k % 2 == 1 else 1, act="silu")
            return b_.conv(hnt, e + ".conv_out")
        self.eps_halves = None
        whole = slice(0, B)
        # ---- the prelude plan: two independent chains (ControlNet side / UNet side)
        suffix = ".attn2.to_k"
        first_block = lambda w_: sorted(n[:-len(suffix + ".weight")] for n in w_.sd if n.endswith(suffix + ".weight"))[0]     # noqa: E731
        self.pre.fork(1)
        with self.pre.on_branch(1):
            self.temb_c = _TimeEmbed(pc, wc, cfg, B, _encoder_resnet_names(cfg))
            hnt = hint_embedding(pc)
            pc.text_kv(self.text, first_block(wc))
        self.temb_u = _TimeEmbed(pu, wu, cfg, B, _encoder_resnet_names(cfg) + up_names)
        pu.text_kv(self.text, first_block(wu))
        self.pre.join(1)
        bu.__dict__["_text_kv"] = pu.__dict__["_text_kv"]       # the main plan's cross-attentions read the prelude's projections
        bc.__dict__["_text_kv"] = pc.__dict__["_text_kv"]
        if split < 2:
            if par:
                p.fork(1)
            with p.on_branch(1 if par else 0):
                cskips, cmid = _build_encoder(bc, cfg, self.latents, self.temb_c, self.text, hint=hnt)
            skips, mid = _build_encoder(bu, cfg, self.latents, self.temb_u, self.text)
            if par:
                p.join(1)
            skips, h = zero_convs(cskips, cmid, skips, mid)
            if split == 1:
                # the low-resolution up blocks stay on the whole batch: their launches are bound by the WEIGHTS they stream (59 - 118 MB per
                # 3 x 3 convolution of the 1280-channel levels), which two half-batch chains would read twice
                first = min(nb - 1, max(0, int(os.environ.get("DWG_DECODER_SPLIT_FROM", "2"))))
                if first > 0:
                    h, skips = decoder(h, skips, whole, 0, first)
                lo, hi = slice(0, B // 2), slice(B // 2, B)
                p.fork(1)
                with p.on_branch(1):
                    e1 = decoder(h[hi], [t[hi] for t in skips], hi, first)
                e0 = decoder(h[lo], [t[lo] for t in skips], lo, first)
                p.join(1)
                self.eps_halves, self.eps = (e0, e1), None
            else:
                self.eps = decoder(h, skips, whole)
        else:
            # four encoder chains (UNet / ControlNet x batch half), then one zero-convolution + decoder chain per half
            lo, hi = slice(0, B // 2), slice(B // 2, B)
            for br in (1, 2, 3):
                p.fork(br)
            enc = {}
            for br, (bld, w_, temb, rows, hint) in {0: (bu, wu, self.temb_u, lo, None), 1: (bc, wc, self.temb_c, lo, hnt),
                                                    2: (bu, wu, self.temb_u, hi, None), 3: (bc, wc, self.temb_c, hi, hnt)}.items():
                with p.on_branch(br):
                    enc[br] = _build_encoder(bld, cfg, self.latents[rows], _Rows(temb, rows), self.text[rows], hint=hint)
            p.join(1)                                   # the lower half's chain (branch 0) needs its ControlNet encoder
            p.ops.append(("fork", 3, 2))                # ... and the upper half's (branch 2) its own
            outs = {}
            for br, cn, rows in ((2, 3, hi), (0, 1, lo)):
                with p.on_branch(br):
                    sk, hh = zero_convs(enc[cn][0], enc[cn][1], enc[br][0], enc[br][1])
                    outs[br] = decoder(hh, sk, rows)
            p.join(2)
            self.eps_halves, self.eps = (outs[0], outs[2]), None

    def _store_prelude_inputs(self, t, text, cond_nchw):
        p = self.plan
        t = t.reshape(-1)
        if t.numel() not in (1, self.B):
            assert self.B % t.numel() == 0, (self.B, t.numel())
            t = t.repeat(self.B // t.numel())                 # [t_0..t_{V-1} | t_0..t_{V-1}]: the batch order of the CFG halves
        if t.dtype == torch.long and t.is_cuda:
            # integer timesteps: rows of a table computed once with the same statements (eight element-wise launches per call otherwise)
            # (rows 0 .. temb_rows - 1 = every timestep of the owner's scheduler, which also indexes its alphas_cumprod with t: a timestep
            # outside the scheduler's range is an error there as here)
            if self._temb_table is None or self._temb_table.shape[0] != self.temb_rows:
                self._temb_table = timestep_embedding(torch.arange(self.temb_rows, device=t.device), self.cfg.block_out_channels[0])
            te = self._temb_table.index_select(0, t.expand(self.B))
        else:
            te = timestep_embedding(t.expand(self.B), self.cfg.block_out_channels[0])
        p.store(self.temb_u.tin, te); p.store(self.temb_c.tin, te)
        p.store(self.text, text)
        if cond_nchw is not None:
            p.store(self.cond, cond_nchw.permute(0, 2, 3, 1), self._cond_stage)

    def prefetch(self, t, text, cond_nchw=None, stream=None):
        """Starts the prelude (time embeddings, hint embedding, text k / v) for the NEXT `run()` now -- on `stream` (a side stream: it then
        runs beside whatever the caller's stream does until `run()`, which waits for it) or on the current stream.  The following
        `set_inputs` only needs the latents; passing t / text / cond there again is allowed and ignored."""
        dev = self.device
        if stream is None or torch.device(dev).type != "cuda":
            self._store_prelude_inputs(t, text, cond_nchw)
            self.pre.run()
            self._pre_ready, self._pre_event = True, None
            return
        main = torch.cuda.current_stream(dev)
        stream.wait_stream(main)                               # the inputs' producers
        with torch.cuda.stream(stream):
            self._store_prelude_inputs(t, text, cond_nchw)
            self.pre.run()
            ev = torch.cuda.Event()
            ev.record(stream)
        for x in (t, text, cond_nchw):
            if torch.is_tensor(x) and x.is_cuda:
                x.record_stream(stream)
        self._pre_ready, self._pre_event = True, ev

    def set_inputs(self, latents_nchw, t=None, text=None, cond_nchw=None):
        """latents [B,4,h,w] fp32, t scalar / [V] (one per view, repeated over the CFG entries) / [B], text [B,77,768],
        cond [V,3,8h,8w] in [0,1] (optional).  After a `prefetch()` only the latents are taken."""
        self.plan.store(self.latents, latents_nchw.permute(0, 2, 3, 1), self._lat_stage)
        if not self._pre_ready:
            self._store_prelude_inputs(t, text, cond_nchw)

    def run(self):
        if self._pre_ready:
            if self._pre_event is not None:
                torch.cuda.current_stream(self.device).wait_event(self._pre_event)
        else:
            self.pre.run()
        self._pre_ready, self._pre_event = False, None
        self.plan.run()
        if self.eps_halves is not None:
            return torch.cat(self.eps_halves, dim=0).permute(0, 3, 1, 2)
        return self.eps.permute(0, 3, 1, 2)   # [B,4,h,w] view (fp32)


class VAEEncoderPlan:
    """AutoencoderKL.encoder + quant_conv forward with the input-gradient backward (no weight gradients: the VAE is frozen
    but sits INSIDE the autograd graph of SDS -- basic.py:368-372).  forward: image [1,3,H,W] in [0,1] -> moments
    [1,8,H/8,W/8] (2x-1 normalisation of VaeImageProcessor fused into the input conversion).  backward: d moments -> d image."""

    def __init__(self, cfg: VAEConfig, sd, device, image_hw=512, dtype="bf16", batch=1, weights=None):
        self.cfg, self.device, self.hw, self.B = cfg, device, image_hw, int(batch)
        self.fwd, self.bwd = Plan(device, dtype), Plan(device, dtype)
        w = weights if weights is not None else Weights(sd, device, dtype)
        self.weights = w            # kernel-layout weight tensors must outlive the plans that point at them
        f, r = Builder(self.fwd, w, cfg.groups, "vaef"), Builder(self.bwd, w, cfg.groups, "vaeb")
        self.x = self.fwd.buf(self.B, image_hw, image_hw, 8, zero=True)
        self._x_stage = self.fwd.stage_like(self.x)
        boc = cfg.block_out_channels
        tape = []      # closures that extend the backward plan, replayed in reverse order

        def conv_f(xin, name, **kw):
            y = f.conv(xin, name, **kw)
            return y

        def resnet_f(xin, pre):
            C = xin.shape[-1]
            n1, s1 = f.groupnorm(xin, pre + ".norm1", 1e-6, True, keep_stats=True)
            h1 = f.conv(n1, pre + ".conv1")
            n2, s2 = f.groupnorm(h1, pre + ".norm2", 1e-6, True, keep_stats=True)
            Cout = h1.shape[-1]
            sc = xin if C == Cout else f.conv(xin, pre + ".conv_shortcut", pad=0)
            out = f.conv(n2, pre + ".conv2", residual=sc)

            def back(dout):
                dn2 = r.conv(dout, None, weight=w.conv(pre + ".conv2", True), bias=False)
                dh1 = r.groupnorm_bwd(h1, dn2, s2, pre + ".norm2", 1e-6, True)
                dn1 = r.conv(dh1, None, weight=w.conv(pre + ".conv1", True), bias=False)
                if C == Cout:                       # identity skip: its gradient is added inside the GroupNorm backward
                    return r.groupnorm_bwd(xin, dn1, s1, pre + ".norm1", 1e-6, True, residual=dout)
                dx = r.groupnorm_bwd(xin, dn1, s1, pre + ".norm1", 1e-6, True)
                return r.conv(dout, None, weight=w.conv(pre + ".conv_shortcut", True), pad=0, bias=False, residual=dx)
            tape.append(back)
            return out

        h = f.conv(self.x, "encoder.conv_in")
        tape.append(lambda d: r.conv(d, None, weight=w.conv("encoder.conv_in", True), bias=False))
        for i, cout in enumerate(boc):
            for j in range(cfg.layers_per_block):
                h = resnet_f(h, "encoder.down_blocks.%d.resnets.%d" % (i, j))
            if i != len(boc) - 1:
                name = "encoder.down_blocks.%d.downsamplers.0.conv" % i
                Hin = h.shape[1]
                h = f.conv(h, name, stride=2, pad=0, out_hw=(Hin // 2, Hin // 2))      # F.pad(0,1,0,1) + conv stride 2
                if os.environ.get("DWG_VAE_DILATED_DGRAD") == "1":     # zero-dilated input, all 9 taps: 4x the multiply-adds
                    tape.append(lambda d, name=name, Hin=Hin: r.conv(d, None, weight=w.conv(name, True), bias=False, stride=1,
                                                                     pad_tl=(2, 2), out_hw=(Hin, Hin), in_dilation=2))
                else:
                    tape.append(lambda d, name=name: r.conv_dgrad_s2(d, name))
        h = resnet_f(h, "encoder.mid_block.resnets.0")
        h = self._attention(f, r, w, h, "encoder.mid_block.attentions.0", tape)
        h = resnet_f(h, "encoder.mid_block.resnets.1")
        n, st = f.groupnorm(h, "encoder.conv_norm_out", 1e-6, True, keep_stats=True)
        tape.append(lambda d, h=h, st=st: r.groupnorm_bwd(h, d, st, "encoder.conv_norm_out", 1e-6, True))
        c8 = f.conv(n, "encoder.conv_out")
        tape.append(lambda d: r.conv(d, None, weight=w.conv("encoder.conv_out", True), bias=False))
        self.moments = f.conv(c8, "quant_conv", pad=0, out_dtype=torch.float32)
        tape.append(lambda d: r.conv(d, None, weight=w.conv("quant_conv", True), pad=0, bias=False))
        # backward plan
        self.dmoments = self.bwd.buf(*self.moments.shape, zero=True)
        d = self.dmoments
        for back in reversed(tape):
            d = back(d)
        self.dx = d     # [1,H,W,8] bf16

    @staticmethod
    def _attention(f: Builder, r: Builder, w: Weights, x, pre, tape):
        """Single-head self-attention of the VAE mid block (d = C = 512, N = 4096): QK^T / softmax / PV as strided GEMMs +
        row softmax, because its backward needs P."""
        B, H, W, C = x.shape
        N = H * W
        scale = float(C) ** -0.5
        n, st = f.groupnorm(x, pre + ".group_norm", 1e-6, False, keep_stats=True)
        nt = n.view(B, N, C)
        q = f.linear(nt, w.lin(pre + ".to_q"), bias=w.f32(pre + ".to_q.bias"), tag="vae_qkv")
        k = f.linear(nt, w.lin(pre + ".to_k"), bias=w.f32(pre + ".to_k.bias"), tag="vae_qkv")
        v = f.linear(nt, w.lin(pre + ".to_v"), bias=w.f32(pre + ".to_v.bias"), tag="vae_qkv")
        # per image (batch1 = B): scores, probabilities and the value product are batched GEMMs over the images of the plan
        S = f.p.buf(B, N, N, dtype=torch.float32)
        f.p.add_gemm(gemm.gemm_raw(q, k, S, N, N, C, (C, 1), (C, 1), N, batch=(B, 1), a_batch=(N * C, 0), b_batch=(N * C, 0),
                                   c_batch=(N * N, 0), name="vae_qk", run=False))
        P = f.p.buf(B, N, N)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        f.p.add_call(f.L.dwg_softmax_rows_forward_dt, f.p.dt, B * N, N, scale, pp(S), N, pp(P), N)
        o = f.p.buf(B, N, C)
        # 2-byte plans: every product below runs with BOTH operands K-contiguous (the direct-to-LDS MFMA kernels: 3-4x the rate of the
        # register-staged strided loader) -- v, do, k, q, P and dS are transposed once each (dwg_transpose_2byte: 4 / 32 MB, ~3 / ~15 us)
        # and the backward uses transposed copies of the four projection weights.  The fp32 plans keep the strided products.
        kc = f.p.esize == 2 or f.p.is_x          # f32x operands exist ONLY K-contiguous (8-channel groups along the contraction)
        if kc:
            vT = f.transpose(v)
            f.p.add_gemm(gemm.gemm_raw(P, vT, o, N, C, N, (N, 1), (N, 1), C, batch=(B, 1), a_batch=(N * N, 0), b_batch=(N * C, 0),
                                       c_batch=(N * C, 0), name="vae_pv", run=False))
        else:
            f.p.add_gemm(gemm.gemm_raw(P, v, o, N, C, N, (N, 1), (1, C), C, batch=(B, 1), a_batch=(N * N, 0), b_batch=(N * C, 0),
                                       c_batch=(N * C, 0), name="vae_pv", run=False))
        out = f.linear(o, w.lin(pre + ".to_out.0"), bias=w.f32(pre + ".to_out.0.bias"), residual=x.view(B, N, C), tag="vae_out")
        out = out.view(B, H, W, C)

        def back(dout):
            dt = dout.view(B, N, C)
            wo = w.lin(pre + ".to_out.0")
            M = B * N
            bk = dict(batch=(B, 1))
            do = r.p.buf(B, N, C)      # d o = dout @ Wo
            if kc:
                r.p.add_gemm(gemm.gemm_raw(dt, w.lin_t(pre + ".to_out.0"), do, M, C, C, (C, 1), (C, 1), C, name="vae_bwd", run=False))
            else:
                r.p.add_gemm(gemm.gemm_raw(dt, wo, do, M, C, C, (C, 1), (1, C), C, name="vae_bwd", run=False))
            dP = r.p.buf(B, N, N, dtype=torch.float32)   # dP = do v^T
            r.p.add_gemm(gemm.gemm_raw(do, v, dP, N, N, C, (C, 1), (C, 1), N, a_batch=(N * C, 0), b_batch=(N * C, 0), c_batch=(N * N, 0),
                                       name="vae_bwd_dp", run=False, **bk))
            dv = r.p.buf(B, N, C)      # dv = P^T do
            if kc:
                PT, doT = r.transpose(P), r.transpose(do)
                r.p.add_gemm(gemm.gemm_raw(PT, doT, dv, N, C, N, (N, 1), (N, 1), C, a_batch=(N * N, 0), b_batch=(N * C, 0), c_batch=(N * C, 0),
                                           name="vae_bwd_dv", run=False, **bk))
            else:
                r.p.add_gemm(gemm.gemm_raw(P, do, dv, N, C, N, (1, N), (1, C), C, a_batch=(N * N, 0), b_batch=(N * C, 0), c_batch=(N * C, 0),
                                           name="vae_bwd_dv", run=False, **bk))
            dS = r.p.buf(B, N, N)
            r.p.add_call(r.L.dwg_softmax_rows_backward_dt, r.p.dt, B * N, N, scale, pp(P), N, pp(dP), N, pp(dS), N)
            dq = r.p.buf(B, N, C)      # dq = dS k
            dk = r.p.buf(B, N, C)      # dk = dS^T q
            if kc:
                kT, qT, dST = r.transpose(k), r.transpose(q), r.transpose(dS)
                r.p.add_gemm(gemm.gemm_raw(dS, kT, dq, N, C, N, (N, 1), (N, 1), C, a_batch=(N * N, 0), b_batch=(N * C, 0), c_batch=(N * C, 0),
                                           name="vae_bwd_dq", run=False, **bk))
                r.p.add_gemm(gemm.gemm_raw(dST, qT, dk, N, C, N, (N, 1), (N, 1), C, a_batch=(N * N, 0), b_batch=(N * C, 0), c_batch=(N * C, 0),
                                           name="vae_bwd_dk", run=False, **bk))
            else:
                r.p.add_gemm(gemm.gemm_raw(dS, k, dq, N, C, N, (N, 1), (1, C), C, a_batch=(N * N, 0), b_batch=(N * C, 0), c_batch=(N * C, 0),
                                           name="vae_bwd_dq", run=False, **bk))
                r.p.add_gemm(gemm.gemm_raw(dS, q, dk, N, C, N, (1, N), (1, C), C, a_batch=(N * N, 0), b_batch=(N * C, 0), c_batch=(N * C, 0),
                                           name="vae_bwd_dk", run=False, **bk))
            dn = r.p.buf(B, N, C, dtype=torch.float32)   # dn = dq Wq + dk Wk + dv Wv (fp32 accumulate across the three products)
            for i, (g_, wn) in enumerate(((dq, ".to_q"), (dk, ".to_k"), (dv, ".to_v"))):
                if kc:
                    r.p.add_gemm(gemm.gemm_raw(g_, w.lin_t(pre + wn), dn, M, C, C, (C, 1), (C, 1), C, accumulate=i > 0, name="vae_bwd_dn", run=False))
                else:
                    r.p.add_gemm(gemm.gemm_raw(g_, w.lin(pre + wn), dn, M, C, C, (C, 1), (1, C), C, accumulate=i > 0, name="vae_bwd_dn", run=False))
            dnb = r.cast_bf16(dn).view(B, H, W, C)
            dx = r.groupnorm_bwd(x, dnb, st, pre + ".group_norm", 1e-6, False)
            return r.add(dx, dout)
        tape.append(back)
        return out

    def encode(self, image_nchw):
        """image [B,3,H,W] fp32 in [0,1] -> moments [B,8,h,w] fp32 (NCHW view)."""
        if self._fused_io(image_nchw):
            img = image_nchw.contiguous()
            _lib.check(self.fwd._lib.dwg_vae_image_pack(self.B, self.hw, self.hw, _lib.ptr(img), _lib.ptr(self.x), self._st()), "dwg_vae_image_pack")
        else:
            self.fwd.store(self.x, (image_nchw * 2.0 - 1.0).permute(0, 2, 3, 1), self._x_stage)
        self.fwd.run()
        return self.moments.permute(0, 3, 1, 2)

    def _st(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _fused_io(self, t):
        """f32x plans: the boundary conversions (2 v - 1 + NHWC + pack; max |g| + power-of-two pre-scale + NHWC + pack; unpack + NCHW + scale
        back) are one HIP launch each instead of 4 / ~14 / 3 element-wise torch launches (DWG_VAE_FUSED_IO=0: the torch statements)."""
        image = t.dim() == 4 and t.shape[1] == 3 and tuple(t.shape[-2:]) == (self.hw, self.hw)
        grad = t.dim() == 4 and t.shape[1] == 8 and tuple(t.shape[-2:]) == tuple(self.moments.shape[1:3])
        return (self.fwd.is_x and t.is_cuda and t.dtype == torch.float32 and (image or grad) and t.shape[0] == self.B
                and os.environ.get("DWG_VAE_FUSED_IO", "1") != "0")

    # the backward pass is LINEAR in the incoming gradient: it is run on 2^k x the gradient, k chosen on the device so that max |g| lands in
    # [GRAD_TARGET / 2, GRAD_TARGET], and the result is scaled back by 2^-k -- both exact.  Without it the 16-bit storage types see the
    # gradient at whatever scale the loss has: an SDS gradient of 1e-4 sits at the bottom of fp16's normal range and its f32x lo halves go
    # subnormal (measured round 4: VAE image gradient 2.0e-6 off the fp32 oracle at scale 1, 1.2e-4 at scale 1e-4; fp32 has no such floor:
    # /root/reference/configs/__init__.py:236,241).  64 leaves three decades of head-room below 65504 for growth inside the network.
    GRAD_TARGET = 64.0

    def backward(self, dmoments_nchw):
        """d loss / d moments [B,8,h,w] -> d loss / d image [B,3,H,W] fp32."""
        g = dmoments_nchw
        if self._fused_io(g):
            g = g.detach().contiguous()
            prescale = os.environ.get("DWG_VAE_GRAD_PRESCALE", "1") != "0"
            inv = torch.empty(1, device=self.device, dtype=torch.float32)
            h, w_ = int(self.moments.shape[1]), int(self.moments.shape[2])
            _lib.check(self.bwd._lib.dwg_vae_grad_prescale_pack(self.B, h * w_, _lib.ptr(g), float(self.GRAD_TARGET if prescale else 0.0),
                                                                _lib.ptr(self.dmoments), _lib.ptr(inv), self._st()), "dwg_vae_grad_prescale_pack")
            self.bwd.run()
            out = torch.empty(self.B, 3, self.hw, self.hw, device=self.device, dtype=torch.float32)
            _lib.check(self.bwd._lib.dwg_vae_dx_unpack(self.B, self.hw, self.hw, _lib.ptr(self.dx), _lib.ptr(inv), _lib.ptr(out), self._st()),
                       "dwg_vae_dx_unpack")
            return out
        inv = None
        if self.bwd.dtype_name in ("f32x", "f16") and os.environ.get("DWG_VAE_GRAD_PRESCALE", "1") != "0":
            amax = g.detach().abs().amax()
            k = torch.floor(torch.log2(self.GRAD_TARGET / amax.clamp_min(1e-30))).clamp(-60.0, 100.0)
            k = torch.where(amax > 0, k, torch.zeros_like(k))
            g = g * torch.exp2(k)                          # exact: a power of two
            inv = torch.exp2(1.0 - k)                      # 2 x 2^-k: the scale back and the d(2 x - 1)/dx of the input normalisation
        self.bwd.store(self.dmoments, g.permute(0, 2, 3, 1))
        self.bwd.run()
        dx = self.bwd.load(self.dx)[..., :3].permute(0, 3, 1, 2)
        return dx * inv if inv is not None else dx * 2.0
