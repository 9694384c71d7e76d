"""The CLIP image encoder in front of the loop, on the MI355X path.

The reference encodes the conditioning image once per clip with ``self.image_encoder(image).image_embeds``
(/root/reference/pipeline/pipeline_stable_video_diffusion_trans.py:164-203; ``image_encoder`` is transformers'
``CLIPVisionModelWithProjection`` [EXT], for SVD the ViT-H/14 of ``image_encoder/``: 32 layers x 1280 channels, 16 heads of 80,
257 tokens, projection to 1024).  This module keeps that class's name, its call (`model(pixel_values).image_embeds`) and its
parameter names - so ``image_encoder/model[.fp16].safetensors`` loads as it is - and runs the forward on the UNet's kernels:

* patch embedding (14x14 stride-14 conv, no bias) = ONE GEMM over the unfolded patches (K = 588 zero-padded to 640) whose
  epilogue adds the position embedding (row-indexed bias) and writes behind the image's class row;
* ``pre_layrnorm`` [sic], then per layer: LayerNorm affine folded into the fused q|k|v projection, dense short-sequence
  attention for head_dim 80 (``lkgd_attn_dense``: the flash kernel is head_dim 64 only), out-projection / fc2 with the residual
  in the epilogue, exact-erf GELU through the GEGLU epilogue with a constant-one "hidden" half (``hidden_act = "gelu"``, ViT-H)
  or ``quick_gelu`` as ``silu(1.702 x) / 1.702`` with both constants folded into fc1 / fc2 (OpenAI ViT-L);
* ``post_layernorm`` of the class token and ``visual_projection`` (no bias).

Runs once per clip: ~0.17 TFLOP per image.  Parity: tests/test_clip_gpu.py against transformers' own fp32 forward of the same
random-init weights (transformers is third-party and not part of the reference tree; the test skips where it is absent).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import LkgdHipError
from .loading import load_config
from .packing import pack_geglu, pack_linear


@dataclass
class CLIPVisionConfig:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    num_channels: int = 3
    image_size: int = 224
    patch_size: int = 14
    projection_dim: int = 1024
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5


def _f32(p):
    return p.detach().to(torch.float32).contiguous()


class CLIPImageProcessor:
    """the one thing the pipeline asks of transformers' ``CLIPImageProcessor`` [EXT] (reference :176-183: ``do_normalize`` only,
    resize / crop / rescale off): ``(image - image_mean) / image_std`` per channel, from ``preprocessor_config.json``"""

    OPENAI_MEAN, OPENAI_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, image_mean=None, image_std=None, **_kw):
        self.image_mean = tuple(image_mean) if image_mean is not None else self.OPENAI_MEAN
        self.image_std = tuple(image_std) if image_std is not None else self.OPENAI_STD

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **_kw):
        raw = load_config(path, subfolder, name="preprocessor_config.json")
        return cls(image_mean=raw.get("image_mean"), image_std=raw.get("image_std"))

    def __call__(self, images, do_normalize=True, do_center_crop=False, do_resize=False, do_rescale=False,
                 return_tensors="pt", **_kw):
        if do_center_crop or do_resize or do_rescale:
            raise LkgdHipError("lkgd_amd CLIPImageProcessor only normalises (the pipeline resizes with resize_with_antialiasing)")
        x = images if isinstance(images, torch.Tensor) else torch.as_tensor(images)
        x = x.float()
        if x.dim() == 3:
            x = x[None]
        if do_normalize:
            mean = torch.tensor(self.image_mean, dtype=torch.float32, device=x.device)[None, :, None, None]
            std = torch.tensor(self.image_std, dtype=torch.float32, device=x.device)[None, :, None, None]
            x = (x - mean) / std
        return SimpleNamespace(pixel_values=x)


class CLIPVisionEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        n = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.class_embedding = nn.Parameter(torch.zeros(cfg.hidden_size))
        self.patch_embedding = nn.Conv2d(cfg.num_channels, cfg.hidden_size, cfg.patch_size, stride=cfg.patch_size, bias=False)
        self.position_embedding = nn.Embedding(n, cfg.hidden_size)


class CLIPAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)


class CLIPMLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.hidden_size, cfg.intermediate_size)
        self.fc2 = nn.Linear(cfg.intermediate_size, cfg.hidden_size)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = CLIPAttention(cfg)
        self.layer_norm1 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = CLIPMLP(cfg)
        self.layer_norm2 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def pack(self, act: str):
        a, g1, b1 = self.self_attn, self.layer_norm1.weight.detach().float(), self.layer_norm1.bias.detach().float()
        wqkv = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0).detach().float()
        bqkv = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], dim=0).detach().float() + wqkv @ b1
        g2, b2 = self.layer_norm2.weight.detach().float(), self.layer_norm2.bias.detach().float()
        w1 = self.mlp.fc1.weight.detach().float()
        bf1 = self.mlp.fc1.bias.detach().float() + w1 @ b2
        w1 = w1 * g2[None, :]
        pk = SimpleNamespace(wqkv=pack_linear(wqkv * g1[None, :]), bqkv=bqkv.contiguous(), wo=pack_linear(a.out_proj.weight),
                             bo=_f32(a.out_proj.bias), b2=_f32(self.mlp.fc2.bias), act=act)
        if act == "gelu":
            # GELU(fc1 x) as GEGLU with a constant-one hidden half: hidden = 0 * x + 1, gate = fc1 x
            wg = torch.cat([torch.zeros_like(w1), w1], dim=0)
            bg = torch.cat([torch.ones_like(bf1), bf1], dim=0)
            pk.w1, pk.b1, pk.half = pack_geglu(wg, bg)
            pk.w2 = pack_linear(self.mlp.fc2.weight)
        else:
            # quick_gelu(x) = x * sigmoid(1.702 x) = silu(1.702 x) / 1.702
            pk.w1, pk.b1, pk.half = pack_linear(w1 * 1.702), (bf1 * 1.702).contiguous(), 0
            pk.w2 = pack_linear(self.mlp.fc2.weight.detach().float() / 1.702)
        self._pk = pk


class CLIPEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class CLIPVisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = CLIPVisionEmbeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)      # the upstream parameter name
        self.encoder = CLIPEncoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPVisionModelWithProjection(nn.Module):
    """transformers' class of the same name: parameter holder + HIP forward; ``model(pixel_values).image_embeds``"""

    def __init__(self, config: Optional[CLIPVisionConfig] = None, **kw):
        super().__init__()
        cfg = config if config is not None else CLIPVisionConfig(**kw)
        d, h = cfg.hidden_size, cfg.num_attention_heads
        if d % h or (d // h) % 8 or d // h > 128 or d % 64 or cfg.intermediate_size % 64 or cfg.projection_dim % 8:
            raise LkgdHipError("CLIP on the HIP path: head_dim a multiple of 8 up to 128, hidden / intermediate sizes multiples "
                               "of 64, projection_dim a multiple of 8")
        if cfg.image_size % cfg.patch_size:
            raise LkgdHipError("CLIP on the HIP path: image_size must be a multiple of patch_size")
        if cfg.hidden_act not in ("gelu", "quick_gelu"):
            raise LkgdHipError(f"CLIP on the HIP path: hidden_act {cfg.hidden_act!r} (gelu and quick_gelu are implemented)")
        self.config = cfg
        self.vision_model = CLIPVisionTransformer(cfg)
        self.visual_projection = nn.Linear(d, cfg.projection_dim, bias=False)
        self._pk = None

    # ---- loading ------------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype=torch.float16, dtype=None,
                        variant: Optional[str] = None, **_ignored):
        """a local ``image_encoder/`` directory: ``config.json`` + ``model[.<variant>].safetensors`` (or ``pytorch_model.bin``)"""
        from .loading import load_state_dict
        raw = load_config(path, subfolder)
        known = {k: v for k, v in raw.items() if k in CLIPVisionConfig.__dataclass_fields__}
        with torch.device("meta"):
            model = cls(CLIPVisionConfig(**known))
        try:
            sd = load_state_dict(path, subfolder, variant, weights_name="model")
        except OSError:
            sd = load_state_dict(path, subfolder, variant, weights_name="pytorch_model")
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}       # a persisted buffer of older checkpoints
        model.load_state_dict(sd, strict=True, assign=True)
        return model.to(dtype if dtype is not None else torch_dtype)

    def save_pretrained(self, path: str, **_kw):
        import json
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        cfg = {"architectures": ["CLIPVisionModelWithProjection"], "model_type": "clip_vision_model"}
        cfg.update(self.config.__dict__)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(cfg, fh, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))

    @property
    def device(self):
        return self.visual_projection.weight.device

    @property
    def dtype(self):
        return self.visual_projection.weight.dtype

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._pk = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pk = None
        return r

    # ---- forward ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def prepare(self):
        if self._pk is not None:
            return
        if self.device.type != "cuda":
            raise LkgdHipError("lkgd_amd CLIP runs on MI355X only: move the module to cuda first")
        cfg, vm = self.config, self.vision_model
        for layer in vm.encoder.layers:
            layer.pack(cfg.hidden_act)
        emb = vm.embeddings
        kp = cfg.num_channels * cfg.patch_size ** 2
        kpad = (kp + 63) // 64 * 64
        wpe = torch.zeros(cfg.hidden_size, kpad, dtype=torch.float16, device=self.device)
        wpe[:, :kp] = emb.patch_embedding.weight.detach().reshape(cfg.hidden_size, kp).to(torch.float16)
        pos = emb.position_embedding.weight.detach().float()
        self._pk = SimpleNamespace(
            wpe=wpe, kp=kp, kpad=kpad, pos=pos[1:].to(torch.float16).contiguous(),
            cls=(emb.class_embedding.detach().float() + pos[0]).to(torch.float16).contiguous(),
            g_pre=_f32(vm.pre_layrnorm.weight), b_pre=_f32(vm.pre_layrnorm.bias),
            g_post=_f32(vm.post_layernorm.weight), b_post=_f32(vm.post_layernorm.bias),
            wproj=pack_linear(self.visual_projection.weight))

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, **_kw):
        """pixel_values [N, 3, image_size, image_size] (already resized and normalised by the feature extractor, reference
        :166-183) -> namespace with ``image_embeds`` [N, projection_dim] (in the module's dtype) and ``last_hidden_state``"""
        self.prepare()
        cfg, pk, dev = self.config, self._pk, self.device
        g = cfg.image_size // cfg.patch_size
        if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (cfg.num_channels, cfg.image_size, cfg.image_size):
            raise ValueError(f"expected pixel_values [N, {cfg.num_channels}, {cfg.image_size}, {cfg.image_size}]")
        N, D, P = pixel_values.shape[0], cfg.hidden_size, g * g
        S = P + 1
        eps = cfg.layer_norm_eps
        # the patch rows [N*P, C*p*p] in the conv weight's (c, ky, kx) order, K zero-padded to a multiple of 64 (layout only)
        x = pixel_values.to(device=dev, dtype=torch.float16)
        patches = torch.zeros(N * P, pk.kpad, dtype=torch.float16, device=dev)
        patches[:, :pk.kp] = (x.reshape(N, cfg.num_channels, g, cfg.patch_size, g, cfg.patch_size)
                              .permute(0, 2, 4, 1, 3, 5).reshape(N * P, pk.kp))
        emb = torch.empty(N * S, D, dtype=torch.float16, device=dev)
        emb.view(N, S, D)[:, 0] = pk.cls
        for n in range(N):       # one GEMM per image: its patch rows land behind the image's class row, + position[1 + m]
            ops.gemm(patches[n * P:(n + 1) * P], pk.wpe, emb[n * S + 1:(n + 1) * S], M=P, N=D, K=pk.kpad,
                     rowbias=pk.pos, rowmap=(1, 1, 1, 1 << 30))
        tok = ops.layernorm(emb, pk.g_pre, pk.b_pre, eps)
        T, heads = N * S, cfg.num_attention_heads
        hd = D // heads
        for layer in self.vision_model.encoder.layers:
            p = layer._pk
            ln = ops.layernorm(tok, None, None, eps)
            qkv = torch.empty(T, 3 * D, dtype=torch.float16, device=dev)
            ops.gemm(ln, p.wqkv, qkv, M=T, N=3 * D, K=D, bias=p.bqkv)
            att = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.attn_dense(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], att, N, S, heads, hd)
            t1 = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.gemm(att, p.wo, t1, M=T, N=D, K=D, bias=p.bo, res1=tok)
            ln2 = ops.layernorm(t1, None, None, eps)
            inner = p.w2.shape[1]
            h = torch.empty(T, inner, dtype=torch.float16, device=dev)
            if p.act == "gelu":
                ops.gemm(ln2, p.w1, h, M=T, N=2 * inner, K=D, bias=p.b1, geglu=p.half)
            else:
                ops.gemm(ln2, p.w1, h, M=T, N=inner, K=D, bias=p.b1)
                h = ops.silu(h)
            tok = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.gemm(h, p.w2, tok, M=T, N=D, K=inner, bias=p.b2, res1=t1)
        cls = tok.view(N, S, D)[:, 0].contiguous()
        pooled = ops.layernorm(cls, pk.g_post, pk.b_post, eps)
        embeds = torch.empty(N, cfg.projection_dim, dtype=torch.float16, device=dev)
        ops.gemm(pooled, pk.wproj, embeds, M=N, N=cfg.projection_dim, K=D)
        return SimpleNamespace(image_embeds=embeds.to(self.dtype), last_hidden_state=tok.view(N, S, D))
