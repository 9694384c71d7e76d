"""The domain / flow ViT-B/16-384 encoders of LKGD on the MI355X path.

The reference builds two ``timm`` ``vit_base_patch16_384()`` models [EXT] and turns a clip into the latent-knowledge inputs of
the UNet: /root/reference/train_models/train_svd_lora.py:1408-1433 (construction; weights = the ``encoder.*`` keys of a
checkpoint) and :1455-1466 (``F.interpolate(size=[384, 384], mode="bilinear")`` -> logits [N, 1000] -> mean over the clip's
frames -> ``domain_features`` / ``flow_features`` [B, 1, 1000]); the inference wiring is
CogVideo-main/finetune/models/cogvideox_i2v/pipeline_cogvideox_image2video.py:794-799.  Parameter names are timm's, so those
checkpoints load (``load_encoder_checkpoint``).  oracle/vit.py is the fp32 restatement the tests compare with (PARITY UNPINNED:
timm is third-party and absent).

Runs once per clip (not on the per-step path): 12 blocks of [N*577, 768] tokens on the UNet's kernels -
* bilinear resize + 16x16 patch unfold in one kernel (``lkgd_vit_patchify``), patch embedding = one GEMM per image whose
  epilogue adds the position embedding (row-indexed bias) and writes straight behind the image's cls row;
* LayerNorm affine folded into qkv / fc1 (as in the UNet), fused qkv GEMM with bias, the head_dim-64 flash attention kernel
  (S = 577, ragged last tile), out-projection / fc2 with the residual in the epilogue;
* exact-erf GELU through the GEGLU epilogue with a constant-one "hidden" half (zero weights, bias 1): out = 1 * gelu(fc1 x).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import LkgdHipError
from .packing import pack_geglu, pack_linear


@dataclass
class ViTConfig:
    img_size: int = 384
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: int = 4
    num_classes: int = 1000


def _f32(p):
    return p.detach().to(torch.float32).contiguous()


def _fold_ln(norm: nn.LayerNorm, w: torch.Tensor, b: torch.Tensor):
    g, be = norm.weight.detach().float(), norm.bias.detach().float()
    w32 = w.detach().float()
    return w32 * g[None, :], w32 @ be + b.detach().float()


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, heads, ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * ratio)

    def pack(self):
        wq, bq = _fold_ln(self.norm1, self.attn.qkv.weight, self.attn.qkv.bias)
        w1, b1 = _fold_ln(self.norm2, self.mlp.fc1.weight, self.mlp.fc1.bias)
        # GELU(fc1 x) as GEGLU with a constant-one hidden half: hidden = 0 * x + 1, gate = fc1 x
        wg = torch.cat([torch.zeros_like(w1), w1], dim=0)
        bg = torch.cat([torch.ones_like(b1), b1], dim=0)
        wp, bp, half = pack_geglu(wg, bg)
        self._pk = SimpleNamespace(wqkv=pack_linear(wq), bqkv=bq.contiguous(), wo=pack_linear(self.attn.proj.weight),
                                   bo=_f32(self.attn.proj.bias), w1=wp, b1=bp, half=half, w2=pack_linear(self.mlp.fc2.weight),
                                   b2=_f32(self.mlp.fc2.bias))


class PatchEmbed(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.proj = nn.Conv2d(cfg.in_chans, cfg.embed_dim, cfg.patch_size, stride=cfg.patch_size)


class VisionTransformer(nn.Module):
    """timm ``VisionTransformer`` (vit_base_patch16_384 by default) - parameter holder + HIP forward"""

    def __init__(self, config: Optional[ViTConfig] = None, **kw):
        super().__init__()
        cfg = config if config is not None else ViTConfig(**kw)
        if cfg.embed_dim // cfg.num_heads != 64 or cfg.embed_dim % 64 or (cfg.in_chans * cfg.patch_size ** 2) % 64:
            raise LkgdHipError("ViT on the HIP path: head_dim 64, embed_dim and in_chans*patch^2 multiples of 64")
        if cfg.num_classes % 8 or cfg.img_size % cfg.patch_size:
            raise LkgdHipError("ViT on the HIP path: num_classes % 8 == 0, img_size % patch_size == 0")
        self.cfg = cfg
        n = (cfg.img_size // cfg.patch_size) ** 2
        self.patch_embed = PatchEmbed(cfg)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, cfg.embed_dim))
        self.blocks = nn.ModuleList([Block(cfg.embed_dim, cfg.num_heads, cfg.mlp_ratio) for _ in range(cfg.depth)])
        self.norm = nn.LayerNorm(cfg.embed_dim, eps=1e-6)
        self.head = nn.Linear(cfg.embed_dim, cfg.num_classes)
        self._pk = None

    @property
    def device(self):
        return self.cls_token.device

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._pk = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pk = None
        return r

    def load_encoder_checkpoint(self, path_or_state):
        """train_svd_lora.py:1419-1433: keep the ``encoder.*`` keys of the checkpoint, strip the prefix, load strictly"""
        sd = torch.load(path_or_state, map_location="cpu", weights_only=True) if isinstance(path_or_state, str) else path_or_state
        enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        return self.load_state_dict(enc)

    @torch.no_grad()
    def prepare(self):
        if self._pk is not None:
            return
        if self.device.type != "cuda":
            raise LkgdHipError("lkgd_amd ViT runs on MI355X only: move the module to cuda first")
        for b in self.blocks:
            b.pack()
        pe = self.patch_embed.proj
        pos = self.pos_embed.detach().float()[0]
        self._pk = SimpleNamespace(
            wpe=pack_linear(pe.weight.detach()), bpe=_f32(pe.bias),
            pos=pos[1:].to(torch.float16).contiguous(),
            cls=(self.cls_token.detach().float()[0, 0] + pos[0]).to(torch.float16).contiguous(),
            gn=_f32(self.norm.weight), bn=_f32(self.norm.bias), wh=pack_linear(self.head.weight), bh=_f32(self.head.bias))

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """images [N, 3, H, W] of any size (resized bilinearly to img_size, as the reference does before the call) ->
        logits [N, num_classes] fp32"""
        self.prepare()
        cfg, pk, dev = self.cfg, self._pk, self.device
        if x.dim() != 4 or x.shape[1] != cfg.in_chans:
            raise ValueError(f"expected [N, {cfg.in_chans}, H, W]")
        N = x.shape[0]
        D, g = cfg.embed_dim, cfg.img_size // cfg.patch_size
        S, P = g * g + 1, g * g
        patches = ops.vit_patchify(x.to(device=dev, dtype=torch.float32), cfg.img_size, cfg.patch_size)     # [N*P, C*p*p]
        tok = torch.empty(N * S, D, dtype=torch.float16, device=dev)
        tok.view(N, S, D)[:, 0] = pk.cls
        Kp = patches.shape[1]
        for n in range(N):       # one GEMM per image: its patch rows land behind the image's cls row, + pos_embed[1 + m]
            ops.gemm(patches[n * P:(n + 1) * P], pk.wpe, tok[n * S + 1:(n + 1) * S], M=P, N=D, K=Kp, bias=pk.bpe,
                     rowbias=pk.pos, rowmap=(1, 1, 1, 1 << 30))
        T, heads = N * S, cfg.num_heads
        for b in self.blocks:
            p = b._pk
            ln = ops.layernorm(tok, None, None, 1e-6)
            qkv = torch.empty(T, 3 * D, dtype=torch.float16, device=dev)
            ops.gemm(ln, p.wqkv, qkv, M=T, N=3 * D, K=D, bias=p.bqkv)
            att = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.attn_spatial(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], att, N, S, heads)
            t1 = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.gemm(att, p.wo, t1, M=T, N=D, K=D, bias=p.bo, res1=tok)
            ln2 = ops.layernorm(t1, None, None, 1e-6)
            inner = p.w2.shape[1]
            h = torch.empty(T, inner, dtype=torch.float16, device=dev)
            ops.gemm(ln2, p.w1, h, M=T, N=2 * inner, K=D, bias=p.b1, geglu=p.half)
            tok = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.gemm(h, p.w2, tok, M=T, N=D, K=inner, bias=p.b2, res1=t1)
        cls = tok.view(N, S, D)[:, 0].contiguous()                      # global_pool = "token"
        cls = ops.layernorm(cls, pk.gn, pk.bn, 1e-6)
        logits = torch.empty(N, cfg.num_classes, dtype=torch.float16, device=dev)
        ops.gemm(cls, pk.wh, logits, M=N, N=cfg.num_classes, K=D, bias=pk.bh)
        return logits.float()

    @torch.no_grad()
    def clip_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """train_svd_lora.py:1455-1461: frames [B, T, C, H, W] -> [B, 1, num_classes] (mean of the per-frame logits)"""
        if pixel_values.dim() != 5:
            raise ValueError("pixel_values must be [batch, frames, channels, height, width]")
        B, T = pixel_values.shape[:2]
        return self.forward(pixel_values.flatten(0, 1)).reshape(B, T, -1).mean(dim=1, keepdim=True)


def vit_base_patch16_384(**kw) -> VisionTransformer:
    """timm's constructor name, as the reference calls it (train_svd_lora.py:1408,1414)"""
    return VisionTransformer(ViTConfig(**kw))
