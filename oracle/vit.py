"""fp32 restatement of timm's ``vit_base_patch16_384`` (the domain / flow encoders of LKGD).  ORACLE - test infrastructure only.

The reference builds them at /root/reference/train_models/train_svd_lora.py:1408-1433 (``vit_base_patch16_384()``, weights
from the ``encoder.*`` keys of a MAE-style checkpoint) and evaluates them at :1455-1466:
``F.interpolate(images, size=[384, 384], mode="bilinear")`` -> model -> logits [N, 1000] -> mean over the clip's frames ->
``domain_features`` / ``flow_features`` [B, 1, 1000] (the same wiring in
CogVideo-main/finetune/models/cogvideox_i2v/pipeline_cogvideox_image2video.py:794-799).

**[EXT] - PARITY UNPINNED**: ``timm.models.vision_transformer`` is neither vendored under /root/reference nor installable here;
this restates the published model (timm 0.9: PatchEmbed conv 16x16 / 16, cls token, learned position embedding [1, 577, 768],
12 pre-norm blocks with LayerNorm eps 1e-6, fused qkv with bias, 12 heads of 64, MLP 4x with exact GELU, final LayerNorm,
``global_pool="token"``, linear head) with timm's parameter names, so the reference's checkpoints load.  Structural gate:
86 859 496 parameters.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


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


VIT_B16_384 = ViTConfig()
TINY_VIT = ViTConfig(img_size=64, embed_dim=128, depth=2, num_heads=2, num_classes=40)


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(o.transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, heads, ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * ratio)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.proj = nn.Conv2d(cfg.in_chans, cfg.embed_dim, cfg.patch_size, stride=cfg.patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformer(nn.Module):
    def __init__(self, cfg: ViTConfig = VIT_B16_384):
        super().__init__()
        self.cfg = cfg
        n = (cfg.img_size // cfg.patch_size) ** 2
        self.patch_embed = PatchEmbed(cfg)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, cfg.embed_dim))
        self.blocks = nn.ModuleList([Block(cfg.embed_dim, cfg.num_heads, cfg.mlp_ratio) for _ in range(cfg.depth)])
        self.norm = nn.LayerNorm(cfg.embed_dim, eps=1e-6)
        self.head = nn.Linear(cfg.embed_dim, cfg.num_classes)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        for b in self.blocks:
            x = b(x)
        return self.head(self.norm(x)[:, 0])


def clip_features(model: VisionTransformer, pixel_values: torch.Tensor) -> torch.Tensor:
    """train_svd_lora.py:1455-1461: [B, T, C, H, W] -> bilinear 384 x 384 -> logits -> mean over T -> [B, 1, classes]"""
    B, T = pixel_values.shape[:2]
    imgs = pixel_values.flatten(0, 1)
    s = model.cfg.img_size
    imgs = F.interpolate(imgs, size=[s, s], mode="bilinear")
    return model(imgs).reshape(B, T, -1).mean(dim=1, keepdim=True)


def init_weights_(m: nn.Module, seed: int) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(m.named_parameters()):
            if "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name in ("cls_token", "pos_embed"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m
