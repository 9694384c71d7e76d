"""fp32 restatement of diffusers 0.27.2 ``AutoencoderKLTemporalDecoder`` (the SVD VAE).  ORACLE - test infrastructure only.

The reference calls it at /root/reference/pipeline/pipeline_stable_video_diffusion_trans.py:205-226 (``vae.encode(image)
.latent_dist.mode()``) and :256-283 (``vae.decode(latents, num_frames=...)`` in chunks of ``decode_chunk_size``).

**[EXT] - PARITY UNPINNED.**  The class lives in diffusers (``models/autoencoders/autoencoder_kl_temporal_decoder.py``,
``models/autoencoders/vae.py`` Encoder, ``models/unets/unet_3d_blocks.py`` MidBlockTemporalDecoder / UpBlockTemporalDecoder,
``models/resnet.py``, ``models/attention_processor.py``), which is neither vendored under /root/reference nor installable
here; this file restates the published 0.27.2 source with diffusers' parameter names so a real ``vae`` checkpoint loads
(structural gate in the tests: 97 742 847 parameters - 34 163 592 encoder + 63 579 183 temporal decoder + 72 quant_conv - for
the SVD config; the published fp16 checkpoint is 195 MB).  Nothing in the reference's own tests or fixtures exercises the VAE.

Semantics restated (SVD ``vae/config.json``: block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4,
scaling_factor 0.18215, force_upcast true):
* Encoder: conv_in 3->128; four DownEncoderBlock2D (two ResnetBlock2D each, eps 1e-6, no time embedding; a stride-2 3x3
  convolution after the asymmetric pad (0,1,0,1) on all but the last); UNetMidBlock2D (resnet, single-head attention with
  head_dim 512 over the GroupNorm'ed tokens, residual, resnet); GroupNorm + SiLU + conv_out -> 8 moments; ``quant_conv`` 1x1;
  ``latent_dist.mode()`` = the mean = moments[:, :4].
* TemporalDecoder: conv_in 4->512; MidBlockTemporalDecoder (SpatioTemporalResBlock, attention, SpatioTemporalResBlock); four
  UpBlockTemporalDecoder (three SpatioTemporalResBlock each + nearest-2x Upsample2D with a 3x3 conv on all but the last);
  GroupNorm(1e-6) + SiLU + conv_out -> 3; ``time_conv_out`` Conv3d (3,1,1) over the frames of a chunk.
  SpatioTemporalResBlock without time embedding: ResnetBlock2D(eps 1e-6), TemporalResnetBlock(eps 1e-5) on [B,C,F,H,W],
  AlphaBlender("learned", switch_spatial_to_temporal_mix=True): alpha = 1 - sigmoid(mix_factor),
  out = alpha * spatial + (1 - alpha) * temporal.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    sample_size: int = 768
    scaling_factor: float = 0.18215
    force_upcast: bool = True


SVD_VAE_CONFIG = VAEConfig()
TINY_VAE_CONFIG = VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=64)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class TemporalResnetBlock(nn.Module):
    def __init__(self, c: int, eps: float = 1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, c, eps=eps)
        self.conv1 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))
        self.norm2 = nn.GroupNorm(32, c, eps=eps)
        self.conv2 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x):          # [B, C, F, H, W]
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class AlphaBlender(nn.Module):
    def __init__(self, alpha: float = 0.0):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha], dtype=torch.float32))


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, 1e-6)
        self.temporal_res_block = TemporalResnetBlock(cout, 1e-5)
        self.time_mixer = AlphaBlender(0.0)

    def forward(self, x, num_frames: int):
        x = self.spatial_res_block(x)
        bf, c, h, w = x.shape
        b = bf // num_frames
        s = x[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        t = self.temporal_res_block(s)
        alpha = 1.0 - torch.sigmoid(self.time_mixer.mix_factor)        # "learned" + switch_spatial_to_temporal_mix
        y = alpha * s + (1.0 - alpha) * t
        return y.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class Attention(nn.Module):
    """diffusers Attention(query_dim=C, heads=C // head_dim, dim_head=head_dim, norm_num_groups=32, eps=1e-6, bias=True,
    residual_connection=True) with AttnProcessor2_0 on a 4-D input"""

    def __init__(self, c: int, head_dim: int):
        super().__init__()
        self.heads = c // head_dim
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        t = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        hd = c // self.heads
        q, k, v = (z.reshape(b, -1, self.heads, hd).transpose(1, 2) for z in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, -1, c)
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(b, c, h, w) + res


class Downsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class UNetMidBlock2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c, c)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(DownEncoderBlock2D(c, co, cfg.layers_per_block, i != len(boc) - 1))
            c = co
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.conv_norm_out = nn.GroupNorm(32, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, c, layers):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(c, c) for _ in range(layers)])
        self.attentions = nn.ModuleList([Attention(c, c)])

    def forward(self, x, num_frames):
        x = self.resnets[0](x, num_frames)
        for r, a in zip(self.resnets[1:], self.attentions):
            x = r(a(x), num_frames)
        return x


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, cin, cout, layers, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(cin if i == 0 else cout, cout) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, x, num_frames):
        for r in self.resnets:
            x = r(x, num_frames)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class TemporalDecoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(boc[-1], cfg.layers_per_block)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpBlockTemporalDecoder(c, co, cfg.layers_per_block + 1, i != len(boc) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(cfg.out_channels, cfg.out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, z, num_frames):
        x = self.conv_in(z)
        x = self.mid_block(x, num_frames)
        for b in self.up_blocks:
            x = b(x, num_frames)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        bf, c, h, w = x.shape
        b = bf // num_frames
        x = x[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.time_conv_out(x)
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class AutoencoderKLTemporalDecoder(nn.Module):
    def __init__(self, cfg: VAEConfig = SVD_VAE_CONFIG):
        super().__init__()
        self.config = SimpleNamespace(**cfg.__dict__)
        self.encoder = Encoder(cfg)
        self.decoder = TemporalDecoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def encode(self, x):
        moments = self.quant_conv(self.encoder(x))
        mean = moments[:, :self.config.latent_channels]
        return SimpleNamespace(latent_dist=SimpleNamespace(mode=lambda: mean, mean=mean))

    def decode(self, z, num_frames: int = 1):
        return SimpleNamespace(sample=self.decoder(z, num_frames))


def init_weights_(m: nn.Module, seed: int) -> nn.Module:
    """seeded fan-in-normal weights, norm affine ~ (1, 0) + noise, small biases (as oracle.unet.init_weights_)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(m.named_parameters()):
            if "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("mix_factor"):
                p.copy_(torch.randn(p.shape, generator=g))
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m
