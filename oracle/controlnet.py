"""fp32 restatement of the reference's ControlNet-SVD encoder.  ORACLE - test infrastructure only.

Follows /root/reference/models/controlnet_sdv.py: ``ControlNetConditioningEmbeddingSVD`` :64-119 (conv_in, three
(3x3, 3x3 stride 2) pairs, zero-initialised conv_out, SiLU between), ``ControlNetSDVModel.__init__`` :156-317 (the UNet's
conv_in / embeddings / down blocks / mid block plus one zero-initialised 1x1 convolution per skip and one for the mid
block) and ``forward`` :441-578 (conditioning embedding added after conv_in, encoder, zero convs, conditioning_scale).

Pinned: tests/golden/controlnet.safetensors holds outputs of the reference's own class executed (through the name-only
stubs of tests/golden/make_goldens.py) over the blocks of oracle/blocks.py with the same weights.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import TimestepEmbedding, Timesteps, UNetMidBlockSpatioTemporal, get_down_block
from .unet import UNetConfig


class ControlNetConditioningEmbeddingSVD(nn.Module):
    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int, ...] = (16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(conditioning_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 1):
            cin, cout = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(cin, cin, 3, padding=1))
            self.blocks.append(nn.Conv2d(cin, cout, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(block_out_channels[-1], conditioning_embedding_channels, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)
        nn.init.zeros_(self.conv_out.bias)

    def forward(self, conditioning):
        b, f, c, h, w = conditioning.shape
        x = F.silu(self.conv_in(conditioning.view(b * f, c, h, w)))
        for blk in self.blocks:
            x = F.silu(blk(x))
        return self.conv_out(x)


class ControlNetSDVModel(nn.Module):
    def __init__(self, cfg: UNetConfig, conditioning_channels: int = 3,
                 conditioning_embedding_out_channels: Tuple[int, ...] = (16, 32, 96, 256)):
        super().__init__()
        self.config = SimpleNamespace(**cfg.__dict__, conditioning_channels=conditioning_channels,
                                      conditioning_embedding_out_channels=conditioning_embedding_out_channels)
        boc = cfg.block_out_channels
        n = len(cfg.down_block_types)
        heads = cfg.num_attention_heads if not isinstance(cfg.num_attention_heads, int) else (cfg.num_attention_heads,) * n
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        ted = boc[0] * 4
        self.time_proj = Timesteps(boc[0], True, 0)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.add_time_proj = Timesteps(cfg.addition_time_embed_dim, True, 0)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted)
        # attribute order = the reference's registration order (:222-249): synthetic-weight generators walk named_modules()
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList()
        self.controlnet_cond_embedding = ControlNetConditioningEmbeddingSVD(boc[0], conditioning_channels,
                                                                            conditioning_embedding_out_channels)

        def zero_conv(c):
            m = nn.Conv2d(c, c, 1)
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
            return m
        out_ch = boc[0]
        self.controlnet_down_blocks.append(zero_conv(out_ch))
        for i, t in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            last = i == n - 1
            self.down_blocks.append(get_down_block(
                t, num_layers=cfg.layers_per_block, transformer_layers_per_block=cfg.transformer_layers_per_block,
                in_channels=in_ch, out_channels=out_ch, temb_channels=ted, add_downsample=not last, resnet_eps=1e-5,
                cross_attention_dim=cfg.cross_attention_dim, num_attention_heads=heads[i], resnet_act_fn="silu"))
            for _ in range(cfg.layers_per_block):
                self.controlnet_down_blocks.append(zero_conv(out_ch))
            if not last:
                self.controlnet_down_blocks.append(zero_conv(out_ch))
        self.controlnet_mid_block = zero_conv(boc[-1])
        self.mid_block = UNetMidBlockSpatioTemporal(boc[-1], temb_channels=ted,
                                                    transformer_layers_per_block=cfg.transformer_layers_per_block,
                                                    cross_attention_dim=cfg.cross_attention_dim,
                                                    num_attention_heads=heads[-1])

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None,
                image_only_indicator=None, return_dict=True, guess_mode=False, conditioning_scale=1.0):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            dtype = torch.float64 if isinstance(timestep, float) else torch.int64
            timesteps = torch.tensor([timesteps], dtype=dtype, device=sample.device)
        elif timesteps.ndim == 0:
            timesteps = timesteps[None].to(sample.device)
        batch_size, num_frames = sample.shape[:2]
        timesteps = timesteps.expand(batch_size)
        emb = self.time_embedding(self.time_proj(timesteps).to(sample.dtype))
        te = self.add_time_proj(added_time_ids.flatten()).reshape(batch_size, -1).to(emb.dtype)
        emb = emb + self.add_embedding(te)
        sample = sample.flatten(0, 1)
        emb = emb.repeat_interleave(num_frames, dim=0)
        enc = encoder_hidden_states.repeat_interleave(num_frames, dim=0)
        sample = self.conv_in(sample)
        if controlnet_cond is not None:
            sample = sample + self.controlnet_cond_embedding(controlnet_cond)
        ioi = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)
        res = (sample,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                sample, r = blk(sample, temb=emb, encoder_hidden_states=enc, image_only_indicator=ioi)
            else:
                sample, r = blk(sample, temb=emb, image_only_indicator=ioi)
            res += r
        sample = self.mid_block(sample, temb=emb, encoder_hidden_states=enc, image_only_indicator=ioi)
        down = [blk(r) * conditioning_scale for r, blk in zip(res, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(sample) * conditioning_scale
        if not return_dict:
            return down, mid
        return SimpleNamespace(down_block_res_samples=down, mid_block_res_sample=mid)
