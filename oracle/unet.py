"""fp32 restatement of the reference's top-level SVD UNets.  ORACLE - test infrastructure only.

Follows
* /root/reference/models/unet_spatio_temporal_condition_controlnet.py:70-245 (construction), :358-508 (forward)
  -> ``UNetSpatioTemporalConditionControlNetModel`` (stock signature), and
* /root/reference/models/unet_spatio_temporal_condition.py:197-225 (LK parameters), :448-693 (forward, LK fuse
  :536-595) -> ``UNetSpatioTemporalConditionModel`` (LKGD signature with positional domain/flow features).

Pinned: tests/golden/unet_wiring_*.safetensors hold outputs of the reference's own ``forward`` executed (via name-only
stubs, tests/golden/make_goldens.py) over the blocks of oracle/blocks.py with the same weights.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import (QuaternionLinearAutograd, TimestepEmbedding, Timesteps, UNetMidBlockSpatioTemporal,
                     get_down_block, get_up_block)


@dataclass
class UNetConfig:
    """Keyword arguments of the reference constructors (unet_..._controlnet.py:70-96)."""
    sample_size: Optional[int] = 96
    in_channels: int = 8
    out_channels: int = 4
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                         "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal")
    up_block_types: Tuple[str, ...] = ("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                                       "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal")
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    layers_per_block: int = 2
    cross_attention_dim: int = 1024
    transformer_layers_per_block: int = 1
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)   # real SVD config; the class default is (5,10,10,20)
    num_frames: int = 14


SVD_CONFIG = UNetConfig()
#: small config used by the parity tests: same topology, head_dim 64, 32-group norms valid, K multiples of 64
TINY_CONFIG = UNetConfig(sample_size=8, block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2),
                         addition_time_embed_dim=64, projection_class_embeddings_input_dim=192,
                         cross_attention_dim=1024, num_frames=4)


class _UNetBase(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.config = SimpleNamespace(**cfg.__dict__)
        boc = cfg.block_out_channels
        n = len(cfg.down_block_types)
        heads = cfg.num_attention_heads if not isinstance(cfg.num_attention_heads, int) else (cfg.num_attention_heads,) * n
        cross = (cfg.cross_attention_dim,) * n
        lpb = [cfg.layers_per_block] * n
        tlpb = [cfg.transformer_layers_per_block] * n

        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        ted = boc[0] * 4
        self.time_proj = Timesteps(boc[0], True, 0)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.add_time_proj = Timesteps(cfg.addition_time_embed_dim, True, 0)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted)

        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            self.down_blocks.append(get_down_block(
                t, num_layers=lpb[i], transformer_layers_per_block=tlpb[i], in_channels=in_ch, out_channels=out_ch,
                temb_channels=ted, add_downsample=i != n - 1, resnet_eps=1e-5, cross_attention_dim=cross[i],
                num_attention_heads=heads[i], resnet_act_fn="silu"))
        self._init_extra(cfg)
        self.mid_block = UNetMidBlockSpatioTemporal(boc[-1], temb_channels=ted, transformer_layers_per_block=tlpb[-1],
                                                    cross_attention_dim=cross[-1], num_attention_heads=heads[-1])
        rboc, rheads = list(reversed(boc)), list(reversed(heads))
        rlpb, rcross, rtlpb = list(reversed(lpb)), list(reversed(cross)), list(reversed(tlpb))
        out_ch = rboc[0]
        for i, t in enumerate(cfg.up_block_types):
            prev = out_ch
            out_ch = rboc[i]
            in_ch = rboc[min(i + 1, n - 1)]
            self.up_blocks.append(get_up_block(
                t, num_layers=rlpb[i] + 1, transformer_layers_per_block=rtlpb[i], in_channels=in_ch,
                out_channels=out_ch, prev_output_channel=prev, temb_channels=ted, add_upsample=i != n - 1,
                resnet_eps=1e-5, resolution_idx=i, cross_attention_dim=rcross[i], num_attention_heads=rheads[i],
                resnet_act_fn="silu"))
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def _init_extra(self, cfg):
        pass

    # -- shared body: reference unet_..._controlnet.py:388-508 ---------------------------------------------------
    def _embed(self, sample, timestep, added_time_ids):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            dtype = torch.float64 if isinstance(timestep, float) else torch.int64
            timesteps = torch.tensor([timesteps], dtype=dtype, device=sample.device)
        elif timesteps.ndim == 0:
            timesteps = timesteps[None].to(sample.device)
        batch_size = sample.shape[0]
        timesteps = timesteps.expand(batch_size)
        t_emb = self.time_proj(timesteps).to(sample.dtype)
        emb = self.time_embedding(t_emb)
        time_embeds = self.add_time_proj(added_time_ids.flatten()).reshape(batch_size, -1).to(emb.dtype)
        return emb + self.add_embedding(time_embeds)

    def _body(self, sample, emb, encoder_hidden_states, down_block_additional_residuals,
              mid_block_additional_residual):
        batch_size, num_frames = sample.shape[:2]
        sample = sample.flatten(0, 1)
        emb = emb.repeat_interleave(num_frames, dim=0)
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_frames, dim=0)
        sample = self.conv_in(sample)
        ioi = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)

        res_samples = (sample,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                sample, r = blk(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                image_only_indicator=ioi)
            else:
                sample, r = blk(sample, temb=emb, image_only_indicator=ioi)
            res_samples += r
            # the reference adds the ControlNet residuals INSIDE the block loop and `zip` truncates
            # (unet_..._controlnet.py:453-462) -> early skips receive their residual several times (App. C1)
            if down_block_additional_residuals is not None:
                new = ()
                for rs, add in zip(res_samples, down_block_additional_residuals):
                    new = new + (rs + add,)
                res_samples = new

        sample = self.mid_block(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                image_only_indicator=ioi)
        if mid_block_additional_residual is not None:
            sample = sample + mid_block_additional_residual

        for blk in self.up_blocks:
            k = len(blk.resnets)
            r, res_samples = res_samples[-k:], res_samples[:-k]
            if blk.has_cross_attention:
                sample = blk(sample, temb=emb, res_hidden_states_tuple=r,
                             encoder_hidden_states=encoder_hidden_states, image_only_indicator=ioi)
            else:
                sample = blk(sample, temb=emb, res_hidden_states_tuple=r, image_only_indicator=ioi)

        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        return sample.reshape(batch_size, num_frames, *sample.shape[1:])


class UNetSpatioTemporalConditionControlNetModel(_UNetBase):
    """Stock signature - reference unet_spatio_temporal_condition_controlnet.py:358-368."""

    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict: bool = True, added_time_ids=None):
        emb = self._embed(sample, timestep, added_time_ids)
        out = self._body(sample, emb, encoder_hidden_states, down_block_additional_residuals,
                         mid_block_additional_residual)
        return SimpleNamespace(sample=out) if return_dict else (out,)


class UNetSpatioTemporalConditionModel(_UNetBase):
    """LKGD signature - reference unet_spatio_temporal_condition.py:448-459 (domain/flow features positional)."""

    def _init_extra(self, cfg):  # reference :197-225
        def dw():
            return nn.Conv1d(1024, 256, kernel_size=1, groups=256, bias=False)
        self.quaternion_lora_dconv = dw()
        self.quaternion_lora_lconv = dw()
        self.quaternion_lora_fconv = dw()
        self.quaternion_lora_fuse = QuaternionLinearAutograd(1024, 512)
        self.quaternion_lora_fuse_fft_mag = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_pha = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_mag0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_fft_pha0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_sf = nn.Sequential(nn.Linear(1024, 256), nn.LeakyReLU(0.1, inplace=True),
                                                     nn.Linear(256, 1024))
        self.quaternion_lora_texts = nn.Parameter(torch.zeros(256))
        self.quaternion_lora_texts_fft_mag = nn.Parameter(torch.zeros(129))
        self.quaternion_lora_texts_fft_pha = nn.Parameter(torch.zeros(129))

    def lk_fuse(self, encoder_hidden_states, domain_features, flow_features):
        """Latent-knowledge fuse, reference :536-595.  Returns the tensor that REPLACES encoder_hidden_states."""
        low = self.quaternion_lora_lconv(encoder_hidden_states.permute(0, 2, 1)).permute(0, 2, 1)
        domain_features = F.interpolate(domain_features, size=1024, mode="linear")
        low_d = self.quaternion_lora_dconv(domain_features.permute(0, 2, 1)).permute(0, 2, 1)
        flow_features = F.interpolate(flow_features, size=1024, mode="linear")
        low_f = self.quaternion_lora_fconv(flow_features.permute(0, 2, 1)).permute(0, 2, 1)
        if low_d.shape[0] != low.shape[0] and low_d.shape[0] == 1:
            low_d = torch.cat([low_d, low_d], dim=0)
            low_f = torch.cat([low_f, low_f], dim=0)
        ctx = self.quaternion_lora_texts.expand_as(low)
        spatial = self.quaternion_lora_fuse(torch.cat([low, low_d, low_f, ctx], dim=-1))

        h_fft, d_fft, f_fft = (torch.fft.rfft(t, dim=-1) for t in (low, low_d, low_f))
        h_mag, h_pha = torch.abs(h_fft), torch.angle(h_fft)
        d_mag, d_pha = torch.abs(d_fft), torch.angle(d_fft)
        f_mag, f_pha = torch.abs(f_fft), torch.angle(f_fft)
        c_mag = self.quaternion_lora_texts_fft_mag.expand_as(h_fft)
        c_pha = self.quaternion_lora_texts_fft_pha.expand_as(h_fft)
        mag = self.quaternion_lora_fuse_fft_mag(
            torch.cat([h_mag[..., :-1], d_mag[..., :-1], f_mag[..., :-1], c_mag[..., :-1]], dim=-1))
        pha = self.quaternion_lora_fuse_fft_pha(
            torch.cat([h_pha[..., :-1], d_pha[..., :-1], f_pha[..., :-1], c_pha[..., :-1]], dim=-1))
        spec = torch.complex(mag * torch.cos(pha), mag * torch.sin(pha))
        mag0 = self.quaternion_lora_fuse_fft_mag0(
            torch.cat([h_mag[..., -1], d_mag[..., -1], f_mag[..., -1], c_mag[..., -1]], dim=-1))
        pha0 = self.quaternion_lora_fuse_fft_pha0(
            torch.cat([h_pha[..., -1], d_pha[..., -1], f_pha[..., -1], c_pha[..., -1]], dim=-1))
        spec0 = torch.complex(mag0 * torch.cos(pha0), mag0 * torch.sin(pha0))
        spec = torch.cat([spec, spec0.unsqueeze(-1)], dim=-1)       # 257 bins
        freq = torch.fft.irfft(spec, dim=-1)                         # length 512
        return self.quaternion_lora_fuse_sf(torch.cat([spatial, freq], dim=-1))

    def forward(self, sample, timestep, encoder_hidden_states, domain_features, flow_features,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict: bool = True, added_time_ids=None):
        emb = self._embed(sample, timestep, added_time_ids)
        encoder_hidden_states = self.lk_fuse(encoder_hidden_states, domain_features, flow_features)
        out = self._body(sample, emb, encoder_hidden_states, down_block_additional_residuals,
                         mid_block_additional_residual)
        return SimpleNamespace(sample=out) if return_dict else (out,)


def init_weights_(model: nn.Module, seed: int = 0, gain: float = 1.0) -> nn.Module:
    """Deterministic synthetic weights: zero-mean fan-in normal so activations stay O(1); norm affine
    parameters near (1, 0) with a small perturbation so they are exercised; mix_factor ~ N(0, 1); biases small."""
    g = torch.Generator().manual_seed(seed)

    def rn(shape):
        return torch.randn(tuple(shape), generator=g)

    with torch.no_grad():
        for mod_name, m in model.named_modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                m.weight.copy_(1.0 + 0.1 * rn(m.weight.shape))
                m.bias.copy_(0.1 * rn(m.bias.shape))
            elif isinstance(m, (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(rn(m.weight.shape) * (gain / fan_in ** 0.5))
                if m.bias is not None:
                    m.bias.copy_(0.02 * rn(m.bias.shape))
            elif isinstance(m, QuaternionLinearAutograd):
                fan_in = m.r_weight.shape[0] * 4
                for w in (m.r_weight, m.i_weight, m.j_weight, m.k_weight):
                    w.copy_(rn(w.shape) * (gain / fan_in ** 0.5))
                m.bias.copy_(0.02 * rn(m.bias.shape))
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.copy_(rn(p.shape))
            elif name.startswith("quaternion_lora_texts"):
                p.copy_(rn(p.shape))
    return model
