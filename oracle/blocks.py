"""fp32 restatement of the diffusers==0.27.2 blocks the SVD UNet is built from.

ORACLE - test infrastructure only (see oracle/__init__.py).  **Parity unpinned**
for this file: diffusers is a third-party dependency of the reference
(/root/reference/requirements.txt:15, ``diffusers==0.27.2``) that is neither
vendored under /root/reference nor installable here.  Each class restates the
published source of that version; the call sites that anchor it are
/root/reference/models/unet_spatio_temporal_condition_controlnet.py:12-13,137-143,
169-181,185-191,218-232 and the in-repo witnesses
/root/reference/patch/patch.py:390-580 (BasicTransformerBlock.forward) and
:582-686 (TemporalBasicTransformerBlock.forward).

Module / parameter names equal diffusers' so a real SVD ``unet`` state-dict loads
unchanged.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- embeddings
class Timesteps(nn.Module):
    """diffusers.models.embeddings.Timesteps (get_timestep_embedding); used at
    reference unet_..._controlnet.py:137,142."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        half = self.num_channels // 2
        exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = torch.exp(exponent)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    """diffusers.models.embeddings.TimestepEmbedding: linear_2(silu(linear_1(x)))."""

    def __init__(self, in_channels: int, time_embed_dim: int, out_dim: Optional[int] = None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


# --------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention with AttnProcessor2_0:
    q/k/v Linear without bias, SDPA with scale 1/sqrt(dim_head), to_out.0 with bias."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.out_dim = query_dim
        self.scale = dim_head ** -0.5
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_k = nn.Linear(kv_dim, self.inner_dim, bias=False)
        self.to_v = nn.Linear(kv_dim, self.inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b, s, _ = hidden_states.shape
        q = self.to_q(hidden_states)
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        hd = self.inner_dim // self.heads
        q = q.view(b, -1, self.heads, hd).transpose(1, 2)
        k = k.view(b, -1, self.heads, hd).transpose(1, 2)
        v = v.view(b, -1, self.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, self.inner_dim)
        o = self.to_out[0](o)
        return self.to_out[1](o)


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)  # exact erf GELU


class FeedForward(nn.Module):
    """diffusers FeedForward(activation_fn="geglu"): net = [GEGLU, Dropout, Linear]."""

    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4):
        super().__init__()
        inner = dim * mult
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """Witness: /root/reference/patch/patch.py:390-580 (non-joint branch :503-508)."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, cross_attention_dim: int):
        super().__init__()
        self.only_cross_attention = False
        self.norm_type = "layer_norm"
        self.pos_embed = None
        self._chunk_size = None
        self._chunk_dim = 0
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, **kw):
        n = self.norm1(hidden_states)
        hidden_states = self.attn1(n) + hidden_states
        n = self.norm2(hidden_states)
        hidden_states = self.attn2(n, encoder_hidden_states=encoder_hidden_states) + hidden_states
        n = self.norm3(hidden_states)
        hidden_states = self.ff(n) + hidden_states
        return hidden_states


class TemporalBasicTransformerBlock(nn.Module):
    """Witness: /root/reference/patch/patch.py:582-686."""

    def __init__(self, dim: int, time_mix_inner_dim: int, num_attention_heads: int, attention_head_dim: int,
                 cross_attention_dim: int):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self._chunk_size = None
        self._chunk_dim = 0
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim)
        self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, hidden_states, num_frames: int, encoder_hidden_states=None):
        batch_frames, seq_length, channels = hidden_states.shape
        batch_size = batch_frames // num_frames
        h = hidden_states[None, :].reshape(batch_size, num_frames, seq_length, channels)
        h = h.permute(0, 2, 1, 3).reshape(batch_size * seq_length, num_frames, channels)
        residual = h
        h = self.ff_in(self.norm_in(h))
        if self.is_res:
            h = h + residual
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states) + h
        ff = self.ff(self.norm3(h))
        h = ff + h if self.is_res else ff
        h = h[None, :].reshape(batch_size, seq_length, num_frames, channels)
        h = h.permute(0, 2, 1, 3).reshape(batch_size * num_frames, seq_length, channels)
        return h


class AlphaBlender(nn.Module):
    """diffusers AlphaBlender(merge_strategy="learned_with_images")."""

    def __init__(self, alpha: float = 0.5):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha], dtype=torch.float32))

    def get_alpha(self, image_only_indicator: torch.Tensor, ndims: int) -> torch.Tensor:
        alpha = torch.where(image_only_indicator.bool(),
                            torch.ones(1, 1, device=image_only_indicator.device),
                            torch.sigmoid(self.mix_factor)[..., None])
        if ndims == 5:
            alpha = alpha[:, None, :, None, None]
        elif ndims == 3:
            alpha = alpha.reshape(-1)[:, None, None]
        else:
            raise ValueError(ndims)
        return alpha

    def forward(self, x_spatial, x_temporal, image_only_indicator):
        alpha = self.get_alpha(image_only_indicator, x_spatial.ndim).to(x_spatial.dtype)
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


#: How TransformerSpatioTemporalModel lays out the temporal cross-attention context rows.
#: "interleaved_0_27": diffusers 0.27.x - rows ordered (pixel, batch) while hidden rows are
#: (batch, pixel).  "batch_major": later diffusers releases.
TIME_CONTEXT_ORDER = "interleaved_0_27"


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, num_attention_heads: int, attention_head_dim: int, in_channels: int,
                 num_layers: int = 1, cross_attention_dim: int = 1024):
        super().__init__()
        inner_dim = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList([
            TemporalBasicTransformerBlock(inner_dim, inner_dim, num_attention_heads, attention_head_dim,
                                          cross_attention_dim)
            for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_proj = Timesteps(in_channels, True, 0)
        self.time_mixer = AlphaBlender(0.5)
        self.proj_out = nn.Linear(inner_dim, in_channels)
        self.time_context_order = None  # None -> module-level TIME_CONTEXT_ORDER

    def forward(self, hidden_states, encoder_hidden_states=None, image_only_indicator=None, return_dict=False):
        batch_frames, _, height, width = hidden_states.shape
        num_frames = image_only_indicator.shape[-1]
        batch_size = batch_frames // num_frames

        time_context = encoder_hidden_states
        first = time_context[None, :].reshape(batch_size, num_frames, -1, time_context.shape[-1])[:, 0]
        order = self.time_context_order or TIME_CONTEXT_ORDER
        if order == "interleaved_0_27":
            time_context = first[None, :].broadcast_to(height * width, batch_size, first.shape[-2], first.shape[-1])
            time_context = time_context.reshape(height * width * batch_size, first.shape[-2], first.shape[-1])
        elif order == "batch_major":
            time_context = first[:, None].broadcast_to(batch_size, height * width, first.shape[-2], first.shape[-1])
            time_context = time_context.reshape(batch_size * height * width, first.shape[-2], first.shape[-1])
        else:
            raise ValueError(order)

        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        inner_dim = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch_frames, height * width, inner_dim)
        hidden_states = self.proj_in(hidden_states)

        num_frames_emb = torch.arange(num_frames, device=hidden_states.device).repeat(batch_size, 1).reshape(-1)
        t_emb = self.time_proj(num_frames_emb).to(hidden_states.dtype)
        emb = self.time_pos_embed(t_emb)[:, None, :]

        for block, temporal_block in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            hidden_states = block(hidden_states, encoder_hidden_states=encoder_hidden_states)
            mix = hidden_states + emb
            mix = temporal_block(mix, num_frames=num_frames, encoder_hidden_states=time_context)
            hidden_states = self.time_mixer(x_spatial=hidden_states, x_temporal=mix,
                                            image_only_indicator=image_only_indicator)

        hidden_states = self.proj_out(hidden_states)
        hidden_states = hidden_states.reshape(batch_frames, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
        output = hidden_states + residual
        return (output,)


# --------------------------------------------------------------------------- res blocks
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class TemporalResnetBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        assert in_channels == out_channels

    def forward(self, x, temb):  # x [B,C,F,H,W], temb [B,F,T]
        h = self.conv1(F.silu(self.norm1(x)))
        t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
        h = h + t
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels, eps)
        self.time_mixer = AlphaBlender(0.5)

    def forward(self, hidden_states, temb, image_only_indicator):
        num_frames = image_only_indicator.shape[-1]
        hidden_states = self.spatial_res_block(hidden_states, temb)
        bf, c, h, w = hidden_states.shape
        b = bf // num_frames
        mix = hidden_states[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        hs = hidden_states[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        temb = temb.reshape(b, num_frames, -1)
        hs = self.temporal_res_block(hs, temb)
        hs = self.time_mixer(x_spatial=mix, x_temporal=hs, image_only_indicator=image_only_indicator)
        return hs.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class Downsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


# --------------------------------------------------------------------------- block containers
#: GroupNorm eps of the res blocks per container (diffusers 0.27.2: CrossAttnDown / Up / CrossAttnUp default
#: ``resnet_eps=1e-6`` and ``get_*_block`` does not forward the 1e-5 the UNet passes; Down and Mid hard-code 1e-5).
#: Kept in ONE table so a wrong recollection is a one-line fix; the 1e-5/1e-6 difference is far below fp16 tolerance.
RESNET_EPS = {
    "CrossAttnDownBlockSpatioTemporal": 1e-6,
    "DownBlockSpatioTemporal": 1e-5,
    "UNetMidBlockSpatioTemporal": 1e-5,
    "UpBlockSpatioTemporal": 1e-6,
    "CrossAttnUpBlockSpatioTemporal": 1e-6,
}


class CrossAttnDownBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, transformer_layers_per_block,
                 num_attention_heads, cross_attention_dim, add_downsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps)
            for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                           transformer_layers_per_block, cross_attention_dim)
            for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, image_only_indicator=None):
        outs = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)[0]
            outs = outs + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs = outs + (hidden_states,)
        return hidden_states, outs


class DownBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, image_only_indicator=None):
        outs = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
            outs = outs + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs = outs + (hidden_states,)
        return hidden_states, outs


class UNetMidBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, transformer_layers_per_block=1,
                 num_attention_heads=1, cross_attention_dim=1280):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps)
                                      for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, in_channels // num_attention_heads, in_channels,
                                           transformer_layers_per_block, cross_attention_dim)
            for _ in range(num_layers)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, image_only_indicator=None):
        hidden_states = self.resnets[0](hidden_states, temb, image_only_indicator)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)[0]
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
        return hidden_states


def _up_resnet_channels(i, num_layers, in_channels, prev_output_channel, out_channels):
    res_skip = in_channels if i == num_layers - 1 else out_channels
    res_in = prev_output_channel if i == 0 else out_channels
    return res_in + res_skip


class UpBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(_up_resnet_channels(i, num_layers, in_channels, prev_output_channel, out_channels),
                                   out_channels, temb_channels, eps)
            for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, image_only_indicator=None):
        for resnet in self.resnets:
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class CrossAttnUpBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 transformer_layers_per_block, num_attention_heads, cross_attention_dim, add_upsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(_up_resnet_channels(i, num_layers, in_channels, prev_output_channel, out_channels),
                                   out_channels, temb_channels, eps)
            for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                           transformer_layers_per_block, cross_attention_dim)
            for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                image_only_indicator=None):
        for resnet, attn in zip(self.resnets, self.attentions):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample,
                   num_attention_heads, resnet_eps=None, cross_attention_dim=None, transformer_layers_per_block=1,
                   resnet_act_fn="silu", **kw):
    """Signature as called at reference unet_..._controlnet.py:169-181."""
    if down_block_type == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers, add_downsample)
    if down_block_type == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers,
                                                transformer_layers_per_block, num_attention_heads,
                                                cross_attention_dim, add_downsample)
    raise ValueError(down_block_type)


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, num_attention_heads, resnet_eps=None, resolution_idx=None, cross_attention_dim=None,
                 transformer_layers_per_block=1, resnet_act_fn="silu", **kw):
    """Signature as called at reference unet_..._controlnet.py:218-232."""
    if up_block_type == "UpBlockSpatioTemporal":
        return UpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                                     add_upsample)
    if up_block_type == "CrossAttnUpBlockSpatioTemporal":
        return CrossAttnUpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels,
                                              num_layers, transformer_layers_per_block, num_attention_heads,
                                              cross_attention_dim, add_upsample)
    raise ValueError(up_block_type)


# --------------------------------------------------------------------------- quaternion linear (core_qnn)
class QuaternionLinearAutograd(nn.Module):
    """Orkis-Research Pytorch-Quaternion-Neural-Networks ``QuaternionLinearAutograd`` (un-pinned dependency of the
    reference, call sites /root/reference/models/unet_spatio_temporal_condition.py:15,213-216).  Forward =
    ``quaternion_linear``: y = x @ W + bias with the Hamilton block matrix.  **Parity unpinned.**
    Weights get a plain fan-in normal init here (the library's quaternion-glorot init is irrelevant to inference)."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        i4, o4 = in_features // 4, out_features // 4
        s = 1.0 / math.sqrt(in_features)
        self.r_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.i_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.j_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.k_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.bias = nn.Parameter(torch.zeros(out_features))

    def hamilton(self) -> torch.Tensor:
        r, i, j, k = self.r_weight, self.i_weight, self.j_weight, self.k_weight
        col_r = torch.cat([r, -i, -j, -k], dim=0)
        col_i = torch.cat([i, r, -k, j], dim=0)
        col_j = torch.cat([j, k, r, -i], dim=0)
        col_k = torch.cat([k, -j, i, r], dim=0)
        return torch.cat([col_r, col_i, col_j, col_k], dim=1)

    def forward(self, x):
        return torch.matmul(x, self.hamilton()) + self.bias
