"""fp32 restatement of the CogVideoX DiT sampling loop with LKGD's latent-knowledge fuse (BASELINE.json
configs[4]).  ORACLE - test infrastructure only.

In-tree reference (restated from files that can be read here, pinned by tests/golden/cogvideox.safetensors = outputs of the
reference's own ``forward``):
* /root/reference/CogVideo-main/finetune/models/cogvideox_i2v/cogvideox_transformer_3d.py - ``CogVideoXBlock`` :41-160,
  ``CogVideoXTransformer3DModel`` :163-335 (construction), ``init_quaternion_modules`` :337-366, ``forward`` :473-638 (time
  embedding, the latent-knowledge fuse on the TEXT embeddings :519-582, patch embedding, 30 blocks, ``norm_final`` on the video
  stream only, ``norm_out``, ``proj_out``, un-patchify);
* /root/reference/CogVideo-main/finetune/models/cogvideox_i2v/pipeline_cogvideox_image2video.py:829-885 - the loop: CFG
  duplication, channel concat with the image latents, dynamic CFG scale, scheduler step.

**[EXT] - PARITY UNPINNED** for what those files import from diffusers >= 0.32 (not vendored, not installable): ``Attention`` with
``qk_norm="layer_norm"`` + ``CogVideoXAttnProcessor2_0``, ``FeedForward("gelu-approximate")``, ``CogVideoXPatchEmbed`` (2x2
patch conv, text projection, 3-D sin-cos position table), ``CogVideoXLayerNormZero``, ``AdaLayerNorm``, ``TimestepEmbedding`` /
``Timesteps``, ``CogVideoXDDIMScheduler`` (SNR-shifted, zero-terminal-SNR alphas, trailing spacing, v-prediction).  They are
restated from the published source with diffusers' parameter names (so a ``transformer/`` checkpoint loads); the golden
fixture executes the reference's in-tree code OVER these restatements, so it pins the wiring, not these interiors.
Structural gate (tests/test_cogvideox.py): 1 693 906 752 parameters for the 2B image-to-video transformer config without the
LK modules (in_channels 32), 1 696 737 100 with them.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import QuaternionLinearAutograd, TimestepEmbedding, Timesteps


@dataclass
class DiTConfig:
    num_attention_heads: int = 30
    attention_head_dim: int = 64
    in_channels: int = 16
    out_channels: int = 16
    time_embed_dim: int = 512
    text_embed_dim: int = 4096
    num_layers: int = 30
    sample_width: int = 90
    sample_height: int = 60
    sample_frames: int = 49
    patch_size: int = 2
    temporal_compression_ratio: int = 4
    max_text_seq_length: int = 226
    spatial_interpolation_scale: float = 1.875
    temporal_interpolation_scale: float = 1.0
    norm_eps: float = 1e-5
    attention_bias: bool = True


COGVIDEOX_2B = DiTConfig()                      # text-to-video 2B
COGVIDEOX_2B_I2V = DiTConfig(in_channels=32)    # the reference's image-to-video variant: latents + image latents
TINY_DIT = DiTConfig(num_attention_heads=2, in_channels=32, out_channels=16, time_embed_dim=64, num_layers=2, sample_width=12,
                     sample_height=8, sample_frames=9, max_text_seq_length=16)


# ------------------------------------------------------------------------------------------------ [EXT] embeddings.py
def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_3d_sincos_pos_embed(embed_dim, spatial_size, temporal_size, spatial_interpolation_scale=1.0,
                            temporal_interpolation_scale=1.0) -> np.ndarray:
    """-> [T, H*W, D]; spatial_size = (width, height); temporal quarter first, then (h half | w half) of the spatial 3/4"""
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    grid_h = np.arange(spatial_size[1], dtype=np.float32) / spatial_interpolation_scale
    grid_w = np.arange(spatial_size[0], dtype=np.float32) / spatial_interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, spatial_size[1], spatial_size[0]])
    spatial = np.concatenate([_sincos_1d(ds // 2, grid[0]), _sincos_1d(ds // 2, grid[1])], axis=1)
    temporal = _sincos_1d(dt, np.arange(temporal_size, dtype=np.float32) / temporal_interpolation_scale)
    spatial = np.repeat(spatial[np.newaxis], temporal_size, axis=0)
    temporal = np.repeat(temporal[:, np.newaxis], spatial_size[0] * spatial_size[1], axis=1)
    return np.concatenate([temporal, spatial], axis=-1)


class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, patch_size=2, patch_size_t=None, in_channels=16, embed_dim=1920, text_embed_dim=4096, bias=True,
                 sample_width=90, sample_height=60, sample_frames=49, temporal_compression_ratio=4,
                 max_text_seq_length=226, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0,
                 use_positional_embeddings=True, use_learned_positional_embeddings=False):
        super().__init__()
        assert patch_size_t is None and use_positional_embeddings and not use_learned_positional_embeddings
        self.patch_size, self.embed_dim = patch_size, embed_dim
        self.sample_height, self.sample_width, self.sample_frames = sample_height, sample_width, sample_frames
        self.temporal_compression_ratio, self.max_text_seq_length = temporal_compression_ratio, max_text_seq_length
        self.spatial_interpolation_scale, self.temporal_interpolation_scale = spatial_interpolation_scale, temporal_interpolation_scale
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=bias)
        self.text_proj = nn.Linear(text_embed_dim, embed_dim)
        self.register_buffer("pos_embedding", self._get_positional_embeddings(sample_height, sample_width, sample_frames),
                             persistent=False)

    def _get_positional_embeddings(self, sample_height, sample_width, sample_frames):
        h, w = sample_height // self.patch_size, sample_width // self.patch_size
        t = (sample_frames - 1) // self.temporal_compression_ratio + 1
        pe = torch.from_numpy(get_3d_sincos_pos_embed(self.embed_dim, (w, h), t, self.spatial_interpolation_scale,
                                                      self.temporal_interpolation_scale)).float().flatten(0, 1)
        joint = torch.zeros(1, self.max_text_seq_length + t * h * w, self.embed_dim)
        joint[:, self.max_text_seq_length:] = pe
        return joint

    def forward(self, text_embeds, image_embeds):
        text_embeds = self.text_proj(text_embeds)
        b, f, c, h, w = image_embeds.shape
        x = self.proj(image_embeds.reshape(-1, c, h, w))
        x = x.view(b, f, *x.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
        embeds = torch.cat([text_embeds, x], dim=1).contiguous()
        pre = (f - 1) * self.temporal_compression_ratio + 1
        if self.sample_height != h or self.sample_width != w or self.sample_frames != pre:
            pos = self._get_positional_embeddings(h, w, pre).to(embeds.device)
        else:
            pos = self.pos_embedding
        return embeds + pos.to(embeds.dtype)


# ------------------------------------------------------------------------------------------------ [EXT] normalization.py
class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim, embedding_dim, elementwise_affine=True, eps=1e-5, bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, hidden_states, encoder_hidden_states, temb):
        shift, scale, gate, enc_shift, enc_scale, enc_gate = self.linear(self.silu(temb)).chunk(6, dim=1)
        hidden_states = self.norm(hidden_states) * (1 + scale)[:, None, :] + shift[:, None, :]
        encoder_hidden_states = self.norm(encoder_hidden_states) * (1 + enc_scale)[:, None, :] + enc_shift[:, None, :]
        return hidden_states, encoder_hidden_states, gate[:, None, :], enc_gate[:, None, :]


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, output_dim=None, norm_elementwise_affine=False, norm_eps=1e-5,
                 chunk_dim=0):
        super().__init__()
        assert chunk_dim == 1 and num_embeddings is None
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)

    def forward(self, x, temb):
        shift, scale = self.linear(self.silu(temb)).chunk(2, dim=1)
        return self.norm(x) * (1 + scale[:, None, :]) + shift[:, None, :]


# ------------------------------------------------------------------------------------------------ [EXT] attention
class CogVideoXAttnProcessor2_0:
    """joint attention over [text | video] tokens with LayerNorm on the per-head queries and keys"""

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None):
        assert image_rotary_emb is None, "CogVideoX-2B uses the sin-cos table, not rotary embeddings"
        tl = encoder_hidden_states.size(1)
        x = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        b, s, _ = x.shape
        hd = attn.inner_dim // attn.heads
        q, k, v = (m(x).view(b, -1, attn.heads, hd).transpose(1, 2) for m in (attn.to_q, attn.to_k, attn.to_v))
        if attn.norm_q is not None:
            q = attn.norm_q(q)
        if attn.norm_k is not None:
            k = attn.norm_k(k)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, attn.heads * hd)
        o = attn.to_out[1](attn.to_out[0](o))
        enc, hid = o.split([tl, o.size(1) - tl], dim=1)
        return hid, enc


FusedCogVideoXAttnProcessor2_0 = CogVideoXAttnProcessor2_0


class Attention(nn.Module):
    def __init__(self, query_dim, dim_head=64, heads=8, qk_norm=None, eps=1e-5, bias=False, out_bias=True, processor=None,
                 **_):
        super().__init__()
        self.inner_dim, self.heads = dim_head * heads, heads
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.norm_q = nn.LayerNorm(dim_head, eps=eps) if qk_norm == "layer_norm" else None
        self.norm_k = nn.LayerNorm(dim_head, eps=eps) if qk_norm == "layer_norm" else None
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = processor if processor is not None else CogVideoXAttnProcessor2_0()

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states, attention_mask, **kw)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False, inner_dim=None,
                 bias=True):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        inner_dim = inner_dim or dim * mult
        layers = [GELU(dim, inner_dim, approximate="tanh", bias=bias), nn.Dropout(dropout),
                  nn.Linear(inner_dim, dim_out or dim, bias=bias)]
        if final_dropout:
            layers.append(nn.Dropout(dropout))
        self.net = nn.ModuleList(layers)

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


# ------------------------------------------------------------------------------------------------ in-tree: block + model
class CogVideoXBlock(nn.Module):
    """cogvideox_transformer_3d.py:41-160"""

    def __init__(self, dim, heads, head_dim, time_embed_dim, attention_bias=True, norm_eps=1e-5):
        super().__init__()
        self.norm1 = CogVideoXLayerNormZero(time_embed_dim, dim, True, norm_eps, bias=True)
        self.attn1 = Attention(query_dim=dim, dim_head=head_dim, heads=heads, qk_norm="layer_norm", eps=1e-6,
                               bias=attention_bias, out_bias=True)
        self.norm2 = CogVideoXLayerNormZero(time_embed_dim, dim, True, norm_eps, bias=True)
        self.ff = FeedForward(dim, activation_fn="gelu-approximate", final_dropout=True, bias=True)

    def forward(self, hidden_states, encoder_hidden_states, temb):
        tl = encoder_hidden_states.size(1)
        n, ne, g, eg = self.norm1(hidden_states, encoder_hidden_states, temb)
        a, ea = self.attn1(hidden_states=n, encoder_hidden_states=ne)
        hidden_states = hidden_states + g * a
        encoder_hidden_states = encoder_hidden_states + eg * ea
        n, ne, g, eg = self.norm2(hidden_states, encoder_hidden_states, temb)
        ff = self.ff(torch.cat([ne, n], dim=1))
        hidden_states = hidden_states + g * ff[:, tl:]
        encoder_hidden_states = encoder_hidden_states + eg * ff[:, :tl]
        return hidden_states, encoder_hidden_states


class CogVideoXTransformer3DModel(nn.Module):
    """cogvideox_transformer_3d.py:163-638 incl. ``init_quaternion_modules`` (:337-366, always built here)"""

    def __init__(self, cfg: DiTConfig = COGVIDEOX_2B_I2V):
        super().__init__()
        self.config = SimpleNamespace(**cfg.__dict__)
        d = cfg.num_attention_heads * cfg.attention_head_dim
        self.patch_embed = CogVideoXPatchEmbed(
            patch_size=cfg.patch_size, in_channels=cfg.in_channels, embed_dim=d, text_embed_dim=cfg.text_embed_dim,
            sample_width=cfg.sample_width, sample_height=cfg.sample_height, sample_frames=cfg.sample_frames,
            temporal_compression_ratio=cfg.temporal_compression_ratio, max_text_seq_length=cfg.max_text_seq_length,
            spatial_interpolation_scale=cfg.spatial_interpolation_scale,
            temporal_interpolation_scale=cfg.temporal_interpolation_scale)
        self.time_proj = Timesteps(d, True, 0)
        self.time_embedding = TimestepEmbedding(d, cfg.time_embed_dim)
        self.transformer_blocks = nn.ModuleList([
            CogVideoXBlock(d, cfg.num_attention_heads, cfg.attention_head_dim, cfg.time_embed_dim, cfg.attention_bias,
                           cfg.norm_eps) for _ in range(cfg.num_layers)])
        self.norm_final = nn.LayerNorm(d, cfg.norm_eps, True)
        self.norm_out = AdaLayerNorm(embedding_dim=cfg.time_embed_dim, output_dim=2 * d, norm_elementwise_affine=True,
                                     norm_eps=cfg.norm_eps, chunk_dim=1)
        self.proj_out = nn.Linear(d, cfg.patch_size * cfg.patch_size * cfg.out_channels)
        # init_quaternion_modules (:337-366)
        self.quaternion_lora_dconv = nn.Conv1d(1024, 256, 1, groups=256, bias=False)
        self.quaternion_lora_lconv = nn.Conv1d(4096, 256, 1, groups=256, bias=False)
        self.quaternion_lora_fconv = nn.Conv1d(1024, 256, 1, groups=256, bias=False)
        self.quaternion_lora_fuse = QuaternionLinearAutograd(1024, 512)
        self.quaternion_lora_fuse_fft_mag = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_pha = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_mag0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_fft_pha0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_sf = nn.Sequential(nn.Linear(1024, 512), nn.LeakyReLU(0.1, inplace=True),
                                                     nn.Linear(512, 4096))
        self.quaternion_lora_texts = nn.Parameter(torch.zeros(256))
        self.quaternion_lora_texts_fft_mag = nn.Parameter(torch.zeros(129))
        self.quaternion_lora_texts_fft_pha = nn.Parameter(torch.zeros(129))

    def lk_fuse(self, encoder_hidden_states, domain_features, flow_features):
        """:519-582 - the fused TEXT embeddings [B, L, 4096] that replace ``encoder_hidden_states``"""
        low = self.quaternion_lora_lconv(encoder_hidden_states.permute(0, 2, 1)).permute(0, 2, 1)
        d = F.interpolate(domain_features, size=1024, mode="linear")
        low_d = self.quaternion_lora_dconv(d.permute(0, 2, 1)).permute(0, 2, 1)
        f = F.interpolate(flow_features, size=1024, mode="linear")
        low_f = self.quaternion_lora_fconv(f.permute(0, 2, 1)).permute(0, 2, 1)
        low_d, low_f = low_d.expand_as(low), low_f.expand_as(low)
        ctx = self.quaternion_lora_texts.expand_as(low)
        spatial = self.quaternion_lora_fuse(torch.cat([low, low_d, low_f, ctx], dim=-1))
        hf, df, ff = (torch.fft.rfft(t, dim=-1) for t in (low, low_d, low_f))
        mags = [torch.abs(hf), torch.abs(df), torch.abs(ff), self.quaternion_lora_texts_fft_mag.expand_as(hf.real)]
        phas = [torch.angle(hf), torch.angle(df), torch.angle(ff), self.quaternion_lora_texts_fft_pha.expand_as(hf.real)]
        mag = self.quaternion_lora_fuse_fft_mag(torch.cat([m[..., :-1] for m in mags], dim=-1))
        pha = self.quaternion_lora_fuse_fft_pha(torch.cat([p[..., :-1] for p in phas], dim=-1))
        spec = torch.complex(mag * torch.cos(pha), mag * torch.sin(pha))
        mag0 = self.quaternion_lora_fuse_fft_mag0(torch.cat([m[..., -1:] for m in mags], dim=-1))
        pha0 = self.quaternion_lora_fuse_fft_pha0(torch.cat([p[..., -1:] for p in phas], dim=-1))
        spec = torch.cat([spec, torch.complex(mag0 * torch.cos(pha0), mag0 * torch.sin(pha0))], dim=-1)
        freq = torch.fft.irfft(spec, dim=-1)
        return self.quaternion_lora_fuse_sf(torch.cat([spatial, freq], dim=-1))

    def forward(self, hidden_states, encoder_hidden_states, timestep, domain_features, flow_features, return_dict=False):
        b, f, c, h, w = hidden_states.shape
        emb = self.time_embedding(self.time_proj(timestep).to(hidden_states.dtype))
        encoder_hidden_states = self.lk_fuse(encoder_hidden_states, domain_features, flow_features)
        x = self.patch_embed(encoder_hidden_states, hidden_states)
        tl = encoder_hidden_states.shape[1]
        enc, hid = x[:, :tl], x[:, tl:]
        for blk in self.transformer_blocks:
            hid, enc = blk(hid, enc, emb)
        hid = self.norm_final(hid)
        hid = self.proj_out(self.norm_out(hid, temb=emb))
        p = self.config.patch_size
        out = hid.reshape(b, f, h // p, w // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        return (out,)


# ------------------------------------------------------------------------------------------------ [EXT] scheduler + loop
class CogVideoXDDIMScheduler:
    """[EXT diffusers scheduling_ddim_cogvideox.py] with CogVideoX-2B's scheduler_config.json: scaled-linear betas 0.00085..0.012,
    snr_shift_scale 3.0, zero-terminal-SNR rescale, trailing spacing, v-prediction, set_alpha_to_one"""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=3.0):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        s = ac.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * (s0 / (s0 - sT))
        self.alphas_cumprod = s ** 2
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64)
        self.num_train_timesteps = num_train_timesteps
        self.init_noise_sigma = 1.0
        self.order = 1

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps / n
        self.timesteps = torch.from_numpy(np.round(np.arange(self.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1)

    def coefficients(self, t: int):
        """(a_t, b_t, sqrt(alpha_t), sqrt(1 - alpha_t)): prev = a_t * sample + b_t * x0, x0 = sqrt(alpha) x - sqrt(1-alpha) v"""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t_ = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        a = ((1 - a_prev) / (1 - a_t_)) ** 0.5
        b = a_prev ** 0.5 - a_t_ ** 0.5 * a
        return float(a), float(b), float(a_t_ ** 0.5), float((1 - a_t_) ** 0.5)

    def step(self, model_output, t: int, sample):
        a, b, sa, sb = self.coefficients(int(t))
        x0 = sa * sample - sb * model_output
        return a * sample + b * x0


def dynamic_guidance(guidance_scale: float, num_inference_steps: int, t: int) -> float:
    """pipeline_cogvideox_image2video.py:866-869"""
    return 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2)


@torch.no_grad()
def denoise(model, scheduler, latents, image_latents, prompt_embeds, domain_features, flow_features, num_inference_steps,
            guidance_scale=6.0, use_dynamic_cfg=True, callback=None):
    """pipeline_cogvideox_image2video.py:829-885: latents [B, F, C, h, w] (init_noise_sigma = 1), image_latents [B, F, C, h, w],
    prompt_embeds [2B, L, 4096] (negative first)"""
    scheduler.set_timesteps(num_inference_steps)
    cfg = guidance_scale > 1.0
    for i, t in enumerate(scheduler.timesteps.tolist()):
        x = torch.cat([latents] * 2) if cfg else latents
        img = torch.cat([image_latents] * 2) if cfg else image_latents
        x = torch.cat([x, img], dim=2)
        ts = torch.full((x.shape[0],), t, dtype=torch.long)
        noise = model(x, prompt_embeds, ts, domain_features, flow_features)[0].float()
        g = dynamic_guidance(guidance_scale, num_inference_steps, t) if use_dynamic_cfg else guidance_scale
        if cfg:
            u, c = noise.chunk(2)
            noise = u + g * (c - u)
        latents = scheduler.step(noise, t, latents)
        if callback is not None:
            callback(i, t, latents)
    return latents


def init_weights_(m: nn.Module, seed: int) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(m.named_parameters()):
            if ("norm" in name) and name.endswith("weight") and p.ndim == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.startswith("quaternion_lora_texts"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif "quaternion" in name and p.ndim == 2 and "weight" in name and not name.endswith(".weight"):
                p.copy_(torch.randn(p.shape, generator=g) / (4 * p.shape[0]) ** 0.5)
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m
