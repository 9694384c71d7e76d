"""The CogVideoX DiT sampling loop with LKGD's latent-knowledge fuse on the MI355X path (BASELINE.json
configs[4]).

Mirrors /root/reference/CogVideo-main/finetune/models/cogvideox_i2v/cogvideox_transformer_3d.py (``CogVideoXBlock`` :41-160,
``CogVideoXTransformer3DModel`` :163-335, ``init_quaternion_modules`` :337-366, ``forward`` :473-638 with the same positional
``domain_features`` / ``flow_features``) and the loop of ``pipeline_cogvideox_image2video.py:829-885`` (CFG duplication, channel
concat with the image latents, dynamic CFG scale :866-869, scheduler step).  The blocks those files import from diffusers >= 0.32
[EXT] are restated with diffusers' parameter names (a ``transformer/`` checkpoint loads by ``load_state_dict``); oracle/cogvideox.py
is the fp32 twin the tests compare with, and tests/golden/cogvideox.safetensors pins the in-tree wiring on the reference's own
``forward``.

MI355X design - the UNet's kernels, one joint token buffer:
* tokens [B * (L_text + L_video), D] fp16, text rows first in every batch entry (the order of the reference's
  ``torch.cat([encoder_hidden_states, hidden_states], dim=1)`` for attention and feed-forward): the two streams are row
  slices, never concatenated or split;
* adaLN-zero: the 6 x D modulation vectors of all 2 x 30 ``CogVideoXLayerNormZero`` layers come from ONE GEMM per step
  ([sum, 512] weights on silu(emb)); ``norm(x) * (1 + scale) + shift`` is the LayerNorm kernel with per-(batch, stream)
  effective affine vectors gamma (1 + scale), beta (1 + scale) + shift (rows up to 2048 channels); the gated residuals of both
  streams are one pass of ``lkgd_gated_add``;
* attention: three projections (bias), per-head LayerNorm of q and k as the LayerNorm kernel over [tokens * heads, 64] rows in
  place, then the head_dim-64 flash kernel over the joint sequence (S = 226 + 17 550, ragged last tile);
* feed-forward: GEMM + ``lkgd_gelu_tanh`` + GEMM;
* the latent-knowledge fuse acts on the TEXT embeddings and is step-invariant: evaluated once per clip in fp32 (as the SVD
  fuse, lkgd_amd/lk_fuse.py); the 3-D sin-cos position table is added in the patch-embedding GEMM's epilogue (row-indexed bias);
* patch unfold / un-patchify at the API edge, CFG combine and the DDIM update on the 1-M-element latents are tensor plumbing
  (PyTorch-ROCm elementwise ops, < 0.1 % of a step).
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

from . import ops
from ._lib import LkgdHipError
from .lk_fuse import hamilton
from .packing import pack_linear
from .unet import QuaternionLinearAutograd, TimestepEmbedding


@dataclass
class DiTConfig:
    """constructor keywords of CogVideoXTransformer3DModel (cogvideox_transformer_3d.py:224-255) used by the 2B models"""
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


def _f32(p):
    return p.detach().to(torch.float32).contiguous()


def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed_3d(embed_dim, width, height, frames, spatial_scale, temporal_scale) -> torch.Tensor:
    """[EXT diffusers embeddings.py get_3d_sincos_pos_embed] -> [frames * height * width, D]: temporal quarter, then the
    (h half | w half) of the spatial three quarters"""
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    gh = np.arange(height, dtype=np.float32) / spatial_scale
    gw = np.arange(width, dtype=np.float32) / spatial_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, height, width])
    spatial = np.concatenate([_sincos_1d(ds // 2, grid[0]), _sincos_1d(ds // 2, grid[1])], axis=1)
    temporal = _sincos_1d(dt, np.arange(frames, dtype=np.float32) / temporal_scale)
    pe = np.concatenate([np.repeat(temporal[:, None], height * width, axis=1), np.repeat(spatial[None], frames, axis=0)], axis=-1)
    return torch.from_numpy(pe).float().flatten(0, 1)


# ------------------------------------------------------------------------------------------------ parameter holders
class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim, embedding_dim, eps):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps)


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, output_dim, eps):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, eps)


class Attention(nn.Module):
    def __init__(self, dim, heads, head_dim, bias):
        super().__init__()
        if head_dim != 64:
            raise LkgdHipError("the attention kernel is built for head_dim 64")
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=bias)
        self.to_k = nn.Linear(dim, dim, bias=bias)
        self.to_v = nn.Linear(dim, dim, bias=bias)
        self.norm_q = nn.LayerNorm(head_dim, eps=1e-6)
        self.norm_k = nn.LayerNorm(head_dim, eps=1e-6)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GELU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim), nn.Dropout(0.0)])


class CogVideoXBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, time_embed_dim, attention_bias, eps):
        super().__init__()
        self.norm1 = CogVideoXLayerNormZero(time_embed_dim, dim, eps)
        self.attn1 = Attention(dim, heads, head_dim, attention_bias)
        self.norm2 = CogVideoXLayerNormZero(time_embed_dim, dim, eps)
        self.ff = FeedForward(dim)

    def pack(self):
        a, f = self.attn1, self.ff

        def lin(m):
            return pack_linear(m.weight), (_f32(m.bias) if m.bias is not None else None)
        self._pk = SimpleNamespace(q=lin(a.to_q), k=lin(a.to_k), v=lin(a.to_v), o=lin(a.to_out[0]),
                                   nq=(_f32(a.norm_q.weight), _f32(a.norm_q.bias)), nk=(_f32(a.norm_k.weight), _f32(a.norm_k.bias)),
                                   f1=lin(f.net[0].proj), f2=lin(f.net[2]))


class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, cfg: DiTConfig, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(cfg.in_channels, dim, kernel_size=(cfg.patch_size, cfg.patch_size), stride=cfg.patch_size)
        self.text_proj = nn.Linear(cfg.text_embed_dim, dim)


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor


class CogVideoXTransformer3DModel(nn.Module):
    def __init__(self, config: Optional[DiTConfig] = None, **kw):
        super().__init__()
        cfg = config if config is not None else DiTConfig(**kw)
        self.config = SimpleNamespace(**cfg.__dict__, patch_size_t=None, use_rotary_positional_embeddings=False,
                                      ofs_embed_dim=None)
        d = cfg.num_attention_heads * cfg.attention_head_dim
        if d % 64 or (cfg.in_channels * cfg.patch_size ** 2) % 64 or cfg.time_embed_dim % 64 or cfg.text_embed_dim % 64 \
                or d > 2048 or (cfg.patch_size ** 2 * cfg.out_channels) % 8:
            raise LkgdHipError("DiT on the HIP path: dims multiples of 64 (K granularity), inner dim <= 2048")
        self.inner_dim = d
        self.patch_embed = CogVideoXPatchEmbed(cfg, d)
        self.time_embedding = TimestepEmbedding(d, cfg.time_embed_dim)
        self.transformer_blocks = nn.ModuleList([
            CogVideoXBlock(d, cfg.num_attention_heads, cfg.attention_head_dim, cfg.time_embed_dim, cfg.attention_bias,
                           cfg.norm_eps) for _ in range(cfg.num_layers)])
        self.norm_final = nn.LayerNorm(d, cfg.norm_eps)
        self.norm_out = AdaLayerNorm(cfg.time_embed_dim, 2 * d, cfg.norm_eps)
        self.proj_out = nn.Linear(d, cfg.patch_size * cfg.patch_size * cfg.out_channels)
        self.init_quaternion_modules()
        self._pk = None
        self._pos = {}

    def init_quaternion_modules(self):
        """cogvideox_transformer_3d.py:337-366 (the reference calls it after construction; here the modules always exist)"""
        self.quaternion_lora_dconv = nn.Conv1d(1024, 256, 1, groups=256, bias=False)
        self.quaternion_lora_lconv = nn.Conv1d(4096, 256, 1, groups=256, bias=False)
        self.quaternion_lora_fconv = nn.Conv1d(1024, 256, 1, groups=256, bias=False)
        self.quaternion_lora_fuse = QuaternionLinearAutograd(1024, 512)
        self.quaternion_lora_fuse_fft_mag = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_pha = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_mag0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_fft_pha0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_sf = nn.Sequential(nn.Linear(1024, 512), nn.LeakyReLU(0.1, inplace=True), nn.Linear(512, 4096))
        self.quaternion_lora_texts = nn.Parameter(torch.zeros(256))
        self.quaternion_lora_texts_fft_mag = nn.Parameter(torch.zeros(129))
        self.quaternion_lora_texts_fft_pha = nn.Parameter(torch.zeros(129))

    # ---- bookkeeping -------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.proj_out.weight.device

    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    def invalidate(self):
        self._pk = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._pk = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pk = None
        return r

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, torch_dtype=None,
                        variant: Optional[str] = None, **_ignored):
        from .loading import build_from_pretrained, load_config, load_state_dict
        raw = load_config(pretrained_model_name_or_path, subfolder)
        for key in ("use_rotary_positional_embeddings", "patch_size_t", "ofs_embed_dim"):
            if raw.get(key):        # the 5B / 1.5 variants (rotary embeddings, temporal patches, ofs embedding) are not built
                raise LkgdHipError(f"CogVideoXTransformer3DModel.from_pretrained: config has {key}={raw[key]!r}; only the "
                                   "2B architecture (learned positional embedding, 2-D patches) is implemented")
        # stock checkpoints have no quaternion_lora_* modules: those may be missing; anything else missing or unexpected
        # (a truncated shard, renamed parameters) would leave meta-initialised garbage behind a non-strict load
        sd_keys = set(load_state_dict(pretrained_model_name_or_path, subfolder, variant).keys())
        m = build_from_pretrained(cls, DiTConfig, pretrained_model_name_or_path, subfolder, torch_dtype, variant, strict=False)
        own = set(m.state_dict().keys())
        missing = sorted(k for k in own - sd_keys if not k.startswith("quaternion_lora_"))
        unexpected = sorted(sd_keys - own)
        if missing or unexpected:
            raise RuntimeError(f"CogVideoXTransformer3DModel.from_pretrained({pretrained_model_name_or_path!r}): missing keys "
                               f"{missing[:5]}{'...' if len(missing) > 5 else ''}, unexpected keys {unexpected[:5]}"
                               f"{'...' if len(unexpected) > 5 else ''}")
        return m

    def save_pretrained(self, save_directory: str, variant: Optional[str] = None, **_ignored):
        from .loading import save_pretrained
        cfg = {k: v for k, v in self.config.__dict__.items() if k in DiTConfig.__dataclass_fields__}
        save_pretrained(self, save_directory, cfg, type(self).__name__, variant)

    @torch.no_grad()
    def prepare(self):
        if self._pk is not None:
            return
        if self.device.type != "cuda":
            raise LkgdHipError("lkgd_amd DiT runs on MI355X only: move the module to cuda first")
        for b in self.transformer_blocks:
            b.pack()
        self.time_embedding.pack()
        pk = SimpleNamespace()
        # every block's two modulation linears as ONE [blocks * 2 * 6D, Te] GEMM per step (+ norm_out's [2D, Te])
        mods = [m for b in self.transformer_blocks for m in (b.norm1.linear, b.norm2.linear)] + [self.norm_out.linear]
        pk.w_mod = torch.cat([pack_linear(m.weight) for m in mods], dim=0).contiguous()
        pk.b_mod = torch.cat([_f32(m.bias) for m in mods]).contiguous()
        nb = len(self.transformer_blocks)
        pk.ln_g = torch.stack([torch.stack([_f32(b.norm1.norm.weight), _f32(b.norm2.norm.weight)]) for b in self.transformer_blocks])
        pk.ln_b = torch.stack([torch.stack([_f32(b.norm1.norm.bias), _f32(b.norm2.norm.bias)]) for b in self.transformer_blocks])
        pk.nb = nb
        pk.fin = (_f32(self.norm_final.weight), _f32(self.norm_final.bias))
        pk.out_g, pk.out_b = _f32(self.norm_out.norm.weight), _f32(self.norm_out.norm.bias)
        pe = self.patch_embed
        pk.w_pe, pk.b_pe = pack_linear(pe.proj.weight.detach()), _f32(pe.proj.bias)          # [D, C*p*p], k = (c, ky, kx)
        pk.w_tx, pk.b_tx = pack_linear(pe.text_proj.weight), _f32(pe.text_proj.bias)
        pk.w_po, pk.b_po = pack_linear(self.proj_out.weight), _f32(self.proj_out.bias)
        self._pk = pk
        self._pos = {}

    def _pos_table(self, f: int, h: int, w: int) -> torch.Tensor:
        key = (f, h, w)
        t = self._pos.get(key)
        if t is None:
            c = self.config
            t = sincos_pos_embed_3d(self.inner_dim, w, h, f, c.spatial_interpolation_scale, c.temporal_interpolation_scale)
            t = self._pos[key] = t.to(device=self.device, dtype=torch.float16).contiguous()
        return t

    # ---- latent-knowledge fuse on the text embeddings (:519-582), once per clip -----------------------------------
    @torch.no_grad()
    def fused_text(self, encoder_hidden_states, domain_features, flow_features) -> torch.Tensor:
        dev = self.device
        e = encoder_hidden_states.to(device=dev, dtype=torch.float32)

        def dw(conv, x, per):       # Conv1d(k=1, groups=256) on the channel axis: `per` inputs per group
            w = conv.weight.detach().float().reshape(256, per)
            return (x.reshape(*x.shape[:-1], 256, per) * w).sum(-1)

        def qlin(q, x):
            return x @ hamilton(q) + q.bias.detach().float()
        low = dw(self.quaternion_lora_lconv, e, 16)
        d = F.interpolate(domain_features.to(device=dev, dtype=torch.float32), size=1024, mode="linear")
        f = F.interpolate(flow_features.to(device=dev, dtype=torch.float32), size=1024, mode="linear")
        low_d = dw(self.quaternion_lora_dconv, d, 4).expand_as(low)
        low_f = dw(self.quaternion_lora_fconv, f, 4).expand_as(low)
        ctx = self.quaternion_lora_texts.detach().float().expand_as(low)
        spatial = qlin(self.quaternion_lora_fuse, torch.cat([low, low_d, low_f, ctx], -1))
        hf, df, ff = (torch.fft.rfft(t.contiguous(), dim=-1) for t in (low, low_d, low_f))
        mags = [hf.abs(), df.abs(), ff.abs(), self.quaternion_lora_texts_fft_mag.detach().float().expand_as(hf.real)]
        phas = [hf.angle(), df.angle(), ff.angle(), self.quaternion_lora_texts_fft_pha.detach().float().expand_as(hf.real)]
        mag = qlin(self.quaternion_lora_fuse_fft_mag, torch.cat([m[..., :-1] for m in mags], -1))
        pha = qlin(self.quaternion_lora_fuse_fft_pha, torch.cat([p[..., :-1] for p in phas], -1))
        l0m, l0p = self.quaternion_lora_fuse_fft_mag0, self.quaternion_lora_fuse_fft_pha0
        mag0 = torch.cat([m[..., -1:] for m in mags], -1) @ l0m.weight.detach().float().T + l0m.bias.detach().float()
        pha0 = torch.cat([p[..., -1:] for p in phas], -1) @ l0p.weight.detach().float().T + l0p.bias.detach().float()
        spec = torch.cat([torch.complex(mag * torch.cos(pha), mag * torch.sin(pha)),
                          torch.complex(mag0 * torch.cos(pha0), mag0 * torch.sin(pha0))], -1)
        freq = torch.fft.irfft(spec, dim=-1)
        sf = self.quaternion_lora_fuse_sf
        x = torch.cat([spatial, freq], -1)
        x = F.leaky_relu(x @ sf[0].weight.detach().float().T + sf[0].bias.detach().float(), 0.1)
        x = x @ sf[2].weight.detach().float().T + sf[2].bias.detach().float()
        return x.to(torch.float16)

    # ---- the per-step forward ------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_tokens(self, hidden_states: torch.Tensor, fused_text: torch.Tensor, timestep, shard=None) -> torch.Tensor:
        """hidden_states [B, F, C, h, w]; fused_text [B, L, 4096] fp16 (``fused_text`` of the prompt embeddings) ->
        [B, F, out_channels, h, w] fp16.  ``shard`` (lkgd_amd.dist_run.ShardInfo): this rank holds ONE batch entry (its CFG
        half) and the latent frames [f0, f0 + F) of the clip; everything is row-local except the attention, whose local
        queries (the replicated text rows + the rank's video rows) attend to the keys / values of ALL frames, all-gathered
        over the frame group every layer."""
        self.prepare()
        pk, cfg, dev = self._pk, self.config, self.device
        B, Fr, C_, H, W = hidden_states.shape
        if shard is not None and B != 1:
            raise LkgdHipError("frame sharding of the DiT supports one batch entry per rank")
        p, D = cfg.patch_size, self.inner_dim
        h, w = H // p, W // p
        Tt, Tv = fused_text.shape[1], Fr * h * w
        L = Tt + Tv
        heads = cfg.num_attention_heads
        # time embedding -> all modulation vectors of the step (one GEMM)
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        t = t.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        emb = self.time_embedding.run(ops.timestep_embedding(t, D))
        semb = ops.silu(emb)
        mod = torch.empty(B, pk.w_mod.shape[0], dtype=torch.float16, device=dev)
        ops.gemm(semb, pk.w_mod, mod, M=B, N=pk.w_mod.shape[0], K=pk.w_mod.shape[1], bias=pk.b_mod)
        mod = mod.float()
        nb = pk.nb
        blk = mod[:, :nb * 12 * D].reshape(B, nb, 2, 6, D)            # (shift, scale, gate, enc_shift, enc_scale, enc_gate)
        g, be = pk.ln_g[None], pk.ln_b[None]                          # [1, nb, 2, D]
        # effective affine of norm(x) * (1 + scale) + shift, [B, nb, 2, stream (0 text, 1 video), D]
        eff_g = torch.stack([g * (1 + blk[:, :, :, 4]), g * (1 + blk[:, :, :, 1])], dim=3).contiguous()
        eff_b = torch.stack([be * (1 + blk[:, :, :, 4]) + blk[:, :, :, 3], be * (1 + blk[:, :, :, 1]) + blk[:, :, :, 0]], dim=3).contiguous()
        gates = torch.stack([blk[:, :, :, 5], blk[:, :, :, 2]], dim=3).permute(1, 2, 0, 3, 4).contiguous()   # [nb, 2, B, stream, D]
        fin = mod[:, nb * 12 * D:].reshape(B, 2, D)                   # norm_out: (shift, scale)
        # patch embedding into the joint buffer: text rows, then video rows (+ position table in the epilogue)
        X = torch.empty(B * L, D, dtype=torch.float16, device=dev)
        txt = fused_text.to(device=dev, dtype=torch.float16).reshape(B * Tt, -1).contiguous()
        xh = hidden_states.to(device=dev, dtype=torch.float16)
        patches = xh.reshape(B, Fr, C_, h, p, w, p).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, Tv, C_ * p * p).contiguous()
        if shard is None:
            pos = self._pos_table(Fr, h, w)
        else:
            pos = self._pos_table(shard.F_total, h, w)[shard.f0 * h * w:(shard.f0 + Fr) * h * w]
            Tv_all = shard.F_total * h * w
            KV = [torch.empty(Tt + Tv_all, D, dtype=torch.float16, device=dev) for _ in range(2)]
        for b in range(B):
            ops.gemm(txt[b * Tt:(b + 1) * Tt], pk.w_tx, X[b * L:b * L + Tt], M=Tt, N=D, K=txt.shape[1], bias=pk.b_tx)
            ops.gemm(patches[b], pk.w_pe, X[b * L + Tt:(b + 1) * L], M=Tv, N=D, K=C_ * p * p, bias=pk.b_pe, rowbias=pos,
                     rowmap=(1, 1, 1, 1 << 30))
        T = B * L
        eps = cfg.norm_eps

        def modnorm(i, which):
            n = torch.empty_like(X)
            for b in range(B):
                r0 = b * L
                ops.layernorm(X[r0:r0 + Tt], eff_g[b, i, which, 0], eff_b[b, i, which, 0], eps, out=n[r0:r0 + Tt])
                ops.layernorm(X[r0 + Tt:r0 + L], eff_g[b, i, which, 1], eff_b[b, i, which, 1], eps, out=n[r0 + Tt:r0 + L])
            return n
        for i, blkm in enumerate(self.transformer_blocks):
            bp = blkm._pk
            n = modnorm(i, 0)
            q, k, v = (torch.empty(T, D, dtype=torch.float16, device=dev) for _ in range(3))
            for dst, (wgt, bias) in ((q, bp.q), (k, bp.k), (v, bp.v)):
                ops.gemm(n, wgt, dst, M=T, N=D, K=D, bias=bias)
            ops.layernorm(q.view(T * heads, 64), bp.nq[0], bp.nq[1], 1e-6, out=q.view(T * heads, 64))     # per-head qk norm
            ops.layernorm(k.view(T * heads, 64), bp.nk[0], bp.nk[1], 1e-6, out=k.view(T * heads, 64))
            a = torch.empty(T, D, dtype=torch.float16, device=dev)
            if shard is None:
                ops.attn_spatial(q, k, v, a, B, L, heads)
            else:                   # keys / values of every frame of this CFG half: text rows (replicated) + gathered video rows
                for full, loc in zip(KV, (k, v)):
                    full[:Tt].copy_(loc[:Tt])
                    full[Tt:].copy_(shard.gather(loc[Tt:]))
                ops.attn_spatial(q, KV[0], KV[1], a, 1, Tt + Tv_all, heads, Sq=L)
            o = torch.empty(T, D, dtype=torch.float16, device=dev)
            ops.gemm(a, bp.o[0], o, M=T, N=D, K=D, bias=bp.o[1])
            X = ops.gated_add(o, gates[i, 0].reshape(2 * B, D), X, L, Tt)
            n = modnorm(i, 1)
            hdn = torch.empty(T, 4 * D, dtype=torch.float16, device=dev)
            ops.gemm(n, bp.f1[0], hdn, M=T, N=4 * D, K=D, bias=bp.f1[1])
            ops.gelu_tanh_(hdn)
            ops.gemm(hdn, bp.f2[0], o, M=T, N=D, K=4 * D, bias=bp.f2[1])
            X = ops.gated_add(o, gates[i, 1].reshape(2 * B, D), X, L, Tt)
        # norm_final on the video stream, norm_out (adaLN), proj_out, un-patchify
        po = pk.w_po.shape[0]
        out_tok = torch.empty(B, Tv, po, dtype=torch.float16, device=dev)
        for b in range(B):
            vid = X[b * L + Tt:(b + 1) * L]
            y = ops.layernorm(vid, pk.fin[0], pk.fin[1], eps)
            gg = (pk.out_g * (1 + fin[b, 1])).contiguous()
            bb = (pk.out_b * (1 + fin[b, 1]) + fin[b, 0]).contiguous()
            y = ops.layernorm(y, gg, bb, eps)
            ops.gemm(y, pk.w_po, out_tok[b], M=Tv, N=po, K=D, bias=pk.b_po)
        out = out_tok.reshape(B, Fr, h, w, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        return out.contiguous()

    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, domain_features, flow_features, timestep_cond=None,
                ofs=None, image_rotary_emb=None, attention_kwargs=None, return_dict: bool = True):
        """cogvideox_transformer_3d.py:473-486 - ``domain_features`` / ``flow_features`` are positional"""
        if image_rotary_emb is not None or ofs is not None or timestep_cond is not None:
            raise LkgdHipError("rotary embeddings / ofs / timestep_cond belong to the 5B and 1.5 models, not to CogVideoX-2B")
        text = self.fused_text(encoder_hidden_states, domain_features, flow_features)
        out = self.forward_tokens(hidden_states, text, timestep)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)


# ------------------------------------------------------------------------------------------------ scheduler + loop
class CogVideoXDDIMScheduler:
    """[EXT diffusers scheduling_ddim_cogvideox.py] with CogVideoX-2B's scheduler_config.json (scaled-linear betas
    0.00085..0.012, snr_shift_scale 3.0, zero-terminal-SNR rescale, trailing spacing, v-prediction, set_alpha_to_one); host
    tables in fp64, the update itself runs on the device"""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=3.0, **_ignored):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        s = ac.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        self.alphas_cumprod = ((s - sT) * (s0 / (s0 - sT))) ** 2
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type="v_prediction",
                                      timestep_spacing="trailing", snr_shift_scale=snr_shift_scale)
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.config.num_train_timesteps
        self.timesteps = torch.from_numpy(np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coefficients(self, t: int):
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        at = self.alphas_cumprod[t]
        ap = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        a = ((1 - ap) / (1 - at)) ** 0.5
        b = ap ** 0.5 - at ** 0.5 * a
        return float(a), float(b), float(at ** 0.5), float((1 - at) ** 0.5)

    def step(self, model_output, timestep, sample, **_):
        a, b, sa, sb = self.coefficients(int(timestep))
        x0 = sa * sample - sb * model_output
        return (a * sample + b * x0,)


def dynamic_guidance(guidance_scale: float, num_inference_steps: int, t: int) -> float:
    """pipeline_cogvideox_image2video.py:866-869"""
    return 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2)


@torch.no_grad()
def denoise(transformer: CogVideoXTransformer3DModel, scheduler: CogVideoXDDIMScheduler, latents, image_latents, prompt_embeds,
            domain_features, flow_features, num_inference_steps: int = 50, guidance_scale: float = 6.0,
            use_dynamic_cfg: bool = True, callback=None) -> torch.Tensor:
    """the loop of pipeline_cogvideox_image2video.py:829-885.  latents / image_latents [B, F, C, h, w]; prompt_embeds
    [2B, L, 4096] (negative first) when guidance_scale > 1.  The latents stay fp32 between steps (``noise_pred.float()`` :863,
    the reference casts them back to the prompt dtype :881 - reproduced)."""
    dev = transformer.device
    scheduler.set_timesteps(num_inference_steps)
    cfg = guidance_scale > 1.0
    text = transformer.fused_text(prompt_embeds, domain_features, flow_features)          # step-invariant: once per clip
    latents = latents.to(device=dev, dtype=torch.float16)
    img = image_latents.to(device=dev, dtype=torch.float16)
    img2 = torch.cat([img] * 2) if cfg else img
    for i, t in enumerate(scheduler.timesteps.tolist()):
        x = torch.cat([latents] * 2) if cfg else latents
        x = torch.cat([x, img2], dim=2)
        noise = transformer.forward_tokens(x, text, float(t)).float()
        g = dynamic_guidance(guidance_scale, num_inference_steps, t) if use_dynamic_cfg else guidance_scale
        if cfg:
            u, c = noise.chunk(2)
            noise = u + g * (c - u)
        latents = scheduler.step(noise, t, latents.float())[0].to(torch.float16)
        if callback is not None:
            callback(i, t, latents)
    return latents
