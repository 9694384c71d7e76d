"""LKGD latent-knowledge fuse, hoisted out of the sampling loop.

Reference: /root/reference/models/unet_spatio_temporal_condition.py:536-595 recomputes this block in every UNet
forward although its inputs (CLIP embedding, domain-ViT logits, flow-ViT logits) are the same for all 25 steps
(SURVEY.md finding 6).  It is ~2 MFLOP of fp32 work on [B,1,1024] vectors (grouped 1x1 conv taps, quaternion linears,
a 256-point real FFT and a 512-point inverse) - pure launch latency, not a kernel-worthy hot spot - so here it runs
ONCE per clip (`pipeline.denoise`) with PyTorch-ROCm fp32 tensor ops on the GPU (hipFFT has no half support; the reference
would fail in fp16 at torch.fft.rfft as well); `unet.forward` callers that pass the same tensor objects hit a one-entry cache.  The per-step UNet forward never touches it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def hamilton(q) -> torch.Tensor:
    """core_qnn quaternion_linear weight: [in, out] block matrix from r/i/j/k [in/4, out/4]"""
    r, i, j, k = (w.detach().float() for w in (q.r_weight, q.i_weight, q.j_weight, q.k_weight))
    return torch.cat([torch.cat([r, -i, -j, -k], 0), torch.cat([i, r, -k, j], 0),
                      torch.cat([j, k, r, -i], 0), torch.cat([k, -j, i, r], 0)], 1)


def _qlin(q, x):
    return x @ hamilton(q) + q.bias.detach().float()


def _dw(conv, x):
    """Conv1d(1024 -> 256, k=1, groups=256) on [B,1,1024] viewed as channels: 4-tap weighted sums"""
    w = conv.weight.detach().float().reshape(256, 4)           # [out, 4 inputs per group]
    return (x.reshape(x.shape[0], 256, 4) * w[None]).sum(-1)[:, None, :]   # [B,1,256]


@torch.no_grad()
def lk_fuse(unet, encoder_hidden_states, domain_features, flow_features) -> torch.Tensor:
    dev = unet.device
    e = encoder_hidden_states.to(device=dev, dtype=torch.float32)
    d = F.interpolate(domain_features.to(device=dev, dtype=torch.float32), size=1024, mode="linear")
    f = F.interpolate(flow_features.to(device=dev, dtype=torch.float32), size=1024, mode="linear")
    low, low_d, low_f = _dw(unet.quaternion_lora_lconv, e), _dw(unet.quaternion_lora_dconv, d), \
        _dw(unet.quaternion_lora_fconv, f)
    if low_d.shape[0] != low.shape[0] and low_d.shape[0] == 1:      # reference :544-546 (1 -> 2 only)
        low_d = torch.cat([low_d, low_d], 0)
        low_f = torch.cat([low_f, low_f], 0)
    ctx = unet.quaternion_lora_texts.detach().float().expand_as(low)
    spatial = _qlin(unet.quaternion_lora_fuse, torch.cat([low, low_d, low_f, ctx], -1))
    hf, df, ff = (torch.fft.rfft(t, dim=-1) for t in (low, low_d, low_f))
    mags = [torch.abs(hf), torch.abs(df), torch.abs(ff),
            unet.quaternion_lora_texts_fft_mag.detach().float().expand_as(hf.real)]
    phas = [torch.angle(hf), torch.angle(df), torch.angle(ff),
            unet.quaternion_lora_texts_fft_pha.detach().float().expand_as(hf.real)]
    mag = _qlin(unet.quaternion_lora_fuse_fft_mag, torch.cat([m[..., :-1] for m in mags], -1))
    pha = _qlin(unet.quaternion_lora_fuse_fft_pha, torch.cat([p[..., :-1] for p in phas], -1))
    spec = torch.complex(mag * torch.cos(pha), mag * torch.sin(pha))
    l0m, l0p = unet.quaternion_lora_fuse_fft_mag0, unet.quaternion_lora_fuse_fft_pha0
    mag0 = torch.cat([m[..., -1] for m in mags], -1) @ l0m.weight.detach().float().T + l0m.bias.detach().float()
    pha0 = torch.cat([p[..., -1] for p in phas], -1) @ l0p.weight.detach().float().T + l0p.bias.detach().float()
    spec0 = torch.complex(mag0 * torch.cos(pha0), mag0 * torch.sin(pha0))
    spec = torch.cat([spec, spec0.unsqueeze(-1)], -1)                 # 257 bins -> irfft length 512
    freq = torch.fft.irfft(spec, dim=-1)
    sf = unet.quaternion_lora_fuse_sf
    x = torch.cat([spatial, freq], -1)
    x = F.leaky_relu(x @ sf[0].weight.detach().float().T + sf[0].bias.detach().float(), 0.1)
    x = x @ sf[2].weight.detach().float().T + sf[2].bias.detach().float()
    return x.to(torch.float16)                                         # REPLACES the CLIP embedding (:595,:613)


def lk_fuse_cached(unet, e, d, f) -> torch.Tensor:
    """one-entry cache for callers that pass the SAME tensor objects every Euler step (`unet.forward` in a caller-owned
    loop).  The entry keeps the three input tensors alive and compares identity + in-place version, so the allocator
    cannot hand a recycled address of a different clip back as a hit; fresh tensors always recompute."""
    c = unet._lk_cache
    ins = (e, d, f)
    if c is not None and all(a is b and a._version == v for a, (b, v) in zip(ins, c[0])):
        return c[1]
    out = lk_fuse(unet, e, d, f)
    unet._lk_cache = (tuple((t, t._version) for t in ins), out)
    return out
