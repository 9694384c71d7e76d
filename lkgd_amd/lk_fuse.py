"""LKGD latent-knowledge fuse, hoisted out of the sampling loop.

Reference: /root/reference/models/unet_spatio_temporal_condition.py:536-595 recomputes this block in every UNet
forward although its inputs (CLIP embedding, domain-ViT logits, flow-ViT logits) are the same for all 25 steps
(SURVEY.md finding 6).  It is ~1.3 M multiply-adds of fp32 work on [B, 1024] vectors (grouped 1x1 conv taps, quaternion
linears, a 256-point real DFT and a 512-sample inverse), so here it runs ONCE per clip (`pipeline.denoise`): one launch of
`lkgd_lk_fuse` (lkgd_amd/csrc/lk_fuse.hip, one workgroup per batch entry, fp32 - the reference's FFT has no half path
either).  Until round 5 this was PyTorch-ROCm tensor ops (rocFFT + hipBLASLt); `unet.forward` callers that pass the same
tensor objects hit a one-entry cache.  The per-step UNet forward never touches it.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import LkgdHipError, check


def hamilton(q) -> torch.Tensor:
    """core_qnn quaternion_linear weight: [in, out] block matrix from r/i/j/k [in/4, out/4]"""
    r, i, j, k = (w.detach().float() for w in (q.r_weight, q.i_weight, q.j_weight, q.k_weight))
    return torch.cat([torch.cat([r, -i, -j, -k], 0), torch.cat([i, r, -k, j], 0),
                      torch.cat([j, k, r, -i], 0), torch.cat([k, -j, i, r], 0)], 1)


def pack_lk(unet):
    """the 18 fp32 operands of lkgd_lk_fuse (include/lkgd_hip.h section 17), matrices as (in, out) row-major; built once per
    weight version (kept on the model's pack)"""
    def f32(t):
        return t.detach().to(device=unet.device, dtype=torch.float32).contiguous()
    sf = unet.quaternion_lora_fuse_sf
    l0m, l0p = unet.quaternion_lora_fuse_fft_mag0, unet.quaternion_lora_fuse_fft_pha0
    ws = [f32(unet.quaternion_lora_lconv.weight.reshape(256, 4)), f32(unet.quaternion_lora_dconv.weight.reshape(256, 4)),
          f32(unet.quaternion_lora_fconv.weight.reshape(256, 4)), f32(unet.quaternion_lora_texts.reshape(256)),
          f32(hamilton(unet.quaternion_lora_fuse)), f32(unet.quaternion_lora_fuse.bias),
          f32(unet.quaternion_lora_texts_fft_mag.reshape(129)), f32(unet.quaternion_lora_texts_fft_pha.reshape(129)),
          f32(hamilton(unet.quaternion_lora_fuse_fft_mag)), f32(unet.quaternion_lora_fuse_fft_mag.bias),
          f32(hamilton(unet.quaternion_lora_fuse_fft_pha)), f32(unet.quaternion_lora_fuse_fft_pha.bias),
          f32(torch.cat([l0m.weight.reshape(4), l0m.bias.reshape(1)])), f32(torch.cat([l0p.weight.reshape(4), l0p.bias.reshape(1)])),
          f32(sf[0].weight.T), f32(sf[0].bias), f32(sf[2].weight.T), f32(sf[2].bias)]
    shapes = [(256, 4)] * 3 + [(256,), (1024, 512), (512,), (129,), (129,), (512, 256), (256,), (512, 256), (256,), (5,), (5,),
              (1024, 256), (256,), (256, 1024), (1024,)]
    for w, shp in zip(ws, shapes):
        if tuple(w.shape) != shp:
            raise LkgdHipError(f"latent-knowledge fuse: parameter of shape {tuple(w.shape)}, expected {shp}")
    ptrs = (C.c_void_p * 18)(*[w.data_ptr() for w in ws])
    return ws, ptrs


@torch.no_grad()
def lk_fuse(unet, encoder_hidden_states, domain_features, flow_features) -> torch.Tensor:
    dev = unet.device
    if dev.type != "cuda":
        raise LkgdHipError("the latent-knowledge fuse runs on the GPU (lkgd_amd has no CPU path)")
    unet.prepare()
    pk = unet._pk
    if getattr(pk, "lk", None) is None:
        pk.lk = pack_lk(unet)
    e = encoder_hidden_states.to(device=dev, dtype=torch.float32).contiguous()
    d = domain_features.to(device=dev, dtype=torch.float32).contiguous()
    f = flow_features.to(device=dev, dtype=torch.float32).contiguous()
    if e.dim() != 3 or e.shape[1] != 1 or e.shape[2] != 1024:
        raise LkgdHipError("latent-knowledge fuse: encoder_hidden_states must be [batch, 1, 1024]")
    B, Bd = e.shape[0], d.shape[0]
    if tuple(d.shape[1:]) != (1, 1000) or tuple(f.shape) != tuple(d.shape):
        raise LkgdHipError("latent-knowledge fuse: domain / flow features must be [1 or batch, 1, 1000]")
    if Bd != B and Bd != 1:
        raise LkgdHipError(f"latent-knowledge fuse: {Bd} feature rows for a batch of {B} (1 or {B})")
    out = torch.empty(B, 1, 1024, dtype=torch.float16, device=dev)
    check(_lib.lib().lkgd_lk_fuse(e.data_ptr(), d.data_ptr(), f.data_ptr(), B, Bd, pk.lk[1], out.data_ptr(), 1024,
                                  torch.cuda.current_stream(dev).cuda_stream), "lkgd_lk_fuse")
    return out                                                         # REPLACES the CLIP embedding (:595,:613)


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
