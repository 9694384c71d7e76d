"""Weight re-layout for the gfx950 kernels (done once at load time, on whatever device the weights live on).

All GEMM-shaped weights become fp16 [N][K] with K contiguous and K ordered the way the kernel's A-operand gather walks
it (include/lkgd_hip.h section 1)."""
from __future__ import annotations

import torch


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> fp16 [N, K]"""
    return w.reshape(w.shape[0], -1).to(torch.float16).contiguous()


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] -> fp16 [Cout, 9*Cin], k = (ky*3+kx)*Cin + c"""
    co, ci = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).to(torch.float16).contiguous()


def pack_conv3x3_c8(w: torch.Tensor) -> torch.Tensor:
    """conv_in weight [Cout, 8, 3, 3] -> fp16 [Cout, 128]: 9 taps x 8 channels, zero padded to K = 128"""
    co, ci = w.shape[:2]
    assert ci == 8, "LKGD_A_CONV3X3_C8 is the conv_in path (8 input channels)"
    out = torch.zeros(co, 128, dtype=torch.float16, device=w.device)
    out[:, :72] = w.permute(0, 2, 3, 1).reshape(co, 72).to(torch.float16)
    return out


def pack_tconv3(w: torch.Tensor) -> torch.Tensor:
    """Conv3d (3,1,1) weight [Cout, Cin, 3, 1, 1] -> fp16 [Cout, 3*Cin], k = kt*Cin + c"""
    co, ci = w.shape[:2]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(co, 3 * ci).to(torch.float16).contiguous()


def geglu_perm(inner: int, device=None) -> torch.Tensor:
    """row permutation of the GEGLU projection [2*inner, K]: every 64 packed rows = 32 hidden rows followed by their 32
    gate rows (inner + ..), so that the 64 output columns one wave owns hold both factors of 32 GEGLU outputs"""
    assert inner % 64 == 0
    t = torch.arange(inner // 32, device=device)[:, None] * 32 + torch.arange(32, device=device)[None, :]
    return torch.cat([t, t + inner], dim=1).reshape(-1)


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    inner = w.shape[0] // 2
    perm = geglu_perm(inner, w.device)
    return w[perm].to(torch.float16).contiguous(), b[perm].to(torch.float32).contiguous()
