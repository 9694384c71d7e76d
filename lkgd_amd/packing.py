"""Weight re-layout for the gfx950 kernels (done once at load time, on whatever device the weights live on).

All GEMM-shaped weights become fp16 [N][K] with K contiguous and K ordered the way the kernel's A-operand gather walks
it (include/lkgd_hip.h section 1)."""
from __future__ import annotations

import torch


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> fp16 [N, K]"""
    return w.reshape(w.shape[0], -1).to(torch.float16).contiguous()


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] -> fp16 [Cout, 9*Cin], k = ((ky*(Cin/64) + c//64)*3 + kx)*64 + c%64: for one kernel
    row and one 64-channel chunk the three horizontal taps are consecutive K-tiles (they read the same source cache lines
    shifted by one pixel - lkgd_amd/csrc/gemm_common.h::conv_k_decode).  Cin must be a multiple of 64."""
    co, ci = w.shape[:2]
    assert ci % 64 == 0, "3x3 implicit GEMM needs Cin % 64 == 0 (conv_in uses pack_conv3x3_c8)"
    t = w.reshape(co, ci // 64, 64, 3, 3).permute(0, 3, 1, 4, 2)      # [co, ky, chunk, kx, 64]
    return t.reshape(co, 9 * ci).to(torch.float16).contiguous()


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


def pack_tfront(wqkv: torch.Tensor, heads: int) -> torch.Tensor:
    """fused projection [3C, C] (rows q | k | v, head-major) -> the MFMA-fragment stream of lkgd_tattn_front:
    [head][q,k,v][fragment i of 16 rows][K-step ks of 32][lane = 16*lq + row][8 halfs at k = 32 ks + 8 lq]"""
    c3, c = wqkv.shape
    assert c3 == 3 * c and c == heads * 64 and c % 32 == 0
    t = wqkv.to(torch.float16).reshape(3, heads, 4, 16, c // 32, 4, 8)      # [which, h, i, row, ks, lq, e]
    return t.permute(1, 0, 2, 4, 5, 3, 6).contiguous().reshape(-1)


def geglu_perm(inner: int, half: int = 32, device=None) -> torch.Tensor:
    """row permutation of the GEGLU projection [2*inner, K]: every 2*half packed rows = `half` hidden rows followed by
    their `half` gate rows (inner + ..), so that the output columns one wave owns hold both factors of its GEGLU
    outputs.  half = 80 for the 256x320-tile kernel (wave = 160 columns), 32 for the 128-wide-tile kernels."""
    assert inner % half == 0
    t = torch.arange(inner // half, device=device)[:, None] * half + torch.arange(half, device=device)[None, :]
    return torch.cat([t, t + inner], dim=1).reshape(-1)


def geglu_half(n_packed_rows: int, k: int = 0) -> int:
    """interleave width the kernels want for a GEGLU projection with 2*inner = n_packed_rows rows and K = k inputs"""
    # the 256x320 kernel, 80 hidden | 80 gate per wave, when the rows tile by 320 (measured: profiles/r01_gemm_shapes_ab*.txt;
    # since its bias strip comes through LDS it also leads at K = 320); otherwise 32 | 32 for the other kernels
    if k >= 320 and n_packed_rows % 320 == 0:
        return 80
    return 32


def pack_geglu(w: torch.Tensor, b: torch.Tensor, half: int = None):
    inner = w.shape[0] // 2
    if half is None:
        half = geglu_half(w.shape[0], w.shape[1])
    perm = geglu_perm(inner, half, w.device)
    return w[perm].to(torch.float16).contiguous(), b[perm].to(torch.float32).contiguous(), half
