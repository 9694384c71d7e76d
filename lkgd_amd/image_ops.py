"""Clip-level image pre-processing of the CLIP branch on the GPU (boundary stage).

`resize_with_antialiasing` keeps the name, argument order and behaviour of the reference's `_resize_with_antialiasing`
(pipeline/pipeline_stable_video_diffusion_trans.py:661-687): Gaussian blur sized from the down-scaling factor (reflect
padding, separable) followed by bicubic interpolation with align_corners=True.  Both steps are HIP kernels
(lkgd_amd/csrc/image_ops.hip, include/lkgd_hip.h section 11); there is no CPU path - a CPU tensor is moved to the GPU and
the result returned on the input's device.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import torch

from . import ops
from ._lib import LkgdHipError


def _geometry(in_hw, out_hw):
    """reference :662-681 - python-float arithmetic and int() truncation reproduced as is"""
    fy, fx = in_hw[0] / out_hw[0], in_hw[1] / out_hw[1]
    sig = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
    return sig, tuple(k + 1 if k % 2 == 0 else k for k in ks)


def _taps(k: int, sigma: float) -> torch.Tensor:
    """`_gaussian` (:736-749) in fp32, as the reference evaluates it"""
    x = torch.arange(k, dtype=torch.float32) - k // 2
    g = torch.exp(-x.pow(2.0) / (2 * torch.tensor(sigma, dtype=torch.float32).pow(2.0)))
    return g / g.sum()


def resize_with_antialiasing(input: torch.Tensor, size: Tuple[int, int], interpolation: str = "bicubic",
                             align_corners: bool = True) -> torch.Tensor:
    if interpolation != "bicubic" or not align_corners:
        raise LkgdHipError("only the reference's call form (bicubic, align_corners=True) is implemented")
    if input.dim() != 4:
        raise ValueError("input must be [batch, channels, height, width]")
    if not torch.cuda.is_available():
        raise LkgdHipError("resize_with_antialiasing needs the GPU (lkgd_amd has no CPU path)")
    src_device = input.device
    x = input.to(device="cuda", dtype=torch.float32).contiguous()
    b, c, h, w = x.shape
    sig, ks = _geometry((h, w), size)
    if ks[0] // 2 >= h or ks[1] // 2 >= w:
        raise ValueError("image smaller than the blur's reflect padding")
    y = ops.conv1d_reflect(x, _taps(ks[1], sig[1]).to(x.device), axis=1)
    y = ops.conv1d_reflect(y, _taps(ks[0], sig[0]).to(x.device), axis=0)
    out = ops.resize_bicubic_ac(y, int(size[0]), int(size[1]))
    return out.to(device=src_device, dtype=input.dtype) if src_device != out.device or input.dtype != out.dtype else out
