"""ORACLE (test infrastructure only - never imported by lkgd_amd): CPU restatement of the clip-level image
pre-processing of the CLIP branch.

`resize_with_antialiasing` follows the reference's `_resize_with_antialiasing`
(pipeline/pipeline_stable_video_diffusion_trans.py:661-687) and its helpers `_gaussian_blur2d` (:752-765), `_filter2d`
(:713-733), `_gaussian` (:736-749), `_compute_padding` (:690-710):
  * per axis, sigma = max((in/out - 1)/2, 0.001); taps = odd(int(max(4 sigma, 3)));
  * Gaussian taps exp(-x^2 / 2 sigma^2), x = -(k//2) .. k//2, normalised to sum 1;
  * blur = a 1-D pass along W, then one along H, each with REFLECT padding of (k-1)//2 | k-1-(k-1)//2;
  * then bicubic interpolation with align_corners=True.
Pinned against tests/golden/image_ops.safetensors (outputs of the reference function itself, tests/golden/make_goldens.py).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F


def blur_geometry(in_hw: Tuple[int, int], out_hw: Tuple[int, int]):
    """((sigma_y, sigma_x), (ky, kx)) exactly as the reference derives them (python floats, int() truncation)"""
    fy, fx = in_hw[0] / out_hw[0], in_hw[1] / out_hw[1]
    sig = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    return sig, tuple(ks)


def gaussian_taps(k: int, sigma: float) -> torch.Tensor:
    s = torch.tensor([[sigma]], dtype=torch.float32)
    x = (torch.arange(k, dtype=torch.float32) - k // 2).expand(1, -1)
    if k % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * s.pow(2.0)))
    return (g / g.sum(-1, keepdim=True)).reshape(-1)


def _pass(x: torch.Tensor, taps: torch.Tensor, axis: int) -> torch.Tensor:
    b, c, h, w = x.shape
    k = taps.numel()
    front = (k - 1) // 2
    rear = k - 1 - front
    if axis == 1:
        xp = F.pad(x, (front, rear, 0, 0), mode="reflect")
        wgt = taps.reshape(1, 1, 1, k)
    else:
        xp = F.pad(x, (0, 0, front, rear), mode="reflect")
        wgt = taps.reshape(1, 1, k, 1)
    y = F.conv2d(xp.reshape(b * c, 1, xp.shape[-2], xp.shape[-1]), wgt.to(x.dtype))
    return y.reshape(b, c, h, w)


def resize_with_antialiasing(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    sig, ks = blur_geometry(tuple(x.shape[-2:]), size)
    y = _pass(x, gaussian_taps(ks[1], sig[1]), axis=1)
    y = _pass(y, gaussian_taps(ks[0], sig[0]), axis=0)
    return F.interpolate(y, size=size, mode="bicubic", align_corners=True)
