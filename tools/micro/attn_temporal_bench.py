#!/usr/bin/env python3
"""Temporal attention (14 frames per pixel) at the four UNet levels: time and q+k+v+out bytes per second.  GPU box only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
DEV = "cuda:0"
for (H, W, C) in ((72, 128, 320), (36, 64, 640), (18, 32, 1280), (9, 16, 1280)):
    B, F, S = 2, 14, H * W
    T = B * F * S
    qkv = torch.randn(T, 3 * C, device=DEV, dtype=torch.float16)
    out = torch.empty(T, C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.attn_temporal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, B, F, S, C // 64)   # noqa: E731
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    print(f"attn_temporal {H}x{W} C={C}: {best*1e3:7.1f} us  {4*T*C*2/best/1e9:5.2f} TB/s")
