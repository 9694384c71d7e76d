#!/usr/bin/env python3
"""lkgd_groupnorm_silu: one launch (a workgroup per (sample, group), values in LDS) against the three launches, on the spatial
GroupNorm maps of the full forward and of sharded ranks (samples = frame-images).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import _lib, ops
DEV = "cuda:0"
L = _lib.lib()


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best * 1e3


for name, imgs in (("full", 28), ("rank of 2", 14), ("rank of 4", 7), ("rank of 8", 4)):
    for (rows, C0, C1) in ((2304, 640, 0), (2304, 640, 320), (576, 1280, 0), (576, 1280, 640), (576, 1280, 1280), (144, 1280, 0), (144, 1280, 1280)):
        C = C0 + C1
        x = torch.randn(imgs * rows, C, device=DEV, dtype=torch.float16)
        x0, x1 = (x[:, :C0], x[:, C0:]) if C1 else (x, None)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        out = torch.empty(imgs * rows, C, device=DEV, dtype=torch.float16)
        res = {}
        for on in (1, 0):
            L.lkgd_debug_set_gn_small(on)
            res[on] = timed(lambda: ops.groupnorm_silu(x0, x1, imgs, rows, g, b, 1e-5, out=out))
        L.lkgd_debug_set_gn_small(1)
        print(f"{name:9s} samples {imgs:2d} x {rows:4d} rows x {C:4d} ch ({rows * C // 32 * 2 // 1024:3d} KiB per group): one launch {res[1]:6.1f} us   three launches {res[0]:6.1f} us", flush=True)
