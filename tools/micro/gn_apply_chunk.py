#!/usr/bin/env python3
"""GroupNorm apply (+SiLU) at several rows-per-workgroup settings (lkgd_debug_set_gn_apply_kb).  GPU box only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops, _lib
DEV = "cuda:0"


def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for kb in (32, 64, 128, 256):
    _lib.lib().lkgd_debug_set_gn_apply_kb(kb)
    line = []
    for (H, W, C) in ((72, 128, 320), (72, 128, 960), (36, 64, 640), (18, 32, 1280), (9, 16, 1280)):
        T = 28 * H * W
        x = torch.randn(T, C, device=DEV, dtype=torch.float16)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        out = torch.empty_like(x)
        st = ops.groupnorm_stats(x, None, 28, H * W, 1e-5)
        ms = min(bench(lambda: ops.groupnorm_apply(x, None, 28, H * W, st, g, b, True, out)) for _ in range(3))
        line.append(f"{H}x{W}x{C}: {ms*1e3:6.1f} us {2*T*C*2/ms/1e9:5.2f} TB/s")
    print(f"apply chunk {kb:3d} KiB: " + " ; ".join(line), flush=True)
