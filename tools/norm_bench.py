#!/usr/bin/env python3
"""HBM-bound kernels at C2 shapes: GB/s against the 8 TB/s (spec) / ~6.3 TB/s (achievable) HBM roof."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lkgd_amd import ops

DEV = "cuda:0"


def bench(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (H, W, C) in ((72, 128, 320), (36, 64, 640), (18, 32, 1280)):
    T = 28 * H * W
    x = torch.randn(T, C, device=DEV, dtype=torch.float16)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    out = torch.empty_like(x)
    ms = bench(lambda: ops.layernorm(x, g, b, 1e-5, out=out))
    print(f"layernorm   T={T:6d} C={C:4d}: {ms*1e3:7.1f} us  {2*T*C*2/ms/1e6:7.1f} GB/s")
    ms = bench(lambda: ops.layernorm(x, None, None, 1e-5, out=out))
    print(f"layernorm-na T={T:6d} C={C:4d}: {ms*1e3:7.1f} us  {2*T*C*2/ms/1e6:7.1f} GB/s")
    ms = bench(lambda: ops.groupnorm_stats(x, None, 28, H * W, 1e-5))
    print(f"gn stats sp T={T:6d} C={C:4d}: {ms*1e3:7.1f} us  {T*C*2/ms/1e6:7.1f} GB/s")
    st = ops.groupnorm_stats(x, None, 28, H * W, 1e-5)
    ms = bench(lambda: ops.groupnorm_apply(x, None, 28, H * W, st, g, b, True, out))
    print(f"gn apply sp T={T:6d} C={C:4d}: {ms*1e3:7.1f} us  {2*T*C*2/ms/1e6:7.1f} GB/s")
    st = ops.groupnorm_stats(x, None, 2, 14 * H * W, 1e-5)
    ms = bench(lambda: ops.groupnorm_stats(x, None, 2, 14 * H * W, 1e-5))
    print(f"gn stats tm T={T:6d} C={C:4d}: {ms*1e3:7.1f} us  {T*C*2/ms/1e6:7.1f} GB/s")
