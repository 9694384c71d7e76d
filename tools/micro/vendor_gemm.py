#!/usr/bin/env python3
"""Where do the hand-written GEMM programs stand against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the
model's PLAIN shapes (no fused epilogue)?  Interleaved in one process after a clock warm-up."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lkgd_amd import ops

DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20):
        a0 @ a0
    torch.cuda.synchronize()
M0 = 28 * 72 * 128
for (name, M, N, K) in (("L0 qkv", M0, 960, 320), ("L1 qkv", M0 // 4, 1920, 640), ("L2 qkv", M0 // 16, 3840, 1280),
                        ("L0 ffout", M0, 320, 1280), ("L1 ffout", M0 // 4, 640, 2560), ("L2 ffout", M0 // 16, 1280, 5120),
                        ("deep K", 32768, 2560, 5120), ("8192^3", 8192, 8192, 8192)):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    fns = {"lkgd": lambda: ops.gemm(a, w, out, M=M, N=N, K=K), "vendor": lambda: torch.matmul(a, w.T, out=out)}
    best = {k: 1e9 for k in fns}
    for rep in range(3):
        for k, fn in fns.items():
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record(); torch.cuda.synchronize()
            best[k] = min(best[k], s.elapsed_time(e) / 10)
    fl = 2.0 * M * N * K
    print("%-9s %7dx%5dx%5d  lkgd %7.3f ms %7.1f TF/s   vendor %7.3f ms %7.1f TF/s" % (
        name, M, N, K, best["lkgd"], fl / best["lkgd"] / 1e9, best["vendor"], fl / best["vendor"] / 1e9))
