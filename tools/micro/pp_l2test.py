#!/usr/bin/env python3
"""Is the ping-pong GEMM limited by where its operands come from?  Same launch with (a) real operands, (b) every A row aliased to one row (row stride 0: everything is L2-resident after the first touch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import _lib, ops
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
for variant in (6, 4, 3):
    _lib.lib().lkgd_debug_set_gemm_variant(variant)
    for (M, N, K) in ((16384, 4096, 5120), (16384, 1280, 5120), (65536, 1280, 1280)):
        for alias in (False, True):
            if alias:
                a = (torch.randn(1, K, device=DEV, dtype=torch.float16) * 0.1).expand(M, K)
                w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1     # weights are dense [N][K] (no row stride)
            else:
                a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
                w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
            out = torch.empty(M, N, device=DEV, dtype=torch.float16)
            fn = lambda: ops.gemm(a, w, out, M=M, N=N, K=K)
            best = 1e9
            for rep in range(3):
                fn(); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): fn()
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 5)
            print("variant %d %6dx%5dx%5d %s  %7.3f ms  %7.1f TF/s" % (variant, M, N, K, "ALIASED" if alias else "real   ",
                                                                      best, 2.0 * M * N * K / best / 1e9), flush=True)
