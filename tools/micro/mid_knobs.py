#!/usr/bin/env python3
"""times one few-row GEMM on the forced four-stage 128x128 program over K (slope = us per K-tile, intercept = fixed cost)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops, _lib
L = _lib.lib()
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
L.lkgd_debug_set_gemm_variant(int(os.environ.get("V", "7")))
L.lkgd_debug_set_gemm_splitk(0)
row = []
for M, N in ((2304, 1280), (9216, 640)):
    ts = []
    for K in (640, 1280, 2560, 5120):
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        ts.append(t(lambda: ops.gemm(a, w, out, M=M, N=N, K=K)))
    slope = (ts[3] - ts[1]) / 60
    row.append(f"M={M} N={N}: " + " ".join(f"{x:6.1f}" for x in ts) + f" us (K=640..5120); {slope:.3f} us/K-tile, fixed {ts[1]-20*slope:.1f}")
print(os.environ.get("LKGD_HIP_LIB", "in-tree").split("_")[-1], " | ".join(row))
