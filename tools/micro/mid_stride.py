#!/usr/bin/env python3
"""Does the row stride of the A operand matter to the 128x128 programs (L2 channel conflicts of 128-byte row pieces that
lie K*2 bytes apart)?  One few-row shape, A as a column slice of a wider buffer."""
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
for M, N, K in ((2304, 1280, 1280), (2304, 1280, 5120), (9216, 640, 2560), (576, 1280, 5120)):
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for pad in (0, 8, 64, 72):
        big = torch.randn(M, K + pad, device=dev, dtype=torch.float16)
        a = big[:, :K]
        row = [f"M={M} N={N} K={K} lda={K+pad:5d}:"]
        for v in (1, 7, 4):
            L.lkgd_debug_set_gemm_variant(v)
            try:
                row.append(f"v{v} {t(lambda: ops.gemm(a, w, out, M=M, N=N, K=K)):6.1f} us")
            except Exception as ex:
                row.append(f"v{v} n/a")
        print("  ".join(row), flush=True)
L.lkgd_debug_set_gemm_variant(0)
