#!/usr/bin/env python3
"""72x128-level spatial norm1 -> QKV: LayerNorm kernel + row-panel GEMM against the GEMM with the LayerNorm folded in
(lkgd_gemm_desc.ln_colsum)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
DEV = "cuda:0"
M, N, K = 28 * 72 * 128, 960, 320
x = (torch.randn(M, K, device=DEV) * 1.2).half()
w = (torch.randn(N, K, device=DEV) / K ** 0.5).half()
b = torch.randn(N, device=DEV) * 0.1
cs = w.float().sum(1).contiguous()
out = torch.empty(M, N, dtype=torch.float16, device=DEV)
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
for _ in range(60): a0 @ a0


def t(fn):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    return best


two = t(lambda: ops.gemm(ops.layernorm(x, None, None, 1e-5), w, out, M=M, N=N, K=K, bias=b))
gem = t(lambda: ops.gemm(x, w, out, M=M, N=N, K=K, bias=b))
fold = t(lambda: ops.gemm(x, w, out, M=M, N=N, K=K, bias=b, ln=(cs, 1e-5)))
print(f"LayerNorm + QKV GEMM {two:.3f} ms (the GEMM alone {gem:.3f}); LayerNorm folded into the GEMM {fold:.3f} ms")
