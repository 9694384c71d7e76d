#!/usr/bin/env python3
"""LayerNorm + Q|K|V in one launch (lkgd_ln_qkv_c320 / _c640) against LayerNorm + GEMM at the row counts of sharded ranks
(ops.ln_qkv_ok's row threshold).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
from lkgd_amd.packing import pack_ln_proj, pack_linear
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
del a0


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best * 1e3


g = torch.Generator().manual_seed(1)
for C, rows in ((640, (64512, 32256, 16128, 9216)), (320, (258048, 129024, 64512, 36864))):
    w = (torch.randn(3 * C, C, generator=g) / C ** 0.5).half()
    b = torch.randn(3 * C, generator=g)
    wl = pack_ln_proj(pack_linear(w).to(DEV), b.to(DEV))
    wq, bq = pack_linear(w).to(DEV), b.to(DEV)
    for T in rows:
        x = torch.randn(T, C, generator=g).half().to(DEV)
        out = torch.empty(T, 3 * C, dtype=torch.float16, device=DEV)
        fused = timed(lambda: ops.ln_qkv(x, wl, out))
        def two():
            ln = ops.layernorm(x, None, None, 1e-5)
            ops.gemm(ln, wq, out, M=T, N=3 * C, K=C, bias=bq)
        sep = timed(two)
        print(f"C={C} T={T:7d} ({(T + 127) // 128:5d} panels): one launch {fused:7.1f} us   LayerNorm + GEMM {sep:7.1f} us", flush=True)
