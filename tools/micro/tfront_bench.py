#!/usr/bin/env python3
"""Front of a temporal transformer block at the 72x128 level: LayerNorm + QKV GEMM + attn_temporal (three launches) against
the fused lkgd_tattn_front.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
from lkgd_amd.packing import pack_linear, pack_tfront
DEV = "cuda:0"
B, Fr, HW, C, heads = 2, 14, 72 * 128, 320, 5
T = B * Fr * HW
x = (torch.randn(T, C, device=DEV) * 1.5).half()
w = (torch.randn(3 * C, C, device=DEV) / C ** 0.5).half()
b = torch.randn(3 * C, device=DEV) * 0.1
wl, wf = pack_linear(w), pack_tfront(w, heads)
qkv = torch.empty(T, 3 * C, dtype=torch.float16, device=DEV)
att = torch.empty(T, C, dtype=torch.float16, device=DEV)
out = torch.empty(T, C, dtype=torch.float16, device=DEV)


def unfused():
    ln = ops.layernorm(x, None, None, 1e-5)
    ops.gemm(ln, wl, qkv, M=T, N=3 * C, K=C, bias=b)
    ops.attn_temporal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], att, B, Fr, HW, heads)


def fused():
    ops.tattn_front(x, wf, b, out, B, Fr, HW, heads)


def t(fn, iters=10):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best


a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
for _ in range(60):
    a0 @ a0
torch.cuda.synchronize()
tu, tf = t(unfused), t(fused)
flop = 2.0 * T * 3 * C * C + 4.0 * B * HW * heads * Fr * Fr * 64
print(f"72x128 temporal block front: unfused (LN + QKV + attention) {tu:.3f} ms, fused {tf:.3f} ms = {flop / tf / 1e9:.0f} TFLOP/s "
      f"({flop / tf / 1e9 / 2500:.3f} of the 2.5 PFLOP/s MFMA peak); rel L2 fused vs unfused "
      f"{((out.float() - att.float()).norm() / att.float().norm()).item():.2e}")
