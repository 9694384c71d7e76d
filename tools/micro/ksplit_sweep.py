#!/usr/bin/env python3
"""Few-row, deep-K shapes on the 256x320 kernel: time per K-slice count (lkgd_debug_set_wide_ksplit) against the automatic
dispatch and the 128x128 program.  Shapes: the 9x16 level (4032 rows) and the levels of frame-sharded ranks.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import _lib, ops
DEV = "cuda:0"
L = _lib.lib()
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()


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


shapes = []   # (name, M, N, K, mode kwargs)
for B, F, H, W, C, tag in ((2, 14, 9, 16, 1280, "L3"), (1, 14, 18, 32, 1280, "L2 rank/2"), (1, 4, 18, 32, 1280, "L2 rank/8"),
                           (1, 4, 36, 64, 640, "L1 rank/8"), (1, 7, 18, 32, 1280, "L2 rank/4")):
    M = B * F * H * W
    shapes.append((f"tconv {tag}", M, C, 3 * C, dict(mode=ops.A_TCONV3, Cin=C, tconv=(F, H * W))))
    shapes.append((f"conv3x3 {tag}", M, C, 9 * C, dict(mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0))))
    shapes.append((f"ffout {tag}", M, C, 4 * C, dict()))
for name, M, N, K, kw in shapes:
    x = torch.randn(M, K if not kw else kw["Cin"], device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.02
    b = torch.zeros(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(x, w, out, M=M, N=N, K=K, bias=b, **kw)   # noqa: E731
    res = {}
    L.lkgd_debug_set_wide_ksplit(0); L.lkgd_debug_set_gemm_variant(0)
    res["auto"] = t(fn)
    L.lkgd_debug_set_gemm_variant(1); res["t128"] = t(fn)
    L.lkgd_debug_set_gemm_variant(4)
    L.lkgd_debug_set_gemm_splitk(0); res["wide/1"] = t(fn); L.lkgd_debug_set_gemm_splitk(1)
    for ks in (2, 3, 4, 5, 6, 8):
        if (K // 64) % ks == 0:
            L.lkgd_debug_set_wide_ksplit(ks)
            res[f"wide/{ks}"] = t(fn)
    L.lkgd_debug_set_wide_ksplit(0); L.lkgd_debug_set_gemm_variant(0)
    best = min(res, key=res.get)
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d} ({K//64:3d} K-tiles, {((M+255)//256)*((N+319)//320):3d} tiles): " +
          "  ".join(f"{k} {v*1e3:6.1f}{'*' if k == best else ' '}" for k, v in res.items()), flush=True)
