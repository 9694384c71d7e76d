#!/usr/bin/env python3
"""Cost / return of the GroupNorm statistics from the producing GEMM's epilogue (lkgd_gemm_desc.colstats): per shape the
GEMM with and without column sums, and the statistics from the columns vs the separate read pass.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
DEV = "cuda:0"
z = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16) * 0.1   # noqa: E731


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
print(f"{'shape':28s}  gemm  +colsums |  stats: read pass  from columns   (ms)")
for name, H, W, C, kind in (("conv3x3 L0 320", 72, 128, 320, "conv"), ("conv3x3 L1 640", 36, 64, 640, "conv"),
                            ("conv3x3 L2 1280", 18, 32, 1280, "conv"), ("tconv L0 320", 72, 128, 320, "tconv"),
                            ("tconv L1 640", 36, 64, 640, "tconv"), ("tconv L2 1280", 18, 32, 1280, "tconv"),
                            ("proj_out L0 320", 72, 128, 320, "lin"), ("proj_out L1 640", 36, 64, 640, "lin")):
    N_IMG, HW = 28, H * W
    M = N_IMG * HW
    x = z(M, C)
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    b = torch.zeros(C, device=DEV)
    if kind == "conv":
        w = z(C, 9 * C)
        kw = dict(M=M, N=C, K=9 * C, bias=b, mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0))
    elif kind == "tconv":
        w = z(C, 3 * C)
        kw = dict(M=M, N=C, K=3 * C, bias=b, mode=ops.A_TCONV3, Cin=C, tconv=(14, HW), res1=x)
    else:
        w = z(C, C)
        kw = dict(M=M, N=C, K=C, bias=b, res1=x)
    t0 = t(lambda: ops.gemm(x, w, out, **kw))
    ns, rows = (N_IMG, HW) if kind != "tconv" else (2, 14 * HW)
    t1 = t(lambda: ops.gemm(x, w, out, colstats=rows, **kw))
    ops.gemm(x, w, out, colstats=rows, **kw)
    has = getattr(out, "_lkgd_colstats", None) is not None
    s1 = t(lambda: ops.groupnorm_stats(out, None, ns, rows, 1e-5)) if has else float("nan")
    ops.COLSTATS = False
    s0 = t(lambda: ops.groupnorm_stats(out, None, ns, rows, 1e-5))
    ops.COLSTATS = True
    print(f"{name:28s} {t0:6.3f}  {t1:6.3f}  |  {s0:6.3f}   {s1:6.3f}", flush=True)
