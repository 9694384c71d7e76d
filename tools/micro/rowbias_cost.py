#!/usr/bin/env python3
"""What does the row-indexed bias (time embedding) cost in the epilogue of the 3x3 / temporal convs?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lkgd_amd import ops
from lkgd_amd.packing import pack_conv3x3, pack_tconv3

DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20):
        a0 @ a0
    torch.cuda.synchronize()
for (name, H, W, C) in (("L0 320", 72, 128, 320), ("L1 640", 36, 64, 640), ("L2 1280", 18, 32, 1280)):
    T = 28 * H * W
    x = torch.randn(T, C, device=DEV, dtype=torch.float16)
    w = pack_conv3x3(torch.randn(C, C, 3, 3, device=DEV) / (9 * C) ** 0.5)
    wt = pack_tconv3(torch.randn(C, C, 3, 1, 1, device=DEV) / (3 * C) ** 0.5)
    b = torch.zeros(C, device=DEV)
    temb = torch.randn(2, C, device=DEV, dtype=torch.float16)
    out = torch.empty(T, C, device=DEV, dtype=torch.float16)
    cases = {
        "conv": lambda: ops.gemm(x, w, out, M=T, N=C, K=9 * C, bias=b, mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0)),
        "conv+temb": lambda: ops.gemm(x, w, out, M=T, N=C, K=9 * C, bias=b, mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0),
                                      rowbias=temb, rowmap=ops.rowmap_div(14 * H * W)),
        "conv+res": lambda: ops.gemm(x, w, out, M=T, N=C, K=9 * C, bias=b, mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0),
                                     res1=x),
        "tconv": lambda: ops.gemm(x, wt, out, M=T, N=C, K=3 * C, bias=b, mode=ops.A_TCONV3, Cin=C, tconv=(14, H * W)),
        "tconv+temb": lambda: ops.gemm(x, wt, out, M=T, N=C, K=3 * C, bias=b, mode=ops.A_TCONV3, Cin=C, tconv=(14, H * W),
                                       rowbias=temb, rowmap=ops.rowmap_div(14 * H * W)),
        "tconv+res": lambda: ops.gemm(x, wt, out, M=T, N=C, K=3 * C, bias=b, mode=ops.A_TCONV3, Cin=C, tconv=(14, H * W),
                                      s_acc=0.5, res1=x),
    }
    best = {k: 1e9 for k in cases}
    for rep in range(4):
        for k, fn in cases.items():
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record(); torch.cuda.synchronize()
            best[k] = min(best[k], s.elapsed_time(e) / 10)
    print(name, "  ".join("%s %.3f" % (k, v) for k, v in best.items()))
