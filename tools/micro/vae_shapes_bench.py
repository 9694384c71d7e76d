#!/usr/bin/env python3
"""The temporal VAE decoder's convolution shapes (14 frames, 128 / 256 / 512 channels: none a multiple of 320) under each GEMM
tile program.  GPU box only."""
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
F = 14
names = {0: "auto", 1: "t128", 3: "strm", 4: "wide"}
tot = {v: 0.0 for v in names}
for (H, W, cin, cout, cnt, kind) in ((72, 128, 512, 512, 8, "conv"), (144, 256, 512, 512, 7, "conv"), (288, 512, 512, 256, 1, "conv"),
                                     (288, 512, 256, 256, 6, "conv"), (576, 1024, 256, 128, 1, "conv"), (576, 1024, 128, 128, 6, "conv"),
                                     (72, 128, 512, 512, 8, "tconv"), (144, 256, 512, 512, 6, "tconv"), (288, 512, 256, 256, 6, "tconv"),
                                     (576, 1024, 128, 128, 6, "tconv")):
    M = F * H * W
    x = torch.randn(M, cin, device=DEV, dtype=torch.float16) * 0.1
    b = torch.zeros(cout, device=DEV)
    out = torch.empty(M, cout, device=DEV, dtype=torch.float16)
    if kind == "conv":
        K = 9 * cin
        w = torch.randn(cout, K, device=DEV, dtype=torch.float16) * 0.02
        fn = lambda: ops.gemm(x, w, out, M=M, N=cout, K=K, bias=b, mode=ops.A_CONV3X3, Cin=cin, conv=(H, W, H, W, 1, 0))   # noqa: E731
    else:
        K = 3 * cin
        w = torch.randn(cout, K, device=DEV, dtype=torch.float16) * 0.02
        fn = lambda: ops.gemm(x, w, out, M=M, N=cout, K=K, bias=b, mode=ops.A_TCONV3, Cin=cin, tconv=(F, H * W))   # noqa: E731
    res = {}
    for rep in range(2):
        for v in names:
            L.lkgd_debug_set_gemm_variant(v)
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3): fn()
            e.record(); torch.cuda.synchronize()
            res[v] = min(res.get(v, 1e9), s.elapsed_time(e) / 3)
    L.lkgd_debug_set_gemm_variant(0)
    fl = 2.0 * M * cout * K
    for v in names: tot[v] += res[v] * cnt
    print(f"{kind:5s} {H:4d}x{W:4d} {cin:3d}->{cout:3d} x{cnt}: " + "  ".join(f"{names[v]} {res[v]:7.3f} ms {fl/res[v]/1e9:5.0f} TF/s" for v in names), flush=True)
    del x, out
print("TOTAL ms: " + "  ".join(f"{names[v]} {tot[v]:7.1f}" for v in names))
