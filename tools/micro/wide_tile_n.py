#!/usr/bin/env python3
"""256- against 320-column tiles of the 256x320 program on shapes whose channel count both widths divide (N = 1280, 3840):
the row counts of full / rank-of-2 / -4 forwards at the 18x32 and 9x16 levels.  GPU box only."""
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


for rows_name, imgs in (("full", 28), ("rank of 2", 14), ("rank of 4", 7)):
    for (per, N, K, kind, res) in ((576, 1280, 1280, "lin", True), (576, 3840, 1280, "lin", False), (576, 1280, 5120, "lin", True),
                                   (576, 1280, 3840, "tconv", True), (576, 1280, 11520, "conv", True), (144, 1280, 11520, "conv", True),
                                   (144, 1280, 3840, "tconv", True)):
        M = imgs * per
        H, W = (18, 32) if per == 576 else (9, 16)
        x = (torch.randn(M, K if kind == "lin" else K // (9 if kind == "conv" else 3), device=DEV) * 0.1).half()
        w = (torch.randn(N, K, device=DEV) * 0.02).half()
        b = torch.zeros(N, device=DEV)
        r = (torch.randn(M, N, device=DEV) * 0.1).half() if res else None
        out = torch.empty(M, N, device=DEV, dtype=torch.float16)
        if kind == "lin":
            fn = lambda: ops.gemm(x, w, out, M=M, N=N, K=K, bias=b, res1=r)   # noqa: E731
        elif kind == "conv":
            fn = lambda: ops.gemm(x, w, out, M=M, N=N, K=K, bias=b, res1=r, mode=ops.A_CONV3X3, Cin=K // 9, conv=(H, W, H, W, 1, 0))   # noqa: E731
        else:
            fn = lambda: ops.gemm(x, w, out, M=M, N=N, K=K, bias=b, res1=r, mode=ops.A_TCONV3, Cin=K // 3, tconv=(imgs, per))   # noqa: E731
        res_t = {}
        L.lkgd_debug_set_gemm_variant(0)
        res_t["auto"] = timed(fn)
        L.lkgd_debug_set_gemm_variant(4)
        for wn in (320, 256):
            L.lkgd_debug_set_wide_tile_n(wn)
            res_t[f"wide{wn}"] = timed(fn)
        L.lkgd_debug_set_wide_tile_n(0)
        L.lkgd_debug_set_gemm_variant(0)
        tm = (M + 255) // 256
        print(f"{rows_name:9s} {kind:5s} {M:6d} x {N:5d} x {K:5d}  tiles {tm * N // 320:4d} / {tm * N // 256:4d}: " +
              "  ".join(f"{k} {v:7.1f} us" for k, v in res_t.items()), flush=True)
