#!/usr/bin/env python3
"""192- against 256-row tiles (and 256- against 320-column ones) of the 256x320 program on the unsliced shapes of sharded
ranks and of the full forward: the per-tile cost factors behind gemm.hip::gemm_wide_form.  GPU box only."""
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


# (name, frame-images, rows per image, H, W, N, K, kind, residual)
CASES = []
for name, imgs in (("full", 28), ("rank of 2", 14), ("rank of 4", 7), ("rank of 8", 4)):
    CASES += [(name, imgs, 9216, 72, 128, 320, 2880, "conv", True), (name, imgs, 9216, 72, 128, 320, 960, "tconv", True),
              (name, imgs, 2304, 36, 64, 640, 5760, "conv", True), (name, imgs, 2304, 36, 64, 640, 1920, "tconv", True),
              (name, imgs, 2304, 36, 64, 640, 2560, "lin", True), (name, imgs, 2304, 36, 64, 640, 640, "lin", True),
              (name, imgs, 576, 18, 32, 1280, 11520, "conv", True), (name, imgs, 576, 18, 32, 1280, 1280, "lin", True),
              (name, imgs, 576, 18, 32, 3840, 1280, "lin", False), (name, imgs, 576, 18, 32, 1280, 5120, "lin", True)]
only = os.environ.get("ONLY")
for (name, imgs, per, H, W, N, K, kind, res) in CASES:
    if only and only not in name:
        continue
    M = imgs * per
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
    L.lkgd_debug_set_wide_ksplit(1) if hasattr(L, "lkgd_debug_set_wide_ksplit") else None
    L.lkgd_debug_set_gemm_splitk(0)
    for wm in (256, 192):
        for wn in ((320, 256) if N % 256 == 0 else (320,)):
            L.lkgd_debug_set_wide_tile_m(wm); L.lkgd_debug_set_wide_tile_n(wn)
            res_t[f"{wm}x{wn}"] = timed(fn)
    L.lkgd_debug_set_wide_tile_m(0); L.lkgd_debug_set_wide_tile_n(0)
    L.lkgd_debug_set_gemm_splitk(1); L.lkgd_debug_set_wide_ksplit(0)
    L.lkgd_debug_set_gemm_variant(0)
    t256, t192 = (M + 255) // 256 * (N // 320), (M + 191) // 192 * (N // 320)
    print(f"{name:9s} {kind:5s} {M:6d} x {N:5d} x {K:5d}  tiles {t256:4d} / {t192:4d}: " +
          "  ".join(f"{k} {v:7.1f}" for k, v in res_t.items()), flush=True)
