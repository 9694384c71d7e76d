#!/usr/bin/env python3
"""Per-wave time split of the 256x320 GEMM program (build: tools/micro/wide_knobs.sh STAMPS -> tools/micro/libwide_STAMPS.so):
s_memtime sums of [wait for the K-tile's loads + barrier], [K-tile bodies incl. the next tile's source preparation], [epilogues]."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import torch
from lkgd_amd import _lib
_lib.LIB_PATH = os.path.join(HERE, "libwide_STAMPS.so")
from lkgd_amd import ops
DEV = "cuda:0"
L = _lib.lib()
L.lkgd_debug_set_gemm_variant(4)


def show(name, out, nk_total_flop_ms=None):
    d = out.view(torch.int64).reshape(-1)[: 256 * 8 * 4].reshape(256, 8, 4).double().cpu()
    m = d.mean(0)
    tot = m[:, 3].mean().item()
    print(f"{name:34s} whole {tot:9.0f} ticks | sync {100 * m[:, 0].mean().item() / tot:5.1f} %  bodies {100 * m[:, 1].mean().item() / tot:5.1f} %  "
          f"epilogue {100 * m[:, 2].mean().item() / tot:5.1f} %   (waves 0-3 sync {100 * m[:4, 0].mean().item() / tot:4.1f} % body {100 * m[:4, 1].mean().item() / tot:4.1f} %; "
          f"waves 4-7 sync {100 * m[4:, 0].mean().item() / tot:4.1f} % body {100 * m[4:, 1].mean().item() / tot:4.1f} %)")


def lin(name, M, N, K, res=False):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    r = torch.zeros(M, N, device=DEV, dtype=torch.float16) if res else None
    for _ in range(3):
        ops.gemm(a, w, out, M=M, N=N, K=K, bias=torch.zeros(N, device=DEV), res1=r)
    torch.cuda.synchronize()
    show(name, out)


def conv(name, H, W, cin, cout, nimg=28):
    M = nimg * H * W
    a = torch.randn(M, cin, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(cout, 9 * cin, device=DEV, dtype=torch.float16) * 0.02
    out = torch.empty(M, cout, device=DEV, dtype=torch.float16)
    for _ in range(3):
        ops.gemm(a, w, out, M=M, N=cout, K=9 * cin, bias=torch.zeros(cout, device=DEV), mode=ops.A_CONV3X3, Cin=cin, conv=(H, W, H, W, 1, 0))
    torch.cuda.synchronize()
    show(name, out)


conv("conv3x3 L1 640->640", 36, 64, 640, 640)
conv("conv3x3 L0 320->320", 72, 128, 320, 320)
lin("lin 32768x2560x5120", 32768, 2560, 5120)
lin("ffout L0 320x1280 +res", 258048, 320, 1280, res=True)
lin("qkv L1 1920x640", 64512, 1920, 640)
