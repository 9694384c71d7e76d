#!/usr/bin/env python3
"""256x320 program over forced K slices on the 72x128-level shapes of a rank of 8 (144 tiles) and of 4 (252 tiles)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("SHARD", "1,4")
import gemm_shapes_bench as G
from lkgd_amd import _lib
L = _lib.lib()
G.warm(2.0)
want = ("conv3x3 L0 320->320", "conv3x3 L0 640->320", "conv3x3 L0 960->320", "tconv L0 320", "conv3x3 L1 640->640", "conv3x3 L1 1280->640",
        "tconv L1 640", "lin L1 ffout 640x2560", "lin L0 ffout 320x1280", "conv3x3 up L1->L0 640")
print("SHARD", os.environ["SHARD"])
print(f"{'shape':28s} " + " ".join(f"{'auto' if k < 0 else 'ks=%d' % k:>8s}" for k in (-1, 1, 2, 3, 4, 5, 6)))
for name, cnt, kind, d in G.shapes():
    if name not in want:
        continue
    row = []
    for k in (-1, 1, 2, 3, 4, 5, 6):
        L.lkgd_debug_set_gemm_variant(0 if k < 0 else 4)
        L.lkgd_debug_set_wide_ksplit(0 if k < 0 else k)
        L.lkgd_debug_set_gemm_splitk(0 if k == 1 else 1)
        best = 1e9
        for _ in range(3):
            try:
                best = min(best, G.run(kind, d, iters=10)[1])
            except Exception:
                best = float("nan")
        row.append(best * 1e3)
    L.lkgd_debug_set_gemm_variant(0); L.lkgd_debug_set_wide_ksplit(0); L.lkgd_debug_set_gemm_splitk(1)
    print(f"{name:28s} " + " ".join(f"{v:8.1f}" for v in row), flush=True)
