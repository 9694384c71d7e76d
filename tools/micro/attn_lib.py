#!/usr/bin/env python3
"""Time the L0 / L1 spatial attention shapes with alternative library builds (tools/micro/libatt_*.so)."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import torch
from lkgd_amd import _lib
if sys.argv[1] != "default": _lib.LIB_PATH = sys.argv[1]
from lkgd_amd import ops
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
for (nb, heads, S) in ((28, 5, 9216), (28, 10, 2304)):
    C = heads * 64
    qkv = torch.randn(nb * S, 3 * C, device=DEV, dtype=torch.float16)
    out = torch.empty(nb * S, C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb, S, heads)
    best = 1e9
    for rep in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    print("  S=%%5d heads=%%2d  %%7.3f ms  %%7.1f TF/s" %% (S, heads, best, 4.0 * S * S * 64 * heads * nb / best / 1e9))
''' % REPO
for lib in ["default"] + sorted(glob.glob(os.path.join(HERE, "libatt_*.so"))):
    print(os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
