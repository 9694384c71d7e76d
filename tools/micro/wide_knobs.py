#!/usr/bin/env python3
"""Time the 256x320 GEMM on the short-K projections of the 72x128 level with one ingredient changed
(tools/micro/wide_knobs.sh builds the variants): where does the time of a K = 320 tile go?"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
from lkgd_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from lkgd_amd import ops
_lib.lib().lkgd_debug_set_gemm_variant(int(os.environ.get('VARIANT', '4')))
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
M0 = 28 * 72 * 128
for (name, M, N, K, geglu, res) in (("L0 geglu", M0, 2560, 320, True, False), ("L0 ffout", M0, 320, 1280, False, True),
                                    ("L0 qkv", M0, 960, 320, False, False), ("L1 geglu", M0 // 4, 5120, 640, True, False),
                                    ("L2 ffout", M0 // 16, 1280, 5120, False, True),
                                    ("deep K", 32768, 2560, 5120, False, False),
                                    ("L1 qkv", M0 // 4, 1920, 640, False, False), ("L1 proj", M0 // 4, 640, 640, False, True),
                                    ("L2 qkv", M0 // 16, 3840, 1280, False, False)):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    # (GEGLU weights would be row-permuted by lkgd_amd.packing; irrelevant for the time)
    out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
    r = torch.randn_like(out) if res else None
    fn = lambda: ops.gemm(a, w, out, M=M, N=N, K=K, geglu=80 if geglu else 0, res1=r)
    best = 1e9
    for rep in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    print("  %%-9s %%7dx%%5dx%%5d  %%7.3f ms  %%7.1f TF/s" %% (name, M, N, K, best, 2.0 * M * N * K / best / 1e9))
''' % REPO

if "--single" in sys.argv:          # tools/micro/lib_ab.py: time the library LKGD_HIP_LIB points at
    subprocess.run([sys.executable, "-c", CHILD, os.environ["LKGD_HIP_LIB"]], check=False)
    sys.exit(0)
for lib in sorted(glob.glob(os.path.join(HERE, "libwide_*.so"))):
    print(os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
