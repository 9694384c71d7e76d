#!/usr/bin/env python3
"""Time the resident-weight GEMM on the K = 320 projections of the 72x128 level with one ingredient changed
(tools/micro/resw_knobs.sh builds the variants): where does the time of a 32-row block go?"""
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
_lib.lib().lkgd_debug_set_gemm_variant(int(os.environ.get('VARIANT', '6')))
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
M0 = 28 * 72 * 128
for (name, M, N, K, geglu, res) in (("L0 geglu", M0, 2560, 320, True, False), ("L0 qkv", M0, 960, 320, False, False),
                                    ("L0 proj", M0, 320, 320, False, True)):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
    r = torch.randn_like(out) if res else None
    b = torch.zeros(N, device=DEV) if (geglu or res) else None
    fn = lambda: ops.gemm(a, w, out, M=M, N=N, K=K, bias=b, geglu=80 if geglu else 0, res1=r)
    best = 1e9
    for rep in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    print(f"  {name:10s} {M:7d}x{N:5d}x{K:5d}   {best:6.3f} ms   {2.0 * M * N * K / best / 1e9:7.1f} TF/s", flush=True)
''' % REPO

for lib in sorted(glob.glob(os.path.join(HERE, "libresw_*.so"))):
    print(os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
