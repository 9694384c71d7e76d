#!/usr/bin/env python3
"""Time the ping-pong GEMM with one ingredient of its K-tile body removed (tools/micro/pp_knobs.sh builds the
variants).  One full round of tiles (M = 16384, N = 4096 -> 1024 tiles = 4 per CU), deep K."""
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
_lib.lib().lkgd_debug_set_gemm_variant(6)
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
for (M, N, K) in ((16384, 4096, 5120), (16384, 4096, 1280), (65536, 1024, 640)):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(a, w, out, M=M, N=N, K=K)
    best = 1e9
    for rep in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    tiles = (M // 256) * (N // 256)
    per_ktile_us = best * 1e3 / ((tiles / 256) * (K // 64))
    print("  %%6dx%%5dx%%5d  %%7.3f ms  %%7.1f TF/s   %%.3f us per K-tile (%%d cycles at 2.0 GHz; MFMA floor 2048)" %% (
        M, N, K, best, 2.0 * M * N * K / best / 1e9, per_ktile_us, per_ktile_us * 2000))
''' % REPO

for lib in sorted(glob.glob(os.path.join(HERE, "libpp_*.so"))):
    print(os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
