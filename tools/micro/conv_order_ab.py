#!/usr/bin/env python3
"""3x3-conv shapes of the C2 forward under two builds of the library (LKGD_HIP_LIB A/B in child processes):
tools/micro/libold_convorder.so (tap-major K order) vs the in-tree build (kx-inner K order).  Timing only."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import gemm_shapes_bench as g
g.warm(2.0)
tot = 0.0
for name, cnt, kind, d in g.shapes():
    if kind != "conv": continue
    best = 1e9
    for _ in range(3):
        flop, ms = g.run(kind, d, iters=8)
        best = min(best, ms)
    tot += best * cnt
    print(f"  {name:30s} {cnt:3d} {best:7.3f} ms {flop / best / 1e9:7.0f} TF/s", flush=True)
print(f"  conv total {tot:.2f} ms")
''' % (REPO, REPO)
for lib in (os.path.join(HERE, "libold_convorder.so"), os.path.join(REPO, "lkgd_amd", "liblkgd_hip.so")):
    print(lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LKGD_HIP_LIB=lib), check=False)
