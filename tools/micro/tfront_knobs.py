#!/usr/bin/env python3
"""Time lkgd_tattn_front at the 72x128 level with one ingredient removed (tools/micro/tfront_knobs.sh builds the variants)."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch
from lkgd_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from lkgd_amd import ops
from lkgd_amd.packing import pack_tfront
DEV = "cuda:0"
B, Fr, HW, C, heads = 2, 14, 72 * 128, 320, 5
T = B * Fr * HW
x = (torch.randn(T, C, device=DEV) * 1.5).half()
w = (torch.randn(3 * C, C, device=DEV) / C ** 0.5).half()
b = torch.randn(3 * C, device=DEV) * 0.1
wf = pack_tfront(w, heads)
out = torch.empty(T, C, dtype=torch.float16, device=DEV)
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
for _ in range(60): a0 @ a0
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    ops.tattn_front(x, wf, b, out, B, Fr, HW, heads); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.tattn_front(x, wf, b, out, B, Fr, HW, heads)
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 10)
print(f"  {best:.3f} ms")
''' % REPO
for lib in sorted(glob.glob(os.path.join(HERE, "libtf_*.so"))):
    print(os.path.basename(lib), end="", flush=True)
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
