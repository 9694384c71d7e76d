#!/usr/bin/env python3
"""A few launches of lkgd_tattn_front at the 72x128 level (B=2, F=14, HW=9216, 5 heads) for rocprofv3 --pmc passes
(tools/micro/tfront_pmc.sh).  TFRONT_CASE=unfused runs the three launches the fused kernel replaces instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
from lkgd_amd.packing import pack_tfront
DEV = "cuda:0"
B, Fr, HW, C, heads = 2, 14, 72 * 128, 320, 5
T = B * Fr * HW
torch.manual_seed(0)
x = (torch.randn(T, C, device=DEV) * 1.5).half()
w = (torch.randn(3 * C, C, device=DEV) / C ** 0.5).half()
b = torch.randn(3 * C, device=DEV) * 0.1
wf = pack_tfront(w, heads)
out = torch.empty(T, C, dtype=torch.float16, device=DEV)
for _ in range(6):
    ops.tattn_front(x, wf, b, out, B, Fr, HW, heads)
torch.cuda.synchronize()
print("ok")
