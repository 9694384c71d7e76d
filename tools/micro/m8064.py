#!/usr/bin/env python3
"""The 18x32 level of ONE CFG half (a rank of 2: M = 14 * 576 = 8064 rows) sits right under the M < 8192 rule of the
dispatcher: which tile program wins there?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import _lib, ops
from lkgd_amd.packing import pack_conv3x3
DEV = "cuda:0"
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
for M, H, W in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or ((8064, 18, 32), (32256, 36, 64), (4032, 18, 32), (16128, 36, 64)):
    C = 1280 if H == 18 else 640
    x = torch.randn(M, C, device=DEV, dtype=torch.float16) * 0.1
    cases = {}
    w = torch.randn(C, 9 * C, device=DEV, dtype=torch.float16) * 0.02
    b = torch.zeros(C, device=DEV)
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    cases["conv3x3"] = (lambda: ops.gemm(x, w, out, M=M, N=C, K=9 * C, bias=b, mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0)), 2.0 * M * C * 9 * C)
    wl = torch.randn(C, C, device=DEV, dtype=torch.float16) * 0.03
    cases["proj+res"] = (lambda: ops.gemm(x, wl, out, M=M, N=C, K=C, bias=b, res1=x), 2.0 * M * C * C)
    wq = torch.randn(3 * C, C, device=DEV, dtype=torch.float16) * 0.03
    oq = torch.empty(M, 3 * C, device=DEV, dtype=torch.float16)
    cases["qkv"] = (lambda: ops.gemm(x, wq, oq, M=M, N=3 * C, K=C), 2.0 * M * 3 * C * C)
    x4 = torch.randn(M, 4 * C, device=DEV, dtype=torch.float16) * 0.1
    wf = torch.randn(C, 4 * C, device=DEV, dtype=torch.float16) * 0.02
    cases["ffout+res"] = (lambda: ops.gemm(x4, wf, out, M=M, N=C, K=4 * C, bias=b, res1=x), 2.0 * M * C * 4 * C)
    for name, (fn, fl) in cases.items():
        res = {}
        for rep in range(2):
            for v in (0, 1, 3, 4):
                _lib.lib().lkgd_debug_set_gemm_variant(v)
                fn(); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                res[v] = min(res.get(v, 1e9), s.elapsed_time(e) / 10)
        _lib.lib().lkgd_debug_set_gemm_variant(0)
        print("M=%6d C=%4d %-10s auto %.3f  t128 %.3f  stream %.3f  wide %.3f  (best %.0f TF/s)" % (
            M, C, name, res[0], res[1], res[3], res[4], fl / min(res.values()) / 1e9))
