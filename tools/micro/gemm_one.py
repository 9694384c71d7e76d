#!/usr/bin/env python3
"""One GEMM shape of the C2 forward a few times (for rocprofv3 --pmc passes): GEMM_CASE = conv (3x3, 36x64, 640 -> 640, K = 5760),
geglu (72x128 GEGLU, K = 320), proj (72x128 attention out-projection with residual, K = 320), ffout (72x128 FF-out, K = 1280)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
DEV = "cuda:0"
case = os.environ.get("GEMM_CASE", "conv")
z = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16) * 0.1   # noqa: E731
if case == "conv":
    H, W, C = 36, 64, 640
    M = 28 * H * W
    x, w, b = z(M, C), z(C, 9 * C), torch.zeros(C, device=DEV)
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(x, w, out, M=M, N=C, K=9 * C, bias=b, mode=ops.A_CONV3X3, Cin=C, conv=(H, W, H, W, 1, 0))   # noqa: E731
elif case == "geglu":
    M, C = 28 * 72 * 128, 320
    x, w, b = z(M, C), z(8 * C, C), torch.zeros(8 * C, device=DEV)
    out = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(x, w, out, M=M, N=8 * C, K=C, bias=b, geglu=80)   # noqa: E731
elif case == "proj":
    M, C = 28 * 72 * 128, 320
    x, w, b, r = z(M, C), z(C, C), torch.zeros(C, device=DEV), z(M, C)
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(x, w, out, M=M, N=C, K=C, bias=b, res1=r)   # noqa: E731
else:
    M, C = 28 * 72 * 128, 320
    x, w, b, r = z(M, 4 * C), z(C, 4 * C), torch.zeros(C, device=DEV), z(M, C)
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(x, w, out, M=M, N=C, K=4 * C, bias=b, res1=r)   # noqa: E731
for _ in range(6):
    fn()
torch.cuda.synchronize()
