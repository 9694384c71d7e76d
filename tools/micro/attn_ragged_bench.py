#!/usr/bin/env python3
"""Spatial attention at S % 128 != 0 (CogVideoX-2B: 17 776 tokens x 30 heads): masked software-pipelined statement vs the
compiler-scheduled kernel.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import _lib, ops
L = _lib.lib()
S, heads, nb = int(os.environ.get("ATTN_S", "17776")), int(os.environ.get("ATTN_HEADS", "30")), int(os.environ.get("ATTN_NB", "2"))
C = heads * 64
g = torch.Generator().manual_seed(1)
qkv = torch.randn(nb * S, 3 * C, generator=g).half().cuda()
out = torch.empty(nb * S, C, dtype=torch.float16, device="cuda")
a0 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
def bench(n=5):
    f = lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb, S, heads)
    for _ in range(2): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
flop = 4.0 * S * S * 64 * heads * nb
for rep in range(2):
    for mode, name in ((1, "compiler-scheduled"), (0, "by rule (masked statement)")):
        L.lkgd_debug_set_attn_pipe(mode)
        ms = bench()
        print(f"S={S} heads={heads} batch={nb}  {name:28s} {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s")
L.lkgd_debug_set_attn_pipe(0)
