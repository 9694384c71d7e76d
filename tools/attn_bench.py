#!/usr/bin/env python3
"""Spatial / temporal attention kernel throughput at the C2 shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lkgd_amd import ops

DEV = "cuda:0"
if os.environ.get("ATTN_WAVES"):        # A/B knob: 4, 8 or 16 waves (128 / 256 / 512 queries) per workgroup
    from lkgd_amd import _lib
    _lib.lib().lkgd_debug_set_attn_waves(int(os.environ["ATTN_WAVES"]))
if os.environ.get("ATTN_KVB"):          # A/B knob: 64 or 128 keys staged per barrier
    from lkgd_amd import _lib
    _lib.lib().lkgd_debug_set_attn_kvb(int(os.environ["ATTN_KVB"]))
if os.environ.get("ATTN_PIPE"):         # 1 = compiler-scheduled kernel everywhere, 2 = software-pipelined program wherever legal
    from lkgd_amd import _lib
    _lib.lib().lkgd_debug_set_attn_pipe(int(os.environ["ATTN_PIPE"]))


def bench(fn, iters=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)      # clock ramp: isolated kernels otherwise read ~15 % slow
import time
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20):
        a0 @ a0
    torch.cuda.synchronize()
tot_ms = tot_f = 0.0
SHAPES = ((28, 5, 9216, 5), (28, 10, 2304, 5), (28, 20, 576, 5), (28, 20, 144, 1), (2, 30, 17776, 30))
if os.environ.get("ATTN_ONLY"):         # the two levels the software-pipelined program serves
    SHAPES = SHAPES[:2]
for (nb, heads, S, layers) in SHAPES:
    C = heads * 64
    qkv = torch.randn(nb * S, 3 * C, device=DEV, dtype=torch.float16)
    out = torch.empty(nb * S, C, device=DEV, dtype=torch.float16)
    ms = bench(lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb, S, heads))
    fl = 4.0 * S * S * 64 * heads * nb
    print(f"spatial S={S:5d} heads={heads:2d}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s  (x{layers})")
    tot_ms += ms * layers
    tot_f += fl * layers
print(f"spatial total per forward: {tot_ms:.2f} ms, {tot_f/tot_ms/1e9:.1f} TFLOP/s")
for (B, F, S, heads, layers) in (() if os.environ.get("ATTN_ONLY") else ((2, 14, 9216, 5, 5), (2, 14, 2304, 10, 5), (2, 14, 576, 20, 5), (2, 14, 144, 20, 1))):
    C = heads * 64
    qkv = torch.randn(B * F * S, 3 * C, device=DEV, dtype=torch.float16)
    out = torch.empty(B * F * S, C, device=DEV, dtype=torch.float16)
    ms = bench(lambda: ops.attn_temporal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, B, F, S, heads))
    gb = B * F * S * C * 2 * 4 / 1e9
    print(f"temporal S={S:5d} heads={heads:2d}: {ms:7.3f} ms  {gb/ms*1e3:7.1f} GB/s (x{layers})")
