#!/usr/bin/env python3
"""One launch of the fused feed-forward against fp32 (stderr visible), then timing against the three-launch chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from lkgd_amd import ops
from lkgd_amd.packing import pack_ff_fused, pack_geglu, pack_linear
T = int(os.environ.get("PROBE_T", "128"))
g = torch.Generator().manual_seed(1)
w1 = torch.randn(2560, 320, generator=g) / 320 ** 0.5
b1 = 0.3 * torch.randn(2560, generator=g)
w2 = torch.randn(320, 1280, generator=g) / 1280 ** 0.5
b2 = 0.3 * torch.randn(320, generator=g)
x = (torch.randn(T, 320, generator=g) * 1.5 + 0.3).half()
ws = pack_ff_fused(w1, b1, w2).cuda()
out = torch.full((T, 320), float("nan"), dtype=torch.float16, device="cuda")
print("launch", flush=True)
ops.ff_fused(x.cuda(), ws, b2.cuda(), out)
torch.cuda.synchronize()
print("done", flush=True)
xf = x.float()
z = F.layer_norm(xf, (320,), None, None, 1e-5)
hg = z @ w1.half().float().T + b1
ref = (hg[:, :1280] * F.gelu(hg[:, 1280:])) @ w2.half().float().T + b2 + xf
err = (out.float().cpu() - ref).abs()
print("T", T, "max err", err.max().item(), "nan", torch.isnan(out).sum().item(), "ref scale", ref.abs().max().item())
bad = err > 2e-2
rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
print("bad", int(bad.sum()), "rows", rows.numel(), rows[:10].tolist(), "cols", cols.numel(), cols[:16].tolist())
if T >= 128 * 256:
    import time
    xd = x.cuda()
    def bench(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
    wp, bp, half = pack_geglu(w1.cuda(), b1.cuda())
    wo = pack_linear(w2).cuda()
    mid = torch.empty(T, 1280, dtype=torch.float16, device="cuda"); ln = torch.empty_like(xd); chain = torch.empty_like(xd)
    b2d = b2.cuda()
    def three():
        ops.layernorm(xd, None, None, 1e-5, out=ln)
        ops.gemm(ln, wp, mid, M=T, N=2560, K=320, bias=bp, geglu=half)
        ops.gemm(mid, wo, chain, M=T, N=320, K=1280, bias=b2d, res1=xd)
    a0 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20): a0 @ a0
        torch.cuda.synchronize()
    for rep in range(2):
        print(f"fused {bench(lambda: ops.ff_fused(xd, ws, b2d, out)):.3f} ms   three launches {bench(three):.3f} ms")
