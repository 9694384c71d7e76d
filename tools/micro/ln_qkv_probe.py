#!/usr/bin/env python3
"""One launch of the fused LayerNorm + QKV projection against fp32 (stderr visible), then timing at the 72x128 level against the
row-panel GEMM with the LayerNorm fold and against LayerNorm + plain GEMM."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from lkgd_amd import ops
from lkgd_amd.packing import pack_ln_proj, pack_linear
from test_ln_qkv_gpu import _weights, _ref
T = int(os.environ.get("PROBE_T", "128"))
C = int(os.environ.get("PROBE_C", "320"))
w, b = _weights(1, C)
g = torch.Generator().manual_seed(2)
x = (torch.randn(T, C, generator=g) * 1.5 + 0.3).half()
ws = pack_ln_proj(w, b).cuda()
out = torch.full((T, 3 * C), float("nan"), dtype=torch.float16, device="cuda")
print("launch", flush=True)
ops.ln_qkv(x.cuda(), ws, out)
torch.cuda.synchronize()
print("done", flush=True)
if T <= 60000:
    ref = _ref(x, w, b)
    err = (out.float().cpu() - ref).abs()
    print("T", T, "max err", err.max().item(), "nan", torch.isnan(out).sum().item(), "ref scale", ref.abs().max().item())
    bad = err > 2e-2
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print("bad", int(bad.sum()), "rows", rows.numel(), rows[:10].tolist(), "cols", cols.numel(), cols[:16].tolist())
else:
    xd = x.cuda()
    def bench(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
    wp = pack_linear(w).cuda(); cs = wp.float().sum(dim=1).contiguous(); bd = b.cuda()
    chain = torch.empty_like(out); ln = torch.empty_like(xd)
    def folded():
        if C == 320: ops.gemm(xd, wp, chain, M=T, N=960, K=320, bias=bd, ln=(cs, 1e-5))
    def two():
        ops.layernorm(xd, None, None, 1e-5, out=ln)
        ops.gemm(ln, wp, chain, M=T, N=3 * C, K=C, bias=bd)
    a0 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20): a0 @ a0
        torch.cuda.synchronize()
    for rep in range(2):
        print(f"one launch {bench(lambda: ops.ln_qkv(xd, ws, out)):.3f} ms   folded row-panel GEMM {bench(folded):.3f} ms   LayerNorm + GEMM {bench(two):.3f} ms")
