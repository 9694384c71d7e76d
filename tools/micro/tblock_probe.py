#!/usr/bin/env python3
"""One launch of the one-kernel temporal attention against fp32 (small case, stderr visible), then timing at the 72x128 level
against the two launches it replaces."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from lkgd_amd import ops
from lkgd_amd.packing import pack_tblock, pack_tfront, pack_linear
from test_tblock_gpu import _weights, _ref
B, Fr, HW = (int(v) for v in os.environ.get("PROBE_SHAPE", "1,14,16").split(","))
wqkv, bqkv, wo, bo = _weights(1)
g = torch.Generator().manual_seed(2)
T = B * Fr * HW
x = (torch.randn(T, 320, generator=g) * 1.5 + 0.3).half()
ws = pack_tblock(wqkv, bqkv, wo).cuda()
out = torch.full((T, 320), float("nan"), dtype=torch.float16, device="cuda")
print("launch", flush=True)
ops.tattn_block(x.cuda(), ws, bo.cuda(), out, B, Fr, HW)
torch.cuda.synchronize()
print("done", flush=True)
if T <= 100000:
    ref = _ref(x, wqkv, bqkv, wo, bo, B, Fr, HW)
    err = (out.float().cpu() - ref).abs()
    print("max err", err.max().item(), "nan", torch.isnan(out).sum().item(), "ref scale", ref.abs().max().item())
    bad = err > 2.5e-2
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print("bad", int(bad.sum()), "rows", rows.numel(), rows[:16].tolist(), "cols", cols.numel(), cols[:16].tolist())
    if rows.numel():
        r = int(rows[0]); print("row", r, "got", out[r, :8].float().tolist(), "ref", ref[r, :8].tolist())
else:
    xd = x.cuda()
    def bench(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
    wf = pack_tfront(wqkv.half().cuda(), 5); wl = pack_linear(wo).cuda(); bq = bqkv.cuda(); bod = bo.cuda()
    att = torch.empty_like(xd); chain = torch.empty_like(xd)
    def two():
        ops.tattn_front(xd, wf, bq, att, B, Fr, HW, 5)
        ops.gemm(att, wl, chain, M=T, N=320, K=320, bias=bod, res1=xd)
    import time
    a0 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20): a0 @ a0
        torch.cuda.synchronize()
    for rep in range(2):
        print(f"one launch {bench(lambda: ops.tattn_block(xd, ws, bod, out, B, Fr, HW)):.3f} ms   front + out-projection {bench(two):.3f} ms")
    two()
    print("max diff vs two launches", (out.float() - chain.float()).abs().max().item())
