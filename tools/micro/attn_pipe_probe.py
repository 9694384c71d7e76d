#!/usr/bin/env python3
"""One small launch of the software-pipelined attention program against fp32 (stderr visible: no pytest capture)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from lkgd_amd import _lib, ops

S = int(os.environ.get("PROBE_S", "128"))
heads, nb = int(os.environ.get("PROBE_HEADS", "1")), int(os.environ.get("PROBE_NB", "1"))
C = heads * 64
g = torch.Generator().manual_seed(1)
q, k, v = (torch.randn(nb * S, C, generator=g).half().cuda() for _ in range(3))
out = torch.full((nb * S, C), float("nan"), dtype=torch.float16, device="cuda")
_lib.lib().lkgd_debug_set_attn_pipe(2)
print("launch", flush=True)
ops.attn_spatial(q, k, v, out, nb, S, heads)
torch.cuda.synchronize()
print("done", flush=True)
qf, kf, vf = (t.float().cpu().reshape(nb, S, heads, 64).transpose(1, 2) for t in (q, k, v))
ref = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(nb * S, C)
err = (out.float().cpu() - ref).abs()
print("S", S, "max err", err.max().item(), "nan", torch.isnan(out).sum().item(), "ref scale", ref.abs().max().item())
bad = (err > 5e-3)
rows = bad.any(1).nonzero().flatten()
print("bad entries", int(bad.sum()), "bad rows", rows.numel(), "first", rows[:8].tolist(), "last", rows[-8:].tolist())
if rows.numel():
    import collections
    print("bad rows by (row // 32):", sorted(collections.Counter((rows // 32).tolist()).items()))
    r0 = int(rows[0])
    print("row", r0, "bad cols", bad[r0].nonzero().flatten().tolist()[:70])
    print(" got", out[r0, :8].float().cpu().tolist(), "\n ref", ref[r0, :8].tolist())
