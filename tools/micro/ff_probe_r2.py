#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from lkgd_amd import ops
from lkgd_amd.packing import pack_ff_fused
T = int(os.environ.get("PROBE_T", "300"))
g = torch.Generator().manual_seed(1)
w1 = torch.randn(2560, 320, generator=g) / 320 ** 0.5
b1 = 0.3 * torch.randn(2560, generator=g)
w2 = torch.randn(320, 1280, generator=g) / 1280 ** 0.5
b2 = 0.3 * torch.randn(320, generator=g)
x = torch.randn(T, 320, generator=g).half()
res2 = torch.randn(T, 320, generator=g).half()
ws = pack_ff_fused(w1, b1, w2).cuda()
out = torch.full((T, 320), float("nan"), dtype=torch.float16, device="cuda")
ops.ff_fused(x.cuda(), ws, b2.cuda(), out, s_acc=0.3, res2=res2.cuda(), r2=0.7)
torch.cuda.synchronize()
xf = x.float()
z = F.layer_norm(xf, (320,), None, None, 1e-5)
hg = z @ w1.half().float().T + b1
y = (hg[:, :1280] * F.gelu(hg[:, 1280:])) @ w2.half().float().T + b2
ref = 0.3 * (y + xf) + 0.7 * res2.float()
noR = 0.3 * (y + xf)
err = (out.float().cpu() - ref).abs()
print("max err", err.max().item(), "vs no-res2 ref:", (out.float().cpu() - noR).abs().max().item())
d = (out.float().cpu() - noR) / 0.7          # what was added as res2
bad = err > 2e-2
rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
print("bad", int(bad.sum()), "rows", rows.numel(), rows[:12].tolist(), "cols", cols.numel(), cols[:24].tolist())
r = int(rows[0]) if rows.numel() else 0
print("row", r, "added:", d[r, :16].tolist())
print("res2  :", res2[r, :16].float().tolist())
