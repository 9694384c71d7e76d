#!/usr/bin/env python3
"""Do the two CFG halves of a batch (same rows twice) leave a kernel bitwise equal?  (test_full_size_loop_properties (2).)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
from lkgd_amd.packing import pack_ff_fused
g = torch.Generator().manual_seed(1)
w1 = torch.randn(2560, 320, generator=g) / 320 ** 0.5
b1 = 0.3 * torch.randn(2560, generator=g)
w2 = torch.randn(320, 1280, generator=g) / 1280 ** 0.5
b2 = 0.3 * torch.randn(320, generator=g)
ws = pack_ff_fused(w1, b1, w2).cuda()
Th = 14 * 9216
x0 = (torch.randn(Th, 320, generator=g) * 1.5).half().cuda()
x = torch.cat([x0, x0])
out = torch.empty_like(x)
ops.ff_fused(x, ws, b2.cuda(), out)
d = (out[:Th] != out[Th:])
print("ff_fused plain: rows that differ", int(d.any(1).sum()), "first", d.any(1).nonzero().flatten()[:8].tolist(),
      "max diff", (out[:Th].float() - out[Th:].float()).abs().max().item())
r2 = torch.randn(Th, 320, generator=g).half().cuda()
r2 = torch.cat([r2, r2])
ops.ff_fused(x, ws, b2.cuda(), out, s_acc=0.3, res2=r2, r2=0.7)
d = (out[:Th] != out[Th:])
print("ff_fused blend: rows that differ", int(d.any(1).sum()), "first", d.any(1).nonzero().flatten()[:8].tolist())
# spatial attention, S = 9216, 5 heads, batch 2 x 2 frames
S, Hh = 9216, 5
q0 = torch.randn(2 * S, 3 * 320, generator=g).half().cuda()
qkv = torch.cat([q0, q0])
o = torch.empty(4 * S, 320, dtype=torch.float16, device="cuda")
ops.attn_spatial(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], o, 4, S, Hh)
d = (o[:2 * S] != o[2 * S:])
print("attention: rows that differ", int(d.any(1).sum()))
pos = (torch.randn(14, 320, generator=g) * 0.5).half().cuda()
ops.ff_fused(x, ws, b2.cuda(), out, 1e-5, pos, ops.rowmap_div_mod(9216, 14))
d = (out[:Th] != out[Th:])
rows = d.any(1).nonzero().flatten()
print("ff_fused row bias: rows that differ", rows.numel(), "first", rows[:8].tolist(), "panels", sorted(set((rows // 128).tolist()))[:12],
      "max diff", (out[:Th].float() - out[Th:].float()).abs().max().item())
