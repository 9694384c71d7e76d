#!/usr/bin/env python3
"""Many launches of the software-pipelined attention program in one process, fresh random data each time, vs fp32."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from lkgd_amd import _lib, ops

_lib.lib().lkgd_debug_set_attn_pipe(2)
for S in [int(x) for x in os.environ.get("PROBE_SS", "640,768,896,1024,1152,640,512,384").split(",")]:
    for heads, nb in ((1, 1), (2, 1)):
        for rep in range(2):
            C = heads * 64
            q, k, v = (torch.randn(nb * S, C).half().cuda() for _ in range(3))
            out = torch.full((nb * S, C), float("nan"), dtype=torch.float16, device="cuda")
            ops.attn_spatial(q, k, v, out, nb, S, heads)
            torch.cuda.synchronize()
            qf, kf, vf = (t.float().cpu().reshape(nb, S, heads, 64).transpose(1, 2) for t in (q, k, v))
            ref = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(nb * S, C)
            err = (out.float().cpu() - ref).abs()
            bad = (err > 5e-3)
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print(f"S={S} heads={heads} rep={rep} max err {err.max().item():.4g} bad rows {rows.numel()}"
                  f" tiles {sorted(set((rows // 32).tolist()))[:20]} cols {cols.numel()}", flush=True)
