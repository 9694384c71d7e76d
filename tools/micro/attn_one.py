#!/usr/bin/env python3
"""One spatial-attention shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
DEV = "cuda:0"
nb, heads, S = 28, 5, 9216
C = heads * 64
qkv = torch.randn(nb * S, 3 * C, device=DEV, dtype=torch.float16)
out = torch.empty(nb * S, C, device=DEV, dtype=torch.float16)
for _ in range(4):
    ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb, S, heads)
torch.cuda.synchronize()
