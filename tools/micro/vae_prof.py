#!/usr/bin/env python3
"""Temporal VAE decode of 14 x 576x1024 (random-init SVD VAE shapes) three times: run under `rocprofv3 --kernel-trace --stats`
for the per-kernel breakdown.  GPU box only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import unet as pu
from lkgd_amd import vae as pv
dev = torch.device("cuda:0")
with torch.device("meta"):
    v = pv.AutoencoderKLTemporalDecoder()
v = v.to(torch.float16).to_empty(device=dev)
pu.init_synthetic_weights_(v, seed=2)
z = torch.randn(14, 4, 72, 128, generator=torch.Generator().manual_seed(1)).half().to(dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = v.decode(z, num_frames=14).sample
    torch.cuda.synchronize(); print(f"decode {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", tuple(out.shape), flush=True)
