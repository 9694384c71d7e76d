#!/usr/bin/env python3
"""Temporal VAE decode of one clip (14 x 576x1024, one chunk; random-init SVD VAE shapes) on the HIP path: wall time per decode,
with and without the GroupNorm sums from the convolutions' epilogues (ops.COLSTATS), and the same decode's result under both
(they must agree to fp16 rounding of the statistics).  GPU box only.
    python tools/micro/vae_decode_time.py            # both settings, three timed decodes each
    PROFILE=1 ... under rocprofv3 --kernel-trace --stats for the per-kernel split"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops
from lkgd_amd import unet as pu
from lkgd_amd import vae as pv

DEV = "cuda:0"
F, h, w = 14, 72, 128
with torch.device("meta"):
    v = pv.AutoencoderKLTemporalDecoder()
v = v.to(torch.float16).to_empty(device=DEV)
pu.init_synthetic_weights_(v, seed=2)
z = (torch.randn(F, 4, h, w, generator=torch.Generator().manual_seed(3)) * 3.0).half().to(DEV)
a0 = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20): a0 @ a0
    torch.cuda.synchronize()
del a0
outs = {}
settings = (True,) if os.environ.get("PROFILE") else (False, True, False, True)
for cs in settings:
    ops.COLSTATS = cs
    r = v.decode(z, num_frames=F).sample
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 1 if os.environ.get("PROFILE") else 3
    for _ in range(reps):
        r = v.decode(z, num_frames=F).sample
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"epilogue GroupNorm sums {'on ' if cs else 'off'}: {dt * 1e3:7.2f} ms per decode, finite {bool(torch.isfinite(r.float()).all())}", flush=True)
    outs[cs] = r.float().cpu()
    del r
if len(outs) == 2:
    d = (outs[True] - outs[False])
    print(f"on vs off: rel L2 {float(d.norm() / outs[False].norm()):.3e}, max abs {float(d.abs().max()):.3e}")
