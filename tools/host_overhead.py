#!/usr/bin/env python3
"""How long does the HOST take to enqueue one UNet forward (Python + ctypes launches) vs the GPU to execute it?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from lkgd_amd import ops

dev = torch.device("cuda", 0)
unet = B.build_unet(dev, False)
ONLY = os.environ.get("LKGD_CASE")          # "0" / "1" / "2": run one case only (for rocprofv3 --stats of that case)
for ci, (frames, h, w, tag) in enumerate(((2, 8, 8, "2 frames 8x8 (launch-bound)"), (14, 72, 128, "C2 full"), (4, 72, 128, "4 frames (8-GPU slice)"), (-4, 72, 128, "1 CFG half x 4 frames (rank of 8)"),
                                            (-7, 72, 128, "1 CFG half x 7 frames (rank of 4)"),
                                            (-14, 72, 128, "1 CFG half x 14 frames (rank of 2)"))):
    if ONLY is not None and int(ONLY) != ci:
        continue
    cfgb = 2 if frames > 0 else 1      # negative frame count: one batch entry (what a CFG-parallel rank runs)
    frames = abs(frames)
    lat0, img, emb, ids = B.synthetic_inputs(dev, frames, h, w)
    tok = ops.prepare_unet_input(lat0.half(), img, 2, 700.0)
    if cfgb == 1:
        tok, emb, ids = tok[:tok.shape[0] // 2].contiguous(), emb[:1].contiguous(), ids[:1].contiguous()
    for _ in range(2):
        unet.forward_tokens(tok, cfgb, frames, h, w, 1.0, emb, ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        unet.forward_tokens(tok, cfgb, frames, h, w, 1.0, emb, ids)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{tag:26s}: host enqueue {t_host*1e3:7.1f} ms / forward, end-to-end {t_all*1e3:7.1f} ms / forward")
    # the same forward replayed from a recorded launch list (lkgd_amd/replay.py)
    from lkgd_amd import replay
    t_dev = torch.ones(cfgb, dtype=torch.float32, device=dev)
    with replay.record() as plan:
        plan.result, _ = unet.forward_tokens(tok, cfgb, frames, h, w, t_dev, emb.half().contiguous(), ids.float().contiguous())
    plan.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan.run()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{'':26s}  replayed    {t_host*1e3:7.1f} ms / forward, end-to-end {t_all*1e3:7.1f} ms / forward "
          f"({len(plan.calls)} launches)")
    del plan
