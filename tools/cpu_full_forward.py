#!/usr/bin/env python3
"""ONE full BASELINE.json configs[1] UNet forward of the fp32 CPU oracle (CFG batch 2 x 14 frames x 72 x 128 latent, real
width: 89.69 algorithmic TFLOP) on the host cores of the box it runs on - the reference's CPU path as SURVEY.md 8d / BASELINE.md
2.4 plan to time it.  Takes ~10-15 minutes on the GPU box's host; prints a progress line per block (so the run is never
silent) and writes gpurun_out/cpu_full_forward.json; the committed copy is profiles/r02_cpu_full_forward.json, which bench.py
quotes next to its bounded live sample."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from oracle import unet as ou
from tools.hostcpus import cpu_facts, host_cpus

# the threads this process can really run (the pool's GPU boxes: a cgroup quota of 16 behind 256 logical CPUs; torch's default of
# 128 intra-op threads is several times slower there).  LKGD_CPU_THREADS overrides.
torch.set_num_threads(int(os.environ.get("LKGD_CPU_THREADS", host_cpus())))


def run_full_forward(F=14, H=72, W=128, verbose=True):
    """builds the fp32 oracle UNet, runs ONE forward on the host cores, returns the result record"""
    t0 = time.time()
    with torch.device("meta"):
        o = ou.UNetSpatioTemporalConditionControlNetModel(ou.SVD_CONFIG)
    o = o.to_empty(device="cpu")
    with torch.no_grad():
        for p in o.parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 1.0 / p[0].numel() ** 0.5)
            else:
                p.fill_(0.0)
        for m in o.modules():
            if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm)):
                m.weight.fill_(1.0)
    out = sys.stderr if not verbose else sys.stdout
    print(f"model built in {time.time() - t0:.0f} s; threads {torch.get_num_threads()}, os.cpu_count {os.cpu_count()}", file=out, flush=True)
    t_start = [time.time()]

    def hook(name):
        def f(mod, inp, res):
            print(f"  {name} done at {time.time() - t_start[0]:.0f} s", file=out, flush=True)
        return f

    for i, b in enumerate(o.down_blocks):
        b.register_forward_hook(hook(f"down_blocks.{i}"))
        for j, r in enumerate(b.resnets):
            r.register_forward_hook(hook(f"down_blocks.{i}.resnets.{j}"))
    o.mid_block.register_forward_hook(hook("mid_block"))
    for i, b in enumerate(o.up_blocks):
        b.register_forward_hook(hook(f"up_blocks.{i}"))
        for j, r in enumerate(b.resnets):
            r.register_forward_hook(hook(f"up_blocks.{i}.resnets.{j}"))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, F, 8, H, W, generator=g)
    enc = torch.randn(2, 1, 1024, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    t_start[0] = time.time()
    with torch.no_grad():
        y = o(x, torch.tensor(1.0), enc, added_time_ids=ids, return_dict=False)[0]
    dt = time.time() - t_start[0]
    tflop = 89.69 * (F / 14.0) * (H * W) / (72 * 128)          # algorithmic, SURVEY.md App. B (attention term scaled linearly: a bound)
    return {"workload": f"one UNet forward, CFG batch 2 x {F} frames x {H}x{W} latent, real-width SVD UNet, fp32 oracle (torch eager)",
            "seconds": round(dt, 1), "algorithmic_tflop": round(tflop, 2), "tflops": round(tflop / dt, 4),
            "threads": torch.get_num_threads(), **cpu_facts(), "finite": bool(torch.isfinite(y).all()),
            "frames_per_s_c2_equivalent": round(14.0 / (25.0 * dt * (14.0 / F) * (72 * 128) / (H * W)), 6)}


if __name__ == "__main__":
    # LKGD_PROGRESS_STDERR=1 (bench.py's child): progress lines on stderr, stdout carries the JSON record only
    res = run_full_forward(int(os.environ.get("FRAMES", "14")), int(os.environ.get("LAT_H", "72")), int(os.environ.get("LAT_W", "128")),
                           verbose=os.environ.get("LKGD_PROGRESS_STDERR", "0") != "1")
    print(json.dumps(res), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "cpu_full_forward.json"), "w") as f:
        json.dump(res, f, indent=1)
