#!/usr/bin/env python3
"""GPU probe of the recorded forward's arena (lkgd_amd/replay.py): (1) the torch ops inside the recorded regions of every loop
variant that break the replay contract (LKGD_REPLAY_STRICT=log), (2) replay from the arena == eager, bit for bit, over N Euler
steps, (3) arena / peak memory of the full-size headline clip.  `python tools/replay_probe.py [--full]`"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from lkgd_amd import replay
from lkgd_amd.pipeline import StableVideoDiffusionPipeline


def variant(name, tiny=True, frames=4, h=16, w=16, steps=3):
    dev = torch.device("cuda", 0)
    lk = name in ("lk", "lkjoint")
    unet = bench.build_unet(dev, tiny, lk)
    pipe = StableVideoDiffusionPipeline(unet=unet)
    lat0, img, emb, ids = bench.synthetic_inputs(dev, frames, h, w)
    if tiny:
        emb = emb[..., :1024]
    dom = flow = None
    if lk:
        dom = torch.randn(1, 1, 1000, generator=torch.Generator().manual_seed(1)).half().to(dev)
        flow = torch.randn(1, 1, 1000, generator=torch.Generator().manual_seed(2)).half().to(dev)
    if name in ("joint", "lkjoint"):
        from lkgd_amd import patch
        patch.apply_patch(pipe, with_temporal_block=True)
        patch.initialize_joint_layers(pipe)
        with torch.no_grad():
            g = torch.Generator().manual_seed(12350)
            for n_, prm in unet.named_parameters():
                if "attn1n" in n_ or "conv1n" in n_:
                    prm.copy_((torch.randn(prm.shape, generator=g) * (0.5 / max(prm.shape[-1], 1) ** 0.5)).to(prm))
        unet.invalidate()
        patch.set_joint_attention_mask(pipe, [0, 1, 0, 1])
        lat0 = torch.cat([lat0, 0.9 * lat0.flip(1)])
        img = torch.stack([img[0], img[0], img[1], 0.8 * img[1]])
        emb = torch.stack([emb[0], emb[0], emb[1], 0.8 * emb[1]])
        ids = ids[:1].repeat(4, 1)
        if dom is not None:
            dom, flow = torch.cat([dom, 0.7 * dom] * 2), torch.cat([flow, 0.6 * flow] * 2)
    ctrl = None
    if name == "controlnet":
        from lkgd_amd import controlnet as pc
        from lkgd_amd import unet as pu
        with torch.device("meta"):
            cn = pc.ControlNetSDVModel(pu.UNetConfig(**{k: v for k, v in unet.config.__dict__.items()
                                                        if k in pu.UNetConfig.__dataclass_fields__}))
        cn = cn.to(torch.float16).to_empty(device=dev)
        pu.init_synthetic_weights_(cn, seed=1)
        pipe.controlnet = cn
        ctrl = (2.0 * torch.rand(1, frames, 3, 8 * h, 8 * w, generator=torch.Generator().manual_seed(3)) - 1.0).half().to(dev).repeat(2, 1, 1, 1, 1)
    pipe.scheduler.set_timesteps(steps)
    s0 = float(pipe.scheduler.init_noise_sigma)

    def run():
        return pipe.denoise((lat0 * s0).half(), img, emb, ids, steps, 1.0, 3.0, domain_features=dom, flow_features=flow,
                            controlnet_condition=ctrl)
    return pipe, run


def main():
    full = "--full" in sys.argv
    replay.STRICT = "log"
    for name in ("stock", "lk", "joint", "lkjoint", "controlnet"):
        pipe, run = variant(name)
        del replay.VIOLATIONS[:]
        pipe.use_replay = True
        a = run().float().cpu()
        a2 = run().float().cpu()           # second call: the arena's cached blocks
        viol = collections.Counter(replay.VIOLATIONS)
        pipe.use_replay = False
        b = run().float().cpu()
        print(f"{name:10s} replay==eager: {torch.equal(a, b)}  second call == first: {torch.equal(a, a2)}  "
              f"arena {pipe.arena_reserved_bytes() / 1e6:.1f} MB  violations: {len(viol)}", flush=True)
        for (op, where), n in sorted(viol.items(), key=lambda kv: -kv[1]):
            print(f"    {n:4d} x {op:24s} {where}")
    if not full:
        return
    replay.STRICT = False
    for name in ("stock", "joint", "controlnet"):
        pipe, run = variant(name, tiny=False, frames=14, h=72, w=128, steps=4)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for mode in (True, False):
            pipe.use_replay = mode
            torch.cuda.reset_peak_memory_stats()
            outs, times = [], []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                outs.append(run())
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            print(f"FULL {name:10s} replay={mode}: 4-step clips {[round(t * 1e3, 1) for t in times]} ms, peak allocated "
                  f"{torch.cuda.max_memory_allocated() / 1e9:.2f} GB (resident before: {base / 1e9:.2f}), reserved "
                  f"{torch.cuda.max_memory_reserved() / 1e9:.2f} GB, arena {pipe.arena_reserved_bytes() / 1e9:.2f} GB, "
                  f"same result over calls: {torch.equal(outs[0], outs[2])}", flush=True)
            ref = outs[0] if mode else ref_replay
            if mode:
                ref_replay = outs[0]
            else:
                print(f"FULL {name:10s} replay == eager bitwise: {torch.equal(ref_replay, outs[0])}", flush=True)
        del pipe, run, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
