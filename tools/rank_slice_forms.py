#!/usr/bin/env python3
"""Compute of ONE rank's slice of a frame-sharded forward on one GPU, with the exchanges replaced by local stand-ins of the
right shapes (no bytes moved): the all-to-all pixel re-sharding of the temporal attention (default) against the all-gather
form (LKGD_TEMPORAL_GATHER=1).  Complements tools/host_overhead.py, whose slices are unsharded forwards of fewer frames.
Prints ms per forward for ranks of 8 (CFG x (4,4,3,3): the 4-frame rank), 4 (CFG x (7,7)) and 2 (CFG-parallel)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from lkgd_amd import dist as ld
from lkgd_amd import ops
from lkgd_amd.dist import make_plan, pixel_splits


class LoopbackShard:
    """lkgd_amd.dist_run.ShardInfo without a process group: every exchange returns a buffer of the shape the real one would"""

    def __init__(self, plan):
        self.plan, self.F_total, self.f0, self.B_total, self.b0 = plan, plan.num_frames, plan.f0, plan.cfg_groups, plan.cfg_index

    def gather(self, local):
        fl = self.plan.f_local
        x = local.reshape(fl, -1, local.shape[-1])
        reps = -(-self.F_total // fl)
        return x.repeat(reps, 1, 1)[: self.F_total].reshape(-1, local.shape[-1]).contiguous()

    def to_pixels(self, local, HW):
        fl, k, si = self.plan.f_local, self.plan.frame_shards, self.plan.shard_index
        px = pixel_splits(HW, k)
        x = local.reshape(fl, HW, -1)[:, : px[si]]
        reps = -(-self.F_total // fl)
        return x.repeat(reps, 1, 1)[: self.F_total].reshape(-1, local.shape[-1]).contiguous()

    def to_frames(self, x, HW):
        fl, F = self.plan.f_local, self.F_total
        xp = x.reshape(F, -1, x.shape[-1])[:fl]
        reps = -(-HW // xp.shape[1])
        return xp.repeat(1, reps, 1)[:, :HW].reshape(-1, x.shape[-1]).contiguous()

    def to_frames_start(self, x, HW):
        return self.to_frames(x, HW), (lambda: None)

    entries = 1

    def allreduce(self, sums):
        return sums

    def halo(self, buf):
        return buf

    def halo_raw(self, first, last, sums):
        """the one-collective form of round 5 (raw boundary frames + GroupNorm sums in one all-gather): this rank's own row k times"""
        from lkgd_amd.dist import SUMS_SLOT
        k, B, n = self.plan.frame_shards, len(first), first[0].numel()
        send = torch.empty(B, 2 * n + SUMS_SLOT, dtype=first[0].dtype, device=first[0].device)
        for b in range(B):
            send[b, :n].copy_(first[b].reshape(-1))
            send[b, n:2 * n].copy_(last[b].reshape(-1))
            send[b, 2 * n:].view(torch.float32).copy_((sums[b] / k).reshape(-1))     # k equal parts add up to the local sums
        return send.unsqueeze(0).repeat(k, 1, 1).contiguous()


dev = torch.device("cuda", 0)
unet = B.build_unet(dev, False)
h, w = 72, 128
for world in (8, 4, 2):
    plan = make_plan(world, 0, 14, cfg=True)
    frames = plan.f_local
    lat0, img, emb, ids = B.synthetic_inputs(dev, frames, h, w)
    tok = ops.prepare_unet_input(lat0.half(), img, 2, 700.0)
    tok, emb1, ids1 = tok[: tok.shape[0] // 2].contiguous(), emb[:1].contiguous(), ids[:1].contiguous()
    shard = LoopbackShard(plan) if plan.frame_shards > 1 else None
    line = f"rank of {world} ({frames} frames of one CFG half)"
    for gather in ((False, True) if shard is not None else (False,)):
        ld.TEMPORAL_GATHER = gather
        from lkgd_amd import replay
        enc = (emb if shard is not None else emb1).half().contiguous()
        t_dev = torch.ones(1, dtype=torch.float32, device=dev)
        unet.forward_tokens(tok, 1, frames, h, w, t_dev, enc, ids1.float().contiguous(), shard=shard)
        with replay.record() as rec:          # the launch list a rank replays (lkgd_amd/dist_run.py), stand-in exchanges included
            rec.result = unet.forward_tokens(tok, 1, frames, h, w, t_dev, enc, ids1.float().contiguous(), shard=shard)[0]
        rec.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            rec.run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        nl = len(rec.calls)
        if os.environ.get("PROFILE") == "1" and not gather:       # per-call device time by entry point (events around every call)
            from collections import defaultdict
            stream = torch.cuda.current_stream().cuda_stream
            evs = []
            for fn, a, name, flop in rec.calls:
                if fn is None:
                    a()
                    continue
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(); fn(*a, stream); e_.record()
                evs.append((name, s_, e_))
            torch.cuda.synchronize()
            fam = defaultdict(lambda: [0, 0.0])
            for name, s_, e_ in evs:
                fam[name][0] += 1; fam[name][1] += s_.elapsed_time(e_)
            print(f"  per entry point, world {world}:")
            for k_, (n_, t_) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
                print(f"    {k_:34s} {n_:4d} calls {t_:7.3f} ms")
        rec.release()
        line += f" | {'all-gather form' if gather else ('pixel re-sharding' if shard is not None else 'no frame sharding')}: {ms:6.1f} ms ({nl} calls)"
    print(line, flush=True)
