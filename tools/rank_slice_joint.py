#!/usr/bin/env python3
"""Compute of ONE rank of the NAMED pipeline - the trans pipeline's [start, end] pair with the `patch` joint-attention hooks, UNet
batch [u_x, u_y, c_x, c_y] (utils/util.py:561-606) - on one GPU, exchanges replaced by local stand-ins of the right shapes (no bytes
moved; cf. tools/rank_slice_forms.py for the single-clip stock pipeline).  A rank of a CFG-parallel x frame-sharded run holds its frame
slice of BOTH clips of its CFG half: twice the rows per launch of a single-clip rank.  Prints ms per forward for the full pair on one
GPU and for ranks of 2 / 4 / 8, and the compute-only scaling they imply."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from lkgd_amd import ops, patch, replay
from lkgd_amd.dist import SUMS_SLOT, make_plan, pixel_splits


class LoopbackPair:
    """lkgd_amd.dist_run.ShardInfo without a process group, `entries` batch entries per rank"""

    def __init__(self, plan, entries):
        self.plan, self.F_total, self.f0, self.entries = plan, plan.num_frames, plan.f0, entries
        self.B_total, self.b0 = plan.cfg_groups * entries, plan.cfg_index * entries

    def _rep(self, x, n_out):            # [n, ...] -> [n_out, ...] by repetition (stand-in data, right shape)
        reps = -(-n_out // x.shape[0])
        return x.repeat(reps, *([1] * (x.dim() - 1)))[:n_out]

    def gather(self, local):
        fl, n, C = self.plan.f_local, self.entries, local.shape[-1]
        x = local.reshape(n, fl, -1, C)
        return torch.stack([self._rep(x[e], self.F_total) for e in range(n)]).reshape(-1, C).contiguous()

    def to_pixels(self, local, HW):
        fl, k, si, n, C = self.plan.f_local, self.plan.frame_shards, self.plan.shard_index, self.entries, local.shape[-1]
        px = pixel_splits(HW, k)[si]
        x = local.reshape(n, fl, HW, C)[:, :, :px]
        return torch.stack([self._rep(x[e], self.F_total) for e in range(n)]).reshape(-1, C).contiguous()

    def to_frames(self, x, HW):
        fl, F, n, C = self.plan.f_local, self.F_total, self.entries, x.shape[-1]
        xp = x.reshape(n, F, -1, C)[:, :fl]
        reps = -(-HW // xp.shape[2])
        return xp.repeat(1, 1, reps, 1)[:, :, :HW].reshape(-1, C).contiguous()

    def to_frames_start(self, x, HW):
        return self.to_frames(x, HW), (lambda: None)

    def halo_raw(self, first, last, sums):
        k, Bn, n = self.plan.frame_shards, len(first), first[0].numel()
        send = torch.empty(Bn, 2 * n + SUMS_SLOT, dtype=first[0].dtype, device=first[0].device)
        for b in range(Bn):
            send[b, :n].copy_(first[b].reshape(-1))
            send[b, n:2 * n].copy_(last[b].reshape(-1))
            send[b, 2 * n:].view(torch.float32).copy_((sums[b] / k).reshape(-1))
        return send.unsqueeze(0).repeat(k, 1, 1).contiguous()


dev = torch.device("cuda", 0)
unet = B.build_unet(dev, False)
patch.apply_patch(unet, with_temporal_block=True)
patch.initialize_joint_layers(unet)
with torch.no_grad():
    g = torch.Generator().manual_seed(12350)
    for name, prm in unet.named_parameters():
        if "attn1n" in name or "conv1n" in name:
            prm.copy_((torch.randn(prm.shape, generator=g) * (0.5 / max(prm.shape[-1], 1) ** 0.5)).to(prm))
unet.invalidate()
patch.set_joint_attention_mask(unet, [0, 1, 0, 1])
patch.set_joint_attention(unet, True)
h, w = 72, 128


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


_, _, emb, ids = B.synthetic_inputs(dev, 14, h, w)
emb4 = torch.stack([emb[0], emb[0], emb[1], 0.8 * emb[1]]).half().contiguous()          # [u_x, u_y, c_x, c_y]
res = {}
for world in (1, 2, 4, 8):
    plan = make_plan(world, 0, 14, cfg=True)
    fl = plan.f_local
    ent = 4 if world == 1 else 2                        # batch entries on the rank
    tok = (torch.randn(ent * fl * h * w, 8, generator=torch.Generator().manual_seed(1)) * 0.5).half().to(dev)
    t_dev = torch.ones(ent, dtype=torch.float32, device=dev)
    ids_l = ids[:1].repeat(ent, 1).float().contiguous()
    shard = None if world == 1 else LoopbackPair(plan, 2)
    with replay.record() as rec:
        rec.result = unet.forward_tokens(tok, ent, fl, h, w, t_dev, emb4, ids_l, shard=shard)[0]
    res[world] = timed(rec.run)
    print(f"joint pair, {'one GPU (4 entries x 14 frames)' if world == 1 else f'rank of {world} (2 entries x {fl} frames)'}: "
          f"{res[world]:7.2f} ms per forward ({len(rec.calls)} launches)", flush=True)
    rec.release()
print("compute-only scaling of the pair (no exchange time): " + ", ".join(f"{w_} GPUs {res[1] / res[w_]:.2f}x" for w_ in (2, 4, 8)))
