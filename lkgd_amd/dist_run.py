"""Multi-GPU execution of ONE clip's denoising loop: CFG-parallel x frame slices, one process per GPU, RCCL.

Exchange points per UNet forward on a rank (frame_shards > 1 only):
  * each temporal GroupNorm  : all-reduce of [1,32,2] fp32 sums                            (22 blocks x 2)
  * each temporal Conv3d     : neighbour send/recv of the two boundary frames [HW,C] each   (22 blocks x 2)
  * each temporal attention  : all-to-all re-sharding [f_local,HW,C] -> [F,HW/k,C] around the attention and back (16 blocks;
                               LKGD_TEMPORAL_GATHER=1: all-gather of the normalised hidden states, K|V projected locally)
and once per step, over ALL ranks, the all-gather of the noise prediction [cfg*F*HW, 4] (1 MB) before the replicated
CFG-combine + Euler update.  With 2 GPUs (pure CFG-parallel) only the last exchange exists.

Correctness notes
  * every rank keeps the full latents (replicated, 1 MB) and the UNet weights; activations are sharded;
  * the temporal cross-attention context of diffusers 0.27 interleaves the CFG halves' embeddings over pixels
    (App. C11), so every rank evaluates the cross-attention bias table for BOTH embeddings;
  * uneven frame slices (14 = 4+4+3+3) are padded to equal counts for RCCL's all-gather and compacted afterwards;
  * several clips per call (round 5; the [start, end] pair of the trans pipelines with the `patch` joint-attention hooks,
    UNet batch [u_x, u_y, c_x, c_y]): a rank holds its frame slice of EVERY clip of its CFG half, so the entries a joint
    hook couples (masks [0,1,0,1]) are on one rank and the hook needs no exchange of its own; the temporal exchanges run
    entry by entry.  8 GPUs = CFG x (4,4,3,3) frames x 2 clips per rank.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import ops, replay
from ._lib import LkgdHipError
from .dist import (ShardPlan, all_gather_into, allreduce_sums, exchange_halo, exchange_with_mirror_start, frames_to_pixels,
                   gather_boundary_frames_and_sums, gather_frames, make_plan, pixels_to_frames, pixels_to_frames_start)


class ShardInfo:
    """what lkgd_amd.unet.Ctx needs to know about this rank's slice.  A rank holds ``entries`` batch entries (one clip: 1;
    the [start, end] pair of the trans pipelines: 2 per CFG half, rows (entry, frame, pixel)); the exchanges below run entry by
    entry - every entry's frames are a contiguous [f_local, HW, C] block of the token matrix."""

    def __init__(self, plan: ShardPlan, frame_group, entries: int = 1):
        self.plan = plan
        self.group = frame_group
        self.F_total = plan.num_frames
        self.f0 = plan.f0
        self.set_entries(entries)

    def set_entries(self, entries: int) -> None:
        """batch entries per CFG half (clips of the call)"""
        self.entries = entries
        self.B_total = self.plan.cfg_groups * entries      # batch entries of the whole job
        self.b0 = self.plan.cfg_index * entries            # first global batch entry of this rank

    def _per_entry(self, x: torch.Tensor, f, rows_out: int):
        """apply f(entry rows, out=) to each entry's contiguous row block; ``rows_out`` result rows per entry.  The result tensor is
        allocated ONCE and every entry's exchange writes its own slice of it: under lkgd_amd.replay the exchanges are the steps
        a plan re-runs, so nothing that is not such a step (a torch.cat of the parts) may stand between them and
        the kernels that read the result through recorded pointers."""
        n = self.entries
        if self.plan.frame_shards == 1:
            return x                          # one frame slice: nothing to exchange
        if n == 1:
            return f(x, None)
        rows = x.shape[0] // n
        out = torch.empty(n * rows_out, x.shape[-1], dtype=x.dtype, device=x.device)
        for e in range(n):
            f(x[e * rows:(e + 1) * rows], out[e * rows_out:(e + 1) * rows_out])
        return out

    def gather(self, local: torch.Tensor) -> torch.Tensor:
        """[entries*f_local*HW, C] tokens -> [entries*F*HW, C] over the frame group"""
        fl = self.plan.f_local
        hw = local.shape[0] // (self.entries * fl)

        def one(t, out):
            return gather_frames(t.reshape(fl, -1, t.shape[-1]), self.plan, self.group, out=out).reshape(-1, t.shape[-1])
        return self._per_entry(local, one, self.plan.num_frames * hw)

    def to_pixels(self, local: torch.Tensor, HW: int) -> torch.Tensor:
        """[entries*f_local*HW, C] tokens (own frames, all pixels) -> [entries*F*px_local, C] (all frames, own pixel slice): all-to-all"""
        from .dist import pixel_splits
        k = self.plan.frame_shards
        pxl = pixel_splits(HW, k)[self.plan.shard_index] if k > 1 else HW

        def one(t, out):
            return frames_to_pixels(t.reshape(self.plan.f_local, HW, t.shape[-1]), self.plan, self.group,
                                    out=out).reshape(-1, t.shape[-1])
        return self._per_entry(local, one, self.plan.num_frames * pxl)

    def to_frames(self, x: torch.Tensor, HW: int) -> torch.Tensor:
        """inverse of to_pixels"""
        F = self.plan.num_frames

        def one(t, out):
            return pixels_to_frames(t.reshape(F, t.shape[0] // F, t.shape[-1]), self.plan, HW, self.group,
                                    out=out).reshape(-1, t.shape[-1])
        return self._per_entry(x, one, self.plan.f_local * HW)

    def to_frames_start(self, x: torch.Tensor, HW: int):
        """to_frames in two halves: (result, finish) - every entry's all-to-all is issued now, finish() (waits + unpacks) before the
        result is read (lkgd_amd/dist.py::pixels_to_frames_start)"""
        if self.plan.frame_shards == 1:
            return x, (lambda: None)
        F, n = self.plan.num_frames, self.entries
        rows, rows_out = x.shape[0] // n, self.plan.f_local * HW
        out = torch.empty(n * rows_out, x.shape[-1], dtype=x.dtype, device=x.device)
        fins = []
        for e in range(n):
            t = x[e * rows:(e + 1) * rows]
            _, fin = pixels_to_frames_start(t.reshape(F, t.shape[0] // F, t.shape[-1]), self.plan, HW, self.group,
                                            out=out[e * rows_out:(e + 1) * rows_out])
            fins.append(fin)

        def finish():
            for f in fins:
                f()
        return out, finish

    def allreduce(self, sums: torch.Tensor) -> torch.Tensor:
        return allreduce_sums(sums, self.plan, self.group)

    def mirror_start(self, x: torch.Tensor):
        """[rows, W] of this shard <-> the same-shaped tensor of the mirror shard (joint attention with flip=True: frame f of an
        entry attends to frame F-1-f of its partner, which the mirror shard holds at local position f_local-1-t).  Returns
        (recv, finish): issued now, finish() before the first read (lkgd_amd/dist.py::exchange_with_mirror_start)"""
        return exchange_with_mirror_start(x, self.plan, self.group)

    def halo_raw(self, first, last, sums: torch.Tensor) -> torch.Tensor:
        """the raw boundary frames and the GroupNorm partial sums of every entry in one all-gather (lkgd_amd/dist.py)"""
        return gather_boundary_frames_and_sums(first, last, sums, self.plan, self.group)

    def halo(self, buf: torch.Tensor) -> torch.Tensor:
        """[entries*(f_local+2)*HW, C] tokens with the own frames in the middle of every entry's block -> neighbours' boundary
        frames in the two end slots of every block"""
        fl, n = self.plan.f_local, self.entries
        rows = buf.shape[0] // n
        for e in range(n):
            exchange_halo(buf[e * rows:(e + 1) * rows].reshape(fl + 2, -1, buf.shape[-1]), self.plan, self.group)
        return buf


class DistDenoiser:
    def __init__(self, pipe, world: int, rank: int, num_frames: int, cfg: bool = True):
        if not dist.is_initialized():
            raise LkgdHipError("torch.distributed is not initialised")
        self.pipe = pipe
        # the FSM hook (lkgd_amd/patch_FSM.py) fuses frames 2k and 2k+1 of a clip: frame slices are then cut at even frames
        fsm = any(getattr(m, "_lkgd_fsm", False) for m in pipe.unet.modules())
        # joint attention with flip=True (patch.apply_patch(flip=True): checkpoints trained with it, utils/util.py:541-561) pairs frame f
        # with frame F-1-f of the partner clip: the frame slices are then symmetric - 14 over 4 = (4,3,3,4) - and every spatial joint
        # block trades its K | V rows with the mirror shard (ShardInfo.mirror)
        info = getattr(pipe.unet, "_tome_info", None)
        flip = bool(info and info.get("args", {}).get("flip", False)) and not fsm
        sharded = world > (2 if cfg else 1)
        self.plan = make_plan(world, rank, num_frames, cfg, frame_unit=2 if (fsm and sharded) else 1, symmetric=flip and sharded)
        # every rank creates every group, in the same order
        self.frame_group = None
        for c in range(self.plan.cfg_groups):
            ranks = list(range(c * self.plan.frame_shards, (c + 1) * self.plan.frame_shards))
            g = dist.new_group(ranks) if self.plan.frame_shards > 1 else None
            if c == self.plan.cfg_index:
                self.frame_group = g
        self.shard = ShardInfo(self.plan, self.frame_group)
        #: replay the step's UNet forward from a recorded launch list (lkgd_amd/replay.py); a frame-sharded rank has
        #: only ~1/8 of the device work per forward and would otherwise wait on the Python module walk
        self.use_replay = os.environ.get("LKGD_NO_REPLAY", "0") != "1"
        self._arenas = replay.ArenaSet()          # the recorded forward's private allocator pool (lkgd_amd/replay.py)
        #: bench.py's per-launch GEMM events (ops.GEMM_EVENTS) are taken on every `event_stride`-th Euler step only: an
        #: event pair per launch is ~600 extra queue packets per forward, which a rank with ~24 ms of device work per
        #: forward would feel (on one GPU every step is sampled)
        self.event_stride = 5

    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, image_latents: torch.Tensor, image_embeddings: torch.Tensor,
                added_time_ids: torch.Tensor, num_inference_steps: int = 25, min_guidance_scale: float = 1.0,
                max_guidance_scale: float = 3.0, domain_features: Optional[torch.Tensor] = None,
                flow_features: Optional[torch.Tensor] = None, controlnet_condition: Optional[torch.Tensor] = None,
                controlnet_cond_scale: float = 1.0) -> torch.Tensor:
        """``pipeline.denoise`` over the ranks.  ``domain_features`` / ``flow_features`` (the LKGD UNet): the fuse is
        replicated (2 MFLOP) and every rank holds the fused embedding of BOTH CFG halves.  ``controlnet_condition``
        [cfg * batch, F, 3, 8h, 8w] (pipeline_stable_video_diffusion_controlnet.py:582-607): every rank embeds the condition frames of
        ITS slice once per clip and runs the ControlNet-SVD encoder on its tokens before the UNet; the residuals are local
        token matrices, so nothing new crosses the links beyond the encoder's own temporal exchanges."""
        pipe, plan = self.pipe, self.plan
        unet, sch = pipe.unet, pipe.scheduler
        dev = unet.device
        B, F, _, H, W = latents.shape
        cfg = 2 if max_guidance_scale > 1 else 1
        if cfg != plan.cfg_groups and not (cfg == 2 and plan.cfg_groups == 1):
            raise LkgdHipError("shard plan was built for classifier-free guidance; got guidance <= 1")
        if F != plan.num_frames:
            raise LkgdHipError(f"the shard plan was built for {plan.num_frames} frames, the latents carry {F}")
        HW = H * W
        latents = latents.to(dev).contiguous()
        image_latents = image_latents.to(device=dev, dtype=torch.float16).contiguous()
        sch.set_timesteps(num_inference_steps, device=None)
        guidance = torch.linspace(min_guidance_scale, max_guidance_scale, F, dtype=torch.float32).to(dev)
        enc = image_embeddings.to(dev)
        if domain_features is not None:
            enc = unet.fused_embedding(enc, domain_features.to(dev), flow_features.to(dev))
        ids = added_time_ids.to(dev)
        vpred = sch.config.prediction_type == "v_prediction"
        from . import patch as _patch
        _patch.set_joint_attention(unet, enable=True)           # reference :555 (no-op unless the model is patched)
        fl, f0, fmax = plan.f_local, plan.f0, plan.f_max
        # batch entries of the UNet batch are ordered (CFG half, clip): [u_x, u_y, c_x, c_y] for the [start, end] pair of the
        # trans pipelines (pipeline_stable_video_diffusion_trans.py:549).  A CFG-parallel rank holds the B clips of ITS half -
        # the pairs the `patch` joint hooks couple (masks [0,1,0,1], utils/util.py:600-606) live on one rank, so the joint
        # attention needs no exchange of its own - and, under frame sharding, the same frame slice of every one of them.
        if plan.cfg_groups == 2:
            b_local, e0 = B, plan.cfg_index * B
        else:
            b_local, e0 = cfg * B, 0
        ids_local = ids[e0:e0 + b_local]
        self.shard.set_entries(b_local)
        send = torch.zeros(b_local, fmax * HW, 4, dtype=torch.float16, device=dev)
        buf = torch.empty(plan.world, b_local, fmax * HW, 4, dtype=torch.float16, device=dev)
        noise_full = torch.empty(cfg * B * F * HW, 4, dtype=torch.float16, device=dev)
        # step-invariant addresses for everything the forward reads, so its launch list can be replayed
        tok = torch.empty(cfg * B * F * HW, 8, dtype=torch.float16, device=dev)
        t_dev = torch.zeros(b_local, dtype=torch.float32, device=dev)
        ids_local = ids_local.to(torch.float32).contiguous()
        enc = enc.to(torch.float16).contiguous()
        if plan.frame_shards == 1:
            tok_local, pick = tok[e0 * F * HW:(e0 + b_local) * F * HW], None
        elif b_local == 1:
            r0 = (e0 * F + f0) * HW
            tok_local, pick = tok[r0:r0 + fl * HW], None
        else:
            tok_local = torch.empty(b_local * fl * HW, 8, dtype=torch.float16, device=dev)
            pick = (tok_local.reshape(b_local, fl, HW, 8), tok.reshape(cfg * B, F, HW, 8)[e0:e0 + b_local, f0:f0 + fl])
        ctrl_local = None
        if controlnet_condition is not None:
            if getattr(pipe, "controlnet", None) is None:
                raise LkgdHipError("controlnet_condition given but the pipeline has no controlnet")
            if controlnet_condition.shape[0] != cfg * B or controlnet_condition.shape[1] != F:
                raise ValueError("controlnet_condition must be [cfg * batch, frames, 3, 8h, 8w] (uncond entries first)")
            cc = controlnet_condition[e0:e0 + b_local]          # the rank's batch entries (its CFG half's clips)
            ctrl_local = cc[:, f0:f0 + fl].to(device=dev, dtype=torch.float16).contiguous()
            pipe.controlnet.prepare()
            pipe.controlnet._cond_tokens(ctrl_local, b_local, fl, H, W)     # once per clip, outside the recorded forward

        def forward():
            down = mid = None
            if ctrl_local is not None:
                down, mid, _ = pipe.controlnet.forward_tokens(tok_local, b_local, fl, H, W, t_dev, enc, ids_local, ctrl_local,
                                                              controlnet_cond_scale, shard=self.shard)
            return unet.forward_tokens(tok_local, b_local, fl, H, W, t_dev, enc, ids_local, down, mid, shard=self.shard)[0]
        recorded = None
        events_all = ops.GEMM_EVENTS
        try:
            for i, t in enumerate(sch.timesteps_host):
                sigma, sigma_next = sch.sigmas_host[i], sch.sigmas_host[i + 1]
                ops.GEMM_EVENTS = events_all if (events_all is not None and i % self.event_stride == self.event_stride // 2) else None
                ops.prepare_unet_input(latents, image_latents, cfg, sigma, out=tok)     # [cfg*F*HW, 8], replicated
                if pick is not None:
                    pick[0].copy_(pick[1])
                t_dev.fill_(float(t))
                if recorded is not None:
                    noise_local = recorded.run(ops.GEMM_EVENTS)
                elif self.use_replay:
                    with replay.record(self._arenas.take(dev, (b_local, fl, H, W, ctrl_local is not None, id(unet._pk)))) as recorded:   # the first step runs for real and is recorded
                        recorded.result = forward()
                    noise_local = recorded.result
                else:
                    noise_local = forward()
                # ---- exchange the noise prediction over all ranks (padded equal counts), compact, replicate the update
                send[:, :fl * HW].copy_(noise_local.reshape(b_local, fl * HW, 4))
                all_gather_into(buf.reshape(-1, 4), send.reshape(-1, 4))
                nf = noise_full.reshape(cfg * B, F * HW, 4)
                for r in range(plan.world):
                    ci, si = divmod(r, plan.frame_shards)
                    n, fs = plan.splits[si], sum(plan.splits[:si])
                    er = ci * b_local if plan.cfg_groups == 2 else 0
                    nf[er:er + b_local, fs * HW:(fs + n) * HW].copy_(buf[r, :, :n * HW])
                ops.cfg_euler_step(noise_full, latents, guidance, cfg, sigma, sigma_next, v_prediction=vpred)
        finally:          # whatever ended the loop: the bench's event list is restored, the plan's activations are freed
            ops.GEMM_EVENTS = events_all
            if recorded is not None:
                recorded.release()
        sch._step_index = num_inference_steps
        return latents


class DistDiTDenoiser:
    """BASELINE.json configs[4] over N GPUs: the CogVideoX DiT loop (lkgd_amd/cogvideox.py) with the SAME decomposition as the
    SVD loop - CFG-parallel x slices of the clip's latent frames (13 latent frames: 2 GPUs = CFG halves, 4 = CFG x (7, 6),
    8 = CFG x (4, 3, 3, 3)).  A rank holds the text tokens (226 rows, replicated - their stream is computed redundantly and stays
    bit-identical across the frame group) and the video tokens of its frames; every per-token op (adaLN LayerNorms, projections,
    qk norm, feed-forward, gated residuals) is local.  The one exchange per layer is the all-gather of the video keys and values
    ([frames, 1350, 1920] fp16 each, padded to equal counts): the rank's queries then run the flash kernel against all 17 776
    keys (``lkgd_attn_spatial_qk``, Sq < S).  Per forward a rank of 8 receives 30 layers x 2 x 10/13 x 67 MB = 3.1 GB over its
    three links (~ 6.8 ms at 153 GB/s per link) next to ~ 33 ms of compute; once per step the noise prediction (1.1 MB) is
    gathered over all ranks and the CFG combine + DDIM update are replicated.  No launch replay here: the modulation vectors
    are recomputed by tensor ops every step."""

    def __init__(self, transformer, scheduler, world: int, rank: int, latent_frames: int, cfg: bool = True):
        if not dist.is_initialized():
            raise LkgdHipError("torch.distributed is not initialised")
        self.transformer, self.scheduler = transformer, scheduler
        self.plan = make_plan(world, rank, latent_frames, cfg)
        self.frame_group = None
        for c in range(self.plan.cfg_groups):
            ranks = list(range(c * self.plan.frame_shards, (c + 1) * self.plan.frame_shards))
            g = dist.new_group(ranks) if self.plan.frame_shards > 1 else None
            if c == self.plan.cfg_index:
                self.frame_group = g
        self.shard = ShardInfo(self.plan, self.frame_group)

    @torch.no_grad()
    def denoise(self, latents, image_latents, prompt_embeds, domain_features, flow_features, num_inference_steps: int = 50,
                guidance_scale: float = 6.0, use_dynamic_cfg: bool = True) -> torch.Tensor:
        """``lkgd_amd.cogvideox.denoise`` over the ranks; latents / image_latents [1, F, C, h, w], prompt_embeds [2, L, 4096]"""
        from .cogvideox import dynamic_guidance
        tr, sch, plan = self.transformer, self.scheduler, self.plan
        dev = tr.device
        B, F, C_, H, W = latents.shape
        if B != 1 or F != plan.num_frames:
            raise LkgdHipError("sharded DiT denoising handles one clip whose latent frame count matches the plan")
        cfg = 2 if guidance_scale > 1.0 else 1
        if plan.cfg_groups == 2 and cfg != 2:
            raise LkgdHipError("shard plan was built for classifier-free guidance; got guidance_scale <= 1")
        if plan.cfg_groups == 1 and cfg == 2 and plan.frame_shards > 1:
            raise LkgdHipError("frame sharding without CFG-parallel needs guidance off (one batch entry per rank)")
        sch.set_timesteps(num_inference_steps)
        text = tr.fused_text(prompt_embeds.to(dev), domain_features.to(dev), flow_features.to(dev))      # replicated, once per clip
        latents = latents.to(device=dev, dtype=torch.float16)
        img = image_latents.to(device=dev, dtype=torch.float16)
        fl, f0, fmax = plan.f_local, plan.f0, plan.f_max
        co = tr.config.out_channels
        per = co * H * W
        send = torch.zeros(fmax * per, dtype=torch.float16, device=dev)
        buf = torch.empty(plan.world * fmax * per, dtype=torch.float16, device=dev)
        noise_full = torch.empty(cfg, F, co, H, W, dtype=torch.float16, device=dev)
        sharded = plan.frame_shards > 1
        for t in sch.timesteps.tolist():
            if plan.cfg_groups == 2:
                x = torch.cat([latents[:, f0:f0 + fl], img[:, f0:f0 + fl]], dim=2)
                out = tr.forward_tokens(x, text[plan.cfg_index:plan.cfg_index + 1], float(t), shard=self.shard if sharded else None)
                send[:fl * per].copy_(out.reshape(-1))
                all_gather_into(buf, send)
                for r in range(plan.world):
                    ci, si = divmod(r, plan.frame_shards)
                    n, fs = plan.splits[si], sum(plan.splits[:si])
                    noise_full[ci, fs:fs + n].copy_(buf[r * fmax * per:r * fmax * per + n * per].reshape(n, co, H, W))
            else:                      # one rank: the whole CFG batch, no exchange
                x = torch.cat([latents] * cfg)
                x = torch.cat([x, torch.cat([img] * cfg)], dim=2)
                noise_full.copy_(tr.forward_tokens(x, text, float(t)))
            noise = noise_full.float()
            g = dynamic_guidance(guidance_scale, num_inference_steps, t) if use_dynamic_cfg else guidance_scale
            n = noise[0:1] + g * (noise[1:2] - noise[0:1]) if cfg == 2 else noise
            latents = sch.step(n, t, latents.float())[0].to(torch.float16)
        return latents
