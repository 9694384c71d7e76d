"""Frame-parallel sharding of one clip over the GPUs of a node (one process per GPU, RCCL over xGMI).

There is no counterpart in the reference (its SVD path is single-GPU); the scheme is the one
BASELINE.json's north_star names and SURVEY.md 8e details:

* ranks = cfg_groups x frame_shards.  With classifier-free guidance the two CFG halves (uncond / cond) are independent
  through the whole UNet, so the first factor of 2 is CFG-parallel and costs ONE tiny exchange per step (the CFG combine);
* inside a CFG half the F frames are split into contiguous slices (14 frames over 4 shards = 4,4,3,3).  All spatial work
  (2-D convs, spatial attention, feed-forwards, per-frame GroupNorm) is local.  Temporal ops couple frames at a pixel:
  - temporal GroupNorm: all-reduce of the [32,2] fp32 partial sums (not the activations),
  - temporal Conv3d (3,1,1): needs one frame of halo on each side only - an equal-count all-gather of every rank's two
    BOUNDARY frames (2 of its 3-7 frames; no padding needed) fills the halo slots of a [f_local + 2] frame buffer,
  - temporal attention (default since round 3, SURVEY.md 8e variant iii): the tokens are RE-SHARDED by pixels around the
    attention - an all-to-all turns [f_local, HW, C] into [F, HW/k, C] (all frames of a pixel slice), LayerNorm + Q|K|V +
    attention run there exactly once per token (the fused lkgd_tattn_front where it applies), a second all-to-all brings the
    attention output back to frame slices.  A rank sends and receives 2 x (k-1)/k of its slice per block instead of receiving
    (k-1) slices, and projects no foreign frames.  LKGD_TEMPORAL_GATHER=1 selects the earlier form: all-gather of the
    normalised hidden states of the frame slices (C channels, padded to equal size because RCCL's all-gather wants equal
    counts), K|V for all frames projected locally.
  Per forward a rank of 8 receives 1.2 GB this way (2.2 GB with whole-slice gathers before every temporal op), a rank
  of 4 0.6 GB (1.5 GB) - SURVEY.md 8e lists the message sizes.

This module is the HOST logic of that scheme: the shard plan and the padded gather.  It is pure torch.distributed
(backend "nccl" == RCCL on the GPU box, "gloo" in the CPU tests), no kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import math
import os

import torch
import torch.distributed as dist


@dataclass
class ShardPlan:
    world: int
    rank: int
    cfg_groups: int            # 1 or 2
    frame_shards: int
    cfg_index: int             # which CFG half this rank computes (0 = uncond, 1 = cond); 0 when cfg_groups == 1
    shard_index: int
    splits: Tuple[int, ...]    # frames per shard
    f0: int                    # first frame of this rank
    num_frames: int

    @property
    def f_local(self) -> int:
        return self.splits[self.shard_index]

    @property
    def f_max(self) -> int:
        return max(self.splits)

    def frame_group_ranks(self) -> List[int]:
        """ranks that hold the other frame slices of the same CFG half"""
        base = self.cfg_index * self.frame_shards
        return list(range(base, base + self.frame_shards))

    def cfg_partner_ranks(self) -> List[int]:
        """ranks that hold the same frame slice of each CFG half"""
        return [c * self.frame_shards + self.shard_index for c in range(self.cfg_groups)]


def split_frames(num_frames: int, shards: int, unit: int = 1, symmetric: bool = False) -> Tuple[int, ...]:
    """contiguous, as even as possible, larger slices first: 14 over 4 -> (4, 4, 3, 3).  ``unit``: slices are cut at multiples of
    ``unit`` frames (2 with the FSM hook, which fuses frames 2k and 2k+1 - patch/patch_FSM.py:405-441: a pair must live on one
    rank): 14 over 4 in pairs -> (4, 4, 4, 2).  ``symmetric``: shard i and shard k-1-i hold equally many frames, so that frame f
    and frame F-1-f sit at mirrored positions of mirrored shards (joint attention with flip=True, patch/patch.py:471-475):
    14 over 4 -> (4, 3, 3, 4)"""
    if symmetric:
        if unit != 1:
            raise ValueError("symmetric frame slices in groups of frames are not supported")
        if shards < 1 or shards > num_frames:
            raise ValueError(f"cannot split {num_frames} frames over {shards} shards")
        q, r = divmod(num_frames, shards)
        if r % 2 and shards % 2 == 0:
            raise ValueError(f"{num_frames} frames do not split symmetrically over {shards} shards")
        out = [q] * shards
        for i in range(r // 2):                      # the extra frames go to the two ends, pairwise
            out[i] += 1
            out[shards - 1 - i] += 1
        if r % 2:
            out[shards // 2] += 1                    # ... and the odd one to the middle shard (its own mirror)
        return tuple(out)
    if unit < 1 or num_frames % unit:
        raise ValueError(f"{num_frames} frames are not whole groups of {unit}")
    units = num_frames // unit
    if shards < 1 or shards > units:
        raise ValueError(f"cannot split {num_frames} frames over {shards} shards" + (f" in groups of {unit}" if unit > 1 else ""))
    q, r = divmod(units, shards)
    return tuple((q + 1 if i < r else q) * unit for i in range(shards))


def make_plan(world: int, rank: int, num_frames: int, cfg: bool, frame_unit: int = 1, symmetric: bool = False) -> ShardPlan:
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    cfg_groups = 2 if (cfg and world >= 2) else 1
    if world % cfg_groups:
        raise ValueError(f"world size {world} must be even when classifier-free guidance is on")
    shards = world // cfg_groups
    splits = split_frames(num_frames, shards, frame_unit, symmetric)
    ci, si = divmod(rank, shards)
    return ShardPlan(world, rank, cfg_groups, shards, ci, si, splits, sum(splits[:si]), num_frames)


# Conv3d halo exchange: the all-gather form is the default until a multi-GPU RCCL run has compared both forms bitwise
# (ADVICE r2); LKGD_HALO_P2P=1 selects the neighbour-only batch_isend_irecv exchange (2 frames in / out per rank instead of
# 2(k-1) delivered), tests/test_dist_gpu.py checks the two forms against each other
_HALO_ALLGATHER = __import__("os").environ.get("LKGD_HALO_P2P", "0") != "1"


def _backend(group) -> str:
    return dist.get_backend(group)


def all_gather_into(buf: torch.Tensor, send: torch.Tensor, group=None) -> None:
    """dist.all_gather_into_tensor; with the gloo backend (CPU tests / single-GPU functional tests) device tensors are
    staged through host memory, RCCL ("nccl") takes the device pointers directly"""
    if _backend(group) == "nccl" or not send.is_cuda:
        dist.all_gather_into_tensor(buf, send, group=group)
        return
    hb, hs = torch.empty(buf.shape, dtype=buf.dtype), send.cpu()
    dist.all_gather_into_tensor(hb, hs, group=group)
    buf.copy_(hb)


def _step(f) -> None:
    """run a host-side exchange step now; under lkgd_amd.replay recording also make it part of the plan"""
    from . import replay
    replay.step(f)


def _dest(out: Optional[torch.Tensor], shape, like: torch.Tensor) -> torch.Tensor:
    """the caller's destination (a contiguous slice of a larger result: several batch entries per rank) or a new tensor"""
    if out is None:
        return torch.empty(shape, dtype=like.dtype, device=like.device)
    if out.numel() != math.prod(shape) or not out.is_contiguous() or out.dtype != like.dtype or out.device != like.device:
        raise ValueError("out= must be a contiguous tensor of the result's size, dtype and device")
    return out.view(shape)


def gather_frames(local: torch.Tensor, plan: ShardPlan, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """local [f_local, ...] -> [num_frames, ...] over the frame group (padded equal-count all-gather + compaction).
    Buffers are allocated outside the replayable steps, so a recorded plan re-runs only copies and the collective.
    ``out``: write the result there (every exchange step of a recorded plan then refreshes the caller's tensor itself)."""
    if plan.frame_shards == 1:
        if out is None:
            return local
        dst = _dest(out, tuple(local.shape), local)
        _step(lambda: dst.copy_(local))
        return dst
    fmax = plan.f_max
    if local.shape[0] != plan.f_local:
        raise ValueError("local frame count does not match the plan")
    if not local.is_contiguous():
        raise ValueError("gather_frames needs a contiguous slice")
    send = local
    if plan.f_local < fmax:
        send = torch.zeros((fmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        head = send[:plan.f_local]
        _step(lambda: head.copy_(local))
    even = all(s == fmax for s in plan.splits)
    buf = _dest(out if even else None, (plan.frame_shards * fmax,) + tuple(local.shape[1:]), local)
    _step(lambda: all_gather_into(buf, send, group))
    if even:
        return buf
    out = _dest(out, (plan.num_frames,) + tuple(local.shape[1:]), local)
    pairs, f = [], 0
    for j, n in enumerate(plan.splits):
        pairs.append((out[f:f + n], buf[j * fmax:j * fmax + n]))
        f += n

    def compact():
        for dst, src in pairs:
            dst.copy_(src)
    _step(compact)
    return out


#: temporal attention of a frame-sharded rank: True = all-gather of the normalised hidden states (round 1-2 form),
#: False = all-to-all re-sharding by pixels around the attention (default)
TEMPORAL_GATHER = __import__("os").environ.get("LKGD_TEMPORAL_GATHER", "0") == "1"


def pixel_splits(HW: int, shards: int) -> Tuple[int, ...]:
    """contiguous pixel ranges per shard, as even as possible, in units of 16 pixels where HW allows (the fused temporal
    front works on panels of 16 pixels): 9216 over 4 -> 2304 each; 144 over 4 -> (48, 32, 32, 32)"""
    unit = 16 if HW % 16 == 0 and HW // 16 >= shards else 1
    q, r = divmod(HW // unit, shards)
    if q == 0:
        raise ValueError(f"cannot split {HW} pixels over {shards} shards")
    return tuple((q + 1 if i < r else q) * unit for i in range(shards))


def _hip_rows(t: torch.Tensor, C: int) -> bool:
    """the pack / unpack around a re-sharding exchange runs as one HIP launch (fp16 rows of whole 16-byte vectors on the GPU);
    CPU tensors (the gloo tests) and other dtypes keep the strided copies"""
    return t.is_cuda and t.dtype == torch.float16 and C % 8 == 0 and os.environ.get("LKGD_NO_SHARD_ROWS", "0") != "1"


def all_to_all_rows(out: torch.Tensor, inp: torch.Tensor, out_rows: List[int], in_rows: List[int], group=None) -> None:
    """dist.all_to_all_single over dim 0 with per-peer row counts; gloo with device tensors is staged through host memory"""
    if _backend(group) == "nccl" or not inp.is_cuda:
        dist.all_to_all_single(out, inp, out_rows, in_rows, group=group)
        return
    ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
    dist.all_to_all_single(ho, hi, out_rows, in_rows, group=group)
    out.copy_(ho)


def frames_to_pixels(local: torch.Tensor, plan: ShardPlan, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """local [f_local, HW, C] (this rank's frames, all pixels) -> [F, px_local, C] (all frames, this rank's pixel slice)
    over the frame group.  Peer r gets this rank's frames of ITS pixel range and sends its frames of ours; the received
    blocks arrive in rank = frame order, so the result is frame-major without a compaction."""
    k, si = plan.frame_shards, plan.shard_index
    if k == 1:
        if out is None:
            return local
        dst = _dest(out, tuple(local.shape), local)
        _step(lambda: dst.copy_(local))
        return dst
    fl, HW, C = local.shape
    if fl != plan.f_local or not local.is_contiguous():
        raise ValueError("frames_to_pixels needs this rank's contiguous [f_local, HW, C] slice")
    px = pixel_splits(HW, k)
    p0 = [sum(px[:r]) for r in range(k)]
    send = torch.empty(fl * HW, C, dtype=local.dtype, device=local.device)
    recv = _dest(out, (plan.num_frames * px[si], C), local)
    in_rows = [fl * px[r] for r in range(k)]
    out_rows = [plan.splits[r] * px[si] for r in range(k)]
    pieces, o = [], 0
    for r in range(k):
        pieces.append((send[o:o + in_rows[r]].view(fl, px[r], C), local[:, p0[r]:p0[r] + px[r], :]))
        o += in_rows[r]

    one_launch = _hip_rows(local, C)

    def step():
        if one_launch:            # the k strided copies as one kernel (lkgd_shard_rows)
            from . import ops
            ops.shard_rows(local, send, fl, HW, C, px, True)
        else:
            for dst, src in pieces:
                dst.copy_(src)
        all_to_all_rows(recv, send, out_rows, in_rows, group)
    _step(step)
    return recv.view(plan.num_frames, px[si], C)


def pixels_to_frames_start(x: torch.Tensor, plan: ShardPlan, HW: int, group=None, out: Optional[torch.Tensor] = None):
    """inverse of frames_to_pixels, in two halves: x [F, px_local, C] -> (result [f_local, HW, C], finish).  The all-to-all is issued
    here - with RCCL as an asynchronous operation on the communicator's stream, which waits for the launch stream at this point -
    and ``finish()`` (the wait + the unpack of the received rows) must run before the result is read.  Whatever the caller enqueues
    in between runs beside the transfer: the temporal block's joint branch beside the main branch's attention output, the main
    out-projection beside the joint branch's (lkgd_amd/unet.py).  Both halves are replay steps."""
    k, si = plan.frame_shards, plan.shard_index
    if k == 1:
        if out is None:
            return x, (lambda: None)
        dst = _dest(out, tuple(x.shape), x)
        _step(lambda: dst.copy_(x))
        return dst, (lambda: None)
    F, pl, C = x.shape
    px = pixel_splits(HW, k)
    if F != plan.num_frames or pl != px[si] or not x.is_contiguous():
        raise ValueError("pixels_to_frames needs the contiguous [F, px_local, C] slice frames_to_pixels produced")
    fl = plan.f_local
    p0 = [sum(px[:r]) for r in range(k)]
    recv = torch.empty(fl * HW, C, dtype=x.dtype, device=x.device)
    out = _dest(out, (fl, HW, C), x)
    in_rows = [plan.splits[r] * pl for r in range(k)]          # frames of shard r are contiguous rows of x
    out_rows = [fl * px[r] for r in range(k)]
    pieces, o = [], 0
    for r in range(k):
        pieces.append((out[:, p0[r]:p0[r] + px[r], :], recv[o:o + out_rows[r]].view(fl, px[r], C)))
        o += out_rows[r]
    flat = x.view(F * pl, C)
    one_launch = _hip_rows(x, C)
    side = _backend(group) == "nccl" and x.is_cuda
    pending = []

    def issue():
        if side:
            pending.append(dist.all_to_all_single(recv, flat, out_rows, in_rows, group=group, async_op=True))
        else:
            all_to_all_rows(recv, flat, out_rows, in_rows, group)

    def wait_and_unpack():
        while pending:
            pending.pop().wait()          # the launch stream waits for the communicator's stream; the host does not block
        if one_launch:
            from . import ops
            ops.shard_rows(recv, out, fl, HW, C, px, False)
        else:
            for dst, src in pieces:
                dst.copy_(src)
    _step(issue)
    return out, (lambda: _step(wait_and_unpack))


def pixels_to_frames(x: torch.Tensor, plan: ShardPlan, HW: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """inverse of frames_to_pixels: x [F, px_local, C] -> [f_local, HW, C] (issued and finished at once)"""
    res, finish = pixels_to_frames_start(x, plan, HW, group, out)
    finish()
    return res


def exchange_halo(buf: torch.Tensor, plan: ShardPlan, group=None) -> torch.Tensor:
    """buf [f_local + 2, ...] with the rank's own frames already in slots 1..f_local: fills slot 0 with the previous
    shard's last frame and slot f_local + 1 with the next shard's first one.  At the ends of the clip the slot is left as
    the caller allocated it (zeros = the Conv3d's zero padding).

    Default: an all-gather of every shard's two boundary frames over the frame group (2(k-1) frames delivered to every
    rank, 6 of them unused at k = 4).  LKGD_HALO_P2P=1: neighbour-only exchange - each rank sends its first frame to shard
    i-1 and its last frame to shard i+1 and receives the two it needs (``batch_isend_irecv``, point-to-point over the one
    xGMI link to each neighbour); opt-in until both forms have been compared on a multi-GPU RCCL world."""
    k, fl, si = plan.frame_shards, plan.f_local, plan.shard_index
    if k == 1:
        return buf
    if buf.shape[0] != fl + 2 or not buf.is_contiguous():
        raise ValueError("halo buffer must be a contiguous [f_local + 2, ...] tensor")
    frame = tuple(buf.shape[1:])
    if _HALO_ALLGATHER:
        send = torch.empty((2,) + frame, dtype=buf.dtype, device=buf.device)
        got = torch.empty((2 * k,) + frame, dtype=buf.dtype, device=buf.device)

        def step():
            send[0].copy_(buf[1])
            send[1].copy_(buf[fl])
            all_gather_into(got, send, group)
            if si > 0:
                buf[0].copy_(got[2 * (si - 1) + 1])
            if si < k - 1:
                buf[fl + 1].copy_(got[2 * (si + 1)])
        _step(step)
        return buf
    ranks = plan.frame_group_ranks()                      # global ranks of the shards of this CFG half
    prev_r = ranks[si - 1] if si > 0 else None
    next_r = ranks[si + 1] if si < k - 1 else None
    direct = _backend(group) == "nccl" or not buf.is_cuda     # gloo with device tensors: staged through host memory

    def step():
        first, last, lo, hi = buf[1], buf[fl], buf[0], buf[fl + 1]
        if not direct:
            first, last = first.cpu(), last.cpu()
            lo, hi = torch.empty_like(first), torch.empty_like(first)
        ops_ = []
        # the same order on both sides of a link: every rank posts [to prev, from prev, to next, from next]
        if prev_r is not None:
            ops_.append(dist.P2POp(dist.isend, first, prev_r, group))
            ops_.append(dist.P2POp(dist.irecv, lo, prev_r, group))
        if next_r is not None:
            ops_.append(dist.P2POp(dist.isend, last, next_r, group))
            ops_.append(dist.P2POp(dist.irecv, hi, next_r, group))
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
        if not direct:
            if prev_r is not None:
                buf[0].copy_(lo)
            if next_r is not None:
                buf[fl + 1].copy_(hi)
    _step(step)
    return buf


def exchange_with_mirror_start(x: torch.Tensor, plan: ShardPlan, group=None):
    """x [rows, W] of this shard -> (recv, finish): recv will hold the [rows, W] tensor of the MIRROR shard k-1-i of the frame group
    (which receives ours) once ``finish()`` has run.  The slices must be symmetric (split_frames(symmetric=True)): both sides then
    hold equally many rows, and local frame t of a shard mirrors local frame f_local-1-t of the other.  One all-to-all over the frame
    group whose only non-empty block is the mirror's (a shard that is its own mirror - the middle one of an odd count - receives its
    own rows).

    This is the one exchange of the sharded forward whose consumer is NOT the next op: the K | V rows of a spatial joint block are
    read by attn1n's attention only, and the block's main branch (QKV, attention, out-projection) stands between.  With RCCL the
    all-to-all is therefore ISSUED here as an asynchronous operation - the communicator's own stream waits for the launch stream
    at this point and the transfer runs beside the kernels enqueued afterwards - and ``finish()`` makes the launch stream wait for
    it right before the first reader.  Both halves are replay steps (lkgd_amd/replay.py).  With gloo (tests; device tensors staged
    through host memory) the exchange completes inside the start step and finish() is empty."""
    k, si = plan.frame_shards, plan.shard_index
    if k == 1:
        return x, (lambda: None)
    if plan.splits[si] != plan.splits[k - 1 - si] or plan.f0 + plan.f_local != plan.num_frames - sum(plan.splits[:k - 1 - si]):
        raise ValueError("exchange_with_mirror needs symmetric frame slices (make_plan(symmetric=True))")
    if not x.is_contiguous() or x.dim() != 2:
        raise ValueError("exchange_with_mirror needs a contiguous [rows, W] tensor")
    recv = torch.empty_like(x)
    rows = [x.shape[0] if r == k - 1 - si else 0 for r in range(k)]
    side = _backend(group) == "nccl" and x.is_cuda
    pending = []

    def issue():
        if side:
            pending.append(dist.all_to_all_single(recv, x, rows, rows, group=group, async_op=True))
        else:
            all_to_all_rows(recv, x, rows, rows, group)

    def wait():
        while pending:
            pending.pop().wait()          # the launch stream waits for the communicator's stream; the host does not block
    _step(issue)
    return recv, (lambda: _step(wait))


def exchange_with_mirror(x: torch.Tensor, plan: ShardPlan, group=None) -> torch.Tensor:
    """exchange_with_mirror_start + finish at once"""
    recv, finish = exchange_with_mirror_start(x, plan, group)
    finish()
    return recv


#: temporal GroupNorm + Conv3d halo of a frame-sharded rank in ONE collective (default): the raw boundary frames and the
#: GroupNorm partial sums travel in the same all-gather, every rank adds the sums in rank order and normalises the two halo
#: frames it received itself.  LKGD_GN_HALO_SPLIT=1 (or the neighbour-only halo form, LKGD_HALO_P2P=1) selects the two-collective
#: form: all-reduce of the sums, then the exchange of the NORMALISED boundary frames.
GN_HALO_FUSED = os.environ.get("LKGD_GN_HALO_SPLIT", "0") != "1"
SUMS_SLOT = 128      # fp16 elements reserved per batch entry for its [32, 2] fp32 sums


def gather_boundary_frames_and_sums(first: List[torch.Tensor], last: List[torch.Tensor], sums: torch.Tensor, plan: ShardPlan,
                                    group=None) -> torch.Tensor:
    """One all-gather over the frame group carrying, per batch entry, this rank's first and last RAW frame (``first[b]`` /
    ``last[b]``: contiguous [HW, C] fp16 views) and its GroupNorm partial sums (``sums`` [B, 32, 2] fp32).  Returns
    got [k, B, 2 * HW * C + SUMS_SLOT] fp16: rank r's entry b = first frame | last frame | sums (as raw fp32 bits).  What
    the two-collective form spends on a latency-bound 256-byte all-reduce per temporal GroupNorm (44 per UNet forward) is
    gone, and the sums are added in rank order by every rank alike (lkgd_groupnorm_finalize_parts): the statistics are
    bitwise the same on all ranks, whatever the collective library's reduction order."""
    k, B = plan.frame_shards, len(first)
    n = first[0].numel()
    if sums.shape != (B, 32, 2) or sums.dtype != torch.float32 or any(t.numel() != n for t in first + last):
        raise ValueError("gather_boundary_frames_and_sums: B frames of equal size and [B, 32, 2] fp32 sums")
    W = 2 * n + SUMS_SLOT
    send = torch.empty(B, W, dtype=first[0].dtype, device=first[0].device)
    got = torch.empty(k, B, W, dtype=send.dtype, device=send.device)
    sview = [send[b, 2 * n:].view(torch.float32) for b in range(B)]         # 64 floats each

    def step():
        for b in range(B):
            send[b, :n].copy_(first[b].reshape(-1))
            send[b, n:2 * n].copy_(last[b].reshape(-1))
            sview[b].copy_(sums[b].reshape(-1))
        all_gather_into(got.view(-1), send.view(-1), group)
    _step(step)
    return got


def allreduce_sums(sums: torch.Tensor, plan: ShardPlan, group=None) -> torch.Tensor:
    """sum of the GroupNorm partial sums over the frame group (fp32, a few hundred bytes)"""
    if plan.frame_shards > 1:
        if _backend(group) == "nccl" or not sums.is_cuda:
            _step(lambda: dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group))
        else:
            def via_host():
                h = sums.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                sums.copy_(h)
            _step(via_host)
    return sums
