"""The reference's ``patch_FSM`` API on the lkgd_amd UNet (track-guided feature fuse between neighbouring batch entries).

Mirrors /root/reference/patch/patch_FSM.py: ``apply_patch`` :640-712, ``remove_patch`` :737-755, ``update_patch``
:759-771 / :903-915, ``collect_from_patch`` :774-786, ``set_joint_attention`` :856-867, ``initialize_joint_layers``
:869-880 (block side :92-97: a zero-initialised ``conv_fuse = Conv2d(2C, 2C, 3, 1, 1)``), ``set_joint_attention_mask``
:887-899.  The hook itself (``ToMeBlock.forward`` :380-441) runs as HIP kernels in
``lkgd_amd.unet.BasicTransformerBlock._fsm``: ``lkgd_fsm_rows`` (include/lkgd_hip.h section 9) for the gather /
scatter-mean along the tracks and the implicit-GEMM 3x3 convolution for ``conv_fuse``.

Usage is the reference's::

    patch_FSM.apply_patch(pipe)                       # spatial blocks
    patch_FSM.initialize_joint_layers(pipe)           # conv_fuse, then load the trained weights
    patch_FSM.update_patch(pipe, track=(src_tracks, dst_tracks, pred_visibility), track_res=(H, W))

``track`` = ``src_tracks`` / ``dst_tracks`` [B*F/2, P, 2] as (x, y) on the ``track_res`` pixel grid and
``pred_visibility`` [B*F/2, P].  Entry 2k of the flattened (batch, frame) axis is the source, 2k+1 the destination.

The per-resolution index work of :381-403 (downsample factor, truncation, clamp of the destination, linear indices) is
done once per ``update_patch`` and UNet level on the device and inverted into CSR lists (target cell -> tracked points)
so the kernels need neither atomics nor a zeroed canvas; the lists are cached in ``_tome_info``.

The temporal ToMeBlock of patch_FSM.py (:506-620) uses ``attn1n`` / ``conv1n``, which patch_FSM's own
``initialize_joint_layers`` never creates; ``with_temporal_block=True`` therefore raises here instead of failing later.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._lib import LkgdHipError
from .patch import _model, isinstance_str
from .patch import set_patch_lora_mask          # noqa: F401  patch_FSM.py:790-814 is the same function


def _fsm_blocks(model):
    for name, m in _model(model).named_modules():
        if getattr(m, "_lkgd_fsm", False):
            yield name, m


def apply_patch(model, seed: int = 123, flip=False, with_spatial_block=True, with_temporal_block=False,
                single_dir=False):
    if with_temporal_block:
        raise LkgdHipError("patch_FSM: the temporal hook needs attn1n/conv1n, which patch_FSM.initialize_joint_layers "
                           "does not create (reference patch_FSM.py:92-97 vs :565-571); use lkgd_amd.patch for it")
    remove_patch(model)
    dm = _model(model)
    dm._tome_info = {"size": None, "hooks": [], "fsm_tables": {},
                     "args": {"generator": None, "seed": seed, "flip": flip, "single_dir": single_dir}}
    for _, m in dm.named_modules():
        if with_spatial_block and isinstance_str(m, "BasicTransformerBlock"):
            m._lkgd_fsm = True
            m._tome_info = dm._tome_info
            m.enable_joint_attention = True      # class-level default of the reference's ToMeBlock (:66)
    return model


def remove_patch(model):
    dm = _model(model)
    for _, m in dm.named_modules():
        if getattr(m, "_lkgd_fsm", False):
            m._lkgd_fsm = False
            m.enable_joint_attention = False
            if "_tome_info" in m.__dict__:
                del m._tome_info
    if "_tome_info" in dm.__dict__ and "fsm_tables" in dm._tome_info:
        del dm._tome_info
    return model


def update_patch(model, **kwargs):
    """set attributes (``track``, ``track_res``, ...) on every patched module (:759-771)"""
    dm = _model(model)
    for _, m in dm.named_modules():
        if "_tome_info" in m.__dict__:
            for k, v in kwargs.items():
                setattr(m, k, v)
    if ("track" in kwargs or "track_res" in kwargs) and "_tome_info" in dm.__dict__:
        dm._tome_info.setdefault("fsm_tables", {}).clear()
    return model


def collect_from_patch(model, attr="tome"):
    return {n: getattr(m, attr) for n, m in _model(model).named_modules() if hasattr(m, attr)}


def set_joint_attention(model, enable=True):
    for _, m in _fsm_blocks(model):
        m.enable_joint_attention = enable
    return model


def initialize_joint_layers(model):
    dm = _model(model)
    for _, m in _fsm_blocks(model):
        dim = m.attn1.out_dim
        w = m.attn1.to_q.weight
        m.conv_fuse = nn.Conv2d(2 * dim, 2 * dim, 3, 1, 1, device=w.device, dtype=w.dtype)
        nn.init.zeros_(m.conv_fuse.weight)
        nn.init.zeros_(m.conv_fuse.bias)
    dm.invalidate()
    return model


def set_joint_attention_mask(model, joint_attn_mask):
    mask = torch.tensor(joint_attn_mask, dtype=torch.bool)
    for _, m in _fsm_blocks(model):
        m.joint_attn_mask = mask      # stored like the reference does (:887-899); the FSM forward does not read it
    return model


# ------------------------------------------------------------------------------------------------ track tables
def _csr(target: torch.Tensor, pairs: int, HW: int):
    """target [pairs, P] int64 cell of each point -> (csr_off int32 [pairs*HW+1], csr_pt int32 [pairs*P]); points of
    one cell keep their original order (stable sort) = the order of a sequential scatter_add"""
    P = target.shape[1]
    key = (target + torch.arange(pairs, device=target.device).unsqueeze(1) * HW).reshape(-1)
    skey, order = torch.sort(key, stable=True)
    off = torch.searchsorted(skey, torch.arange(pairs * HW + 1, device=target.device))
    assert order.numel() == pairs * P
    return off.to(torch.int32).contiguous(), order.to(torch.int32).contiguous()


def track_tables(block, ctx):
    """index tables of the current UNet level: (forward, backward), each (csr_off, csr_pt, gather_idx, vis) for
    ``ops.fsm_rows``.  forward: dst tokens -> src grid (:405-418); backward: fused tokens -> dst grid (:429-437)."""
    info = block.__dict__.get("_tome_info")
    track, track_res = getattr(block, "track", None), getattr(block, "track_res", None)
    if info is None or track is None or track_res is None:
        raise LkgdHipError("FSM hook enabled but no tracks set: patch_FSM.update_patch(model, track=(src, dst, vis), "
                           "track_res=(H, W))")
    cache = info.setdefault("fsm_tables", {})
    key = (ctx.H, ctx.W, ctx.N, ctx.b0, ctx.f0, ctx.B_total, ctx.F_total)
    if key in cache:
        return cache[key]
    HW, pairs = ctx.HW, ctx.N // 2
    track_h, track_w = (int(v) for v in track_res[-2:])
    downsample = int(math.ceil(math.sqrt((track_h * track_w) // HW)))        # :382-384
    feat_h, feat_w = track_h // downsample, track_w // downsample
    if (feat_h, feat_w) != (ctx.H, ctx.W):
        raise LkgdHipError(f"FSM hook: track_res {track_h}x{track_w} / {downsample} = {feat_h}x{feat_w} does not match "
                           f"the {ctx.H}x{ctx.W} feature grid")
    src_tracks, dst_tracks, vis = (torch.as_tensor(t).to(ctx.device) for t in track)
    all_pairs = ctx.B_total * ctx.F_total // 2            # the tracks describe the WHOLE call's (batch, frame) pairs
    if src_tracks.shape != dst_tracks.shape or src_tracks.shape[0] != all_pairs or src_tracks.shape[-1] != 2 \
            or tuple(vis.shape) != tuple(src_tracks.shape[:2]):
        raise LkgdHipError(f"FSM hook: tracks must be [{all_pairs}, P, 2] (x, y) with visibility [{all_pairs}, P]; got "
                           f"{tuple(src_tracks.shape)}, {tuple(dst_tracks.shape)}, {tuple(vis.shape)}")
    if all_pairs != pairs:
        # a sharded rank (lkgd_amd/dist_run.py): its local pair (entry b, q) is pair ((b0 + b) F + f0) / 2 + q of the call
        own = [((ctx.b0 + b) * ctx.F_total + ctx.f0) // 2 + q for b in range(ctx.B) for q in range(ctx.F // 2)]
        sel = torch.tensor(own, dtype=torch.long, device=ctx.device)
        src_tracks, dst_tracks, vis = src_tracks[sel], dst_tracks[sel], vis[sel]
    src = (src_tracks / downsample).long()                                    # :398-399 (truncation toward zero)
    dst = (dst_tracks / downsample).long()
    dst_x = dst[..., 0].clamp(0, feat_w - 1)                                  # :400-401 (only dst is clamped)
    dst_y = dst[..., 1].clamp(0, feat_h - 1)
    src_idx = src[..., 0] + src[..., 1] * feat_w                              # :402-403
    dst_idx = dst_x + dst_y * feat_w
    if bool(((src_idx < 0) | (src_idx >= HW)).any()):
        raise LkgdHipError("FSM hook: a source track lies outside the feature grid (the reference's scatter_add "
                           "raises an index error for it; only destination tracks are clamped)")
    visf = vis.to(torch.float32).reshape(-1).contiguous()
    fwd = _csr(src_idx, pairs, HW) + (dst_idx.reshape(-1).to(torch.int32).contiguous(), visf)
    bwd = _csr(dst_idx, pairs, HW) + (src_idx.reshape(-1).to(torch.int32).contiguous(), visf)
    cache[key] = (fwd, bwd)
    return cache[key]


def set_joint_layer_requires_grad(model, requires_grad):
    """patch_FSM.py:72-90,:816-827 (training knob, literal): the fuse convolution's parameters"""
    for _, m in _fsm_blocks(model):
        if hasattr(m, "conv_fuse"):
            m.conv_fuse.requires_grad_(requires_grad)
    return model


def initialize_joint_lora(model, adapter_name, joint_adapter_name):
    """patch_FSM.py:843-854 calls ``ToMeBlock.initialize_joint_lora``, which patch_FSM.py only has as a comment (:107-120):
    the reference raises AttributeError here; so does this"""
    raise AttributeError("'ToMeBlock' object has no attribute 'initialize_joint_lora' (commented out in patch_FSM.py:107-120)")
