"""Restatement of the reference's joint-attention block hooks.  ORACLE - test infrastructure only.

Follows /root/reference/patch/patch.py: ``initialize_joint_layers`` :143-172 (attn1n = deepcopy(attn1), zero-init
``conv1n``), ``ToMeBlock.forward`` joint branch :438-501, ``forward_temporal`` joint branch :616-658, partner selection
by boolean-mask swap :466-468, optional frame flip :471-475, ``post`` variants :484-494.

Pinned: tests/golden/patch_joint.safetensors holds outputs of the reference's own ToMeBlock code (run through
name-only stubs by tests/golden/make_goldens.py) on the blocks of oracle/blocks.py.
"""
from __future__ import annotations

import copy
import functools
import math

import torch
import torch.nn as nn

from .blocks import BasicTransformerBlock, TemporalBasicTransformerBlock


def initialize_joint_layers(block: nn.Module, post: str = "conv") -> None:
    block.attn1n = copy.deepcopy(block.attn1)
    dim = block.attn1n.out_dim
    if post == "conv":
        block.conv1n = nn.Linear(dim, dim, bias=False)
        nn.init.zeros_(block.conv1n.weight)
    elif post == "scale":
        block.scale1n = nn.Parameter(torch.zeros(1, 1, dim))
    elif post == "conv_fuse":
        block.conv1n = nn.Linear(dim * 2, dim * 2, bias=False)
        nn.init.zeros_(block.conv1n.weight)
    else:
        raise AssertionError(post)
    block.post = post
    block.joint_scale = 1.0


def _partner(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """joint_encoder_hidden_states[~m] = x[m]; [m] = x[~m]  (patch.py:466-468)."""
    mask = mask.repeat_interleave(x.shape[0] // len(mask), dim=0)
    out = torch.empty_like(x)
    out[~mask] = x[mask]
    out[mask] = x[~mask]
    return out, mask


def _post(block, o1n, mask):
    if block.post == "conv":
        return block.conv1n(o1n)
    if block.post == "scale":
        return block.scale1n * o1n
    cat = torch.cat([o1n[mask], o1n[~mask]], dim=-1)
    fx, fy = block.conv1n(cat).chunk(2, dim=-1)
    o1n = o1n.clone()
    o1n[mask] = fx
    o1n[~mask] = fy
    return o1n


def basic_block_forward(block: BasicTransformerBlock, hidden_states, encoder_hidden_states, joint_mask,
                        enable_joint=True, flip=False, n_frames=None):
    """patch.py:390-580 for norm_type == "layer_norm"."""
    n = block.norm1(hidden_states)
    attn_output = block.attn1(n)
    if enable_joint:
        partner, mask = _partner(n, joint_mask)
        if flip:
            bt, s, c = partner.shape
            partner = partner.reshape(bt // n_frames, n_frames, s, c).flip(dims=[1]).reshape(bt, s, c)
        o1n = block.attn1n(n, encoder_hidden_states=partner)
        attn_output = attn_output + _post(block, o1n, mask) * block.joint_scale
    hidden_states = attn_output + hidden_states
    hidden_states = block.attn2(block.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
    return block.ff(block.norm3(hidden_states)) + hidden_states


def temporal_block_forward(block: TemporalBasicTransformerBlock, hidden_states, num_frames, encoder_hidden_states,
                           joint_mask, enable_joint=True):
    """patch.py:582-686."""
    bf, s, c = hidden_states.shape
    b = bf // num_frames
    h = hidden_states[None, :].reshape(b, num_frames, s, c).permute(0, 2, 1, 3).reshape(b * s, num_frames, c)
    residual = h
    h = block.ff_in(block.norm_in(h))
    if block.is_res:
        h = h + residual
    n = block.norm1(h)
    attn_output = block.attn1(n)
    if enable_joint:
        partner, mask = _partner(n, joint_mask)
        o1n = block.attn1n(n, encoder_hidden_states=partner)
        if block.post == "conv":
            o1n = block.conv1n(o1n)
        elif block.post == "scale":
            o1n = block.scale1n * o1n
        attn_output = attn_output + o1n
    h = attn_output + h
    h = block.attn2(block.norm2(h), encoder_hidden_states=encoder_hidden_states) + h
    ff = block.ff(block.norm3(h))
    h = ff + h if block.is_res else ff
    return h[None, :].reshape(b, s, num_frames, c).permute(0, 2, 1, 3).reshape(b * num_frames, s, c)


def apply_joint(model: nn.Module, joint_mask, post: str = "conv", flip: bool = False, spatial: bool = True,
                temporal: bool = True) -> None:
    """patch.apply_patch + initialize_joint_layers + set_joint_attention_mask on an oracle UNet (patch.py:719-817,:966-996):
    the spatial / temporal transformer blocks get ``attn1n`` + post layer and the joint forwards above.  The frame count
    for ``flip`` comes from the sample shape the reference records in a forward pre-hook (:691-697)."""
    mask = torch.as_tensor(joint_mask, dtype=torch.bool)
    info = {"size": None}
    model.register_forward_pre_hook(lambda mod, args: info.__setitem__("size", args[0].shape))
    for m in model.modules():
        if spatial and type(m) is BasicTransformerBlock:
            initialize_joint_layers(m, post)
            m.enable_joint_attention = True
            m.forward = functools.partial(_joint_spatial_forward, m, mask, flip, info)
        elif temporal and type(m) is TemporalBasicTransformerBlock:
            initialize_joint_layers(m, post)
            m.enable_joint_attention = True
            m.forward = functools.partial(_joint_temporal_forward, m, mask)


def _joint_spatial_forward(block, mask, flip, info, hidden_states, encoder_hidden_states=None, **_):
    return basic_block_forward(block, hidden_states, encoder_hidden_states, mask, block.enable_joint_attention, flip,
                               info["size"][1] if info["size"] is not None else None)


def _joint_temporal_forward(block, mask, hidden_states, num_frames=None, encoder_hidden_states=None, **_):
    return temporal_block_forward(block, hidden_states, num_frames, encoder_hidden_states, mask,
                                  block.enable_joint_attention)


# ------------------------------------------------------------------------------------------------ FSM hook (a15)
def initialize_fsm_layers(block: nn.Module) -> None:
    """patch_FSM.py:92-97: zero-initialised 3x3 ``conv_fuse`` on 2C channels"""
    c = block.attn1.out_dim
    block.conv_fuse = nn.Conv2d(2 * c, 2 * c, 3, 1, 1)
    nn.init.zeros_(block.conv_fuse.weight)
    nn.init.zeros_(block.conv_fuse.bias)


def fsm_block_forward(block: BasicTransformerBlock, hidden_states, encoder_hidden_states, track, track_res,
                      enable=True):
    """/root/reference/patch/patch_FSM.py ``ToMeBlock.forward`` :312-504 for norm_type == "layer_norm": self-attention,
    then the track-guided fuse :380-441 between even (src) and odd (dst) batch entries, then cross-attention and FF.
    ``track`` = (src_tracks [B/2,P,2] (x,y), dst_tracks, pred_visibility [B/2,P]); ``track_res`` = (..., H, W)."""
    n = block.norm1(hidden_states)
    hidden_states = block.attn1(n) + hidden_states
    if enable:
        track_h, track_w = track_res[-2:]
        downsample = int(math.ceil(math.sqrt((track_h * track_w) // hidden_states.shape[1])))
        feat_h, feat_w = track_h // downsample, track_w // downsample
        src_tracks, dst_tracks, vis = track
        src_tracks = (src_tracks / downsample).long()
        dst_tracks = (dst_tracks / downsample).long()
        dst_tracks[..., 0] = dst_tracks[..., 0].clamp(0, feat_w - 1)
        dst_tracks[..., 1] = dst_tracks[..., 1].clamp(0, feat_h - 1)
        src_idx = src_tracks[..., 0] + src_tracks[..., 1] * feat_w
        dst_idx = dst_tracks[..., 0] + dst_tracks[..., 1] * feat_w
        src_feats, dst_feats = hidden_states[::2], hidden_states[1::2]
        B, N, C = src_feats.shape
        visx = vis.unsqueeze(-1).expand(B, -1, C)
        invisible = visx == 0
        visf = visx.to(src_feats)
        gathered = torch.gather(dst_feats, 1, dst_idx.unsqueeze(-1).expand(B, -1, C)).clone()
        gathered[invisible] = 0
        canvas = torch.zeros_like(src_feats)
        scat = torch.scatter_add(canvas, 1, src_idx.unsqueeze(-1).expand(B, -1, C), gathered)
        cnt = torch.scatter_add(canvas, 1, src_idx.unsqueeze(-1).expand(B, -1, C), visf)
        reduced_src = scat / (cnt + 1e-6)
        cat = torch.cat([src_feats, reduced_src], dim=-1)
        cat = cat.reshape(B, feat_h, feat_w, 2 * C).permute(0, 3, 1, 2)
        fused = block.conv_fuse(cat).permute(0, 2, 3, 1).reshape(B, N, 2 * C)
        src_fused, sdst_fused = fused.chunk(2, dim=-1)
        regathered = torch.gather(sdst_fused, 1, src_idx.unsqueeze(-1).expand(B, -1, C)).clone()
        regathered[invisible] = 0
        canvas = torch.zeros_like(dst_feats)
        dfused = torch.scatter_add(canvas, 1, dst_idx.unsqueeze(-1).expand(B, -1, C), regathered)
        cntd = torch.scatter_add(canvas, 1, dst_idx.unsqueeze(-1).expand(B, -1, C), visf)
        reduced_dst = dfused / (cntd + 1e-6)
        fused2 = torch.stack([src_fused, reduced_dst], dim=1).reshape(2 * B, N, C)
        hidden_states = hidden_states + fused2
    hidden_states = block.attn2(block.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
    return block.ff(block.norm3(hidden_states)) + hidden_states


def apply_fsm(model: nn.Module, track, track_res) -> None:
    """patch_FSM.apply_patch + initialize_joint_layers + update_patch on an oracle UNet: every spatial
    BasicTransformerBlock gets ``conv_fuse`` and the FSM forward (the temporal blocks stay stock, with_temporal_block
    defaults to False, patch_FSM.py:645)."""
    for m in model.modules():
        if type(m) is BasicTransformerBlock:
            initialize_fsm_layers(m)
            m.forward = functools.partial(_fsm_forward, m, track, track_res)


def _fsm_forward(block, track, track_res, hidden_states, encoder_hidden_states=None, **_):
    return fsm_block_forward(block, hidden_states, encoder_hidden_states, track, track_res)
