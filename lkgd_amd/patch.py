"""The reference's ``patch`` API on the lkgd_amd UNet (joint attention ``attn1n`` between paired batch entries).

Mirrors /root/reference/patch/patch.py: ``apply_patch`` :719-817, ``remove_patch`` :820-838, ``update_patch`` :841-853,
``collect_from_patch`` :856-870, ``set_joint_attention`` :938-950, ``set_joint_scale`` :952-963,
``initialize_joint_layers`` :966-977 (block side :143-172), ``set_joint_attention_mask`` :985-996.

The reference swaps ``module.__class__`` to a ToMeBlock whose Python ``forward`` does the extra attention in eager
PyTorch.  Here the blocks already own the joint branch as HIP kernel launches (lkgd_amd/unet.py
``BasicTransformerBlock._joint`` / ``TemporalBasicTransformerBlock._joint``: the spatial flash-attention /
temporal-attention kernels with a K/V batch permutation + the zero-init ``conv1n`` projection as a GEMM epilogue), so
"patching" marks which blocks take part and carries the flags.  Blocks are found the way the reference finds them: by
class NAME in the MRO (patch/utils.py:4-16).  The boolean-mask partner selection (:466-468) becomes a static int32
permutation computed on the host from the 4-entry mask (masks are constants, utils/util.py:600-606).

LoRA-mask plumbing (``set_patch_lora_mask`` / ``hack_lora_forward`` / ``initialize_joint_lora`` :57-92,:872-936) works on the
``lkgd_amd.lora.Linear`` wrappers: the reference's INFERENCE loader calls it (utils/util.py:598-604).  A per-entry masked
adapter is a per-entry weight here (lkgd_amd/lora.py), not an extra low-rank GEMM.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from ._lib import LkgdHipError


def isinstance_str(x: object, cls_name: str) -> bool:
    return any(c.__name__ == cls_name for c in x.__class__.__mro__)


def _model(model):
    return model.unet if hasattr(model, "unet") else model


def _patched_blocks(model):
    for name, m in _model(model).named_modules():
        if getattr(m, "_lkgd_patched", False):
            yield name, m


def apply_patch(model, seed: int = 123, flip=False, with_spatial_block=True, with_temporal_block=False,
                single_dir=False, name_skip=None):
    remove_patch(model)
    dm = _model(model)
    dm._tome_info = {"size": None, "hooks": [],
                     "args": {"generator": None, "seed": seed, "flip": flip, "single_dir": single_dir}}
    for name, m in dm.named_modules():
        if name_skip is not None and name_skip in name:
            continue
        is_s = isinstance_str(m, "BasicTransformerBlock")
        is_t = isinstance_str(m, "TemporalBasicTransformerBlock")
        if (with_spatial_block and is_s) or (with_temporal_block and is_t):
            m._lkgd_patched = True
            m._tome_info = dm._tome_info
            m.enable_joint_attention = True      # class-level default of the reference's ToMeBlock (:104)
            m.joint_scale = 1.0
    return model


def remove_patch(model):
    dm = _model(model)
    for _, m in dm.named_modules():
        if getattr(m, "_lkgd_patched", False):
            m._lkgd_patched = False
            m.enable_joint_attention = False
    if hasattr(dm, "_tome_info"):
        del dm._tome_info
    return model


def update_patch(model, **kwargs):
    for _, m in _model(model).named_modules():
        if hasattr(m, "_tome_info"):
            for k, v in kwargs.items():
                setattr(m, k, v)
    return model


def collect_from_patch(model, attr="tome"):
    return {n: getattr(m, attr) for n, m in _model(model).named_modules() if hasattr(m, attr)}


def initialize_joint_layers(model, post="conv", add_norm=False):
    if add_norm:
        # reference patch.py:447-448 calls norm1n(norm_hidden_states, timestep); TransformerSpatioTemporalModel passes its
        # blocks no timestep (None), so on the SVD UNet the reference's own add_norm forward cannot run - the option belongs
        # to the image (UNet2DCondition) joint-diffusion models of utils/util.py:700-712, outside this path
        raise LkgdHipError("add_norm=True (AdaLayerNormContinuous norm1n conditioned on `timestep`) does not apply to the "
                           "SVD UNet: its transformer blocks receive no timestep")
    dm = _model(model)
    for _, m in _patched_blocks(model):
        m.attn1n = copy.deepcopy(m.attn1)
        dim = m.attn1n.out_dim
        dev, dt = m.attn1.to_q.weight.device, m.attn1.to_q.weight.dtype
        for stale in ("conv1n", "scale1n"):
            if hasattr(m, stale):
                delattr(m, stale)
        if post == "conv":
            m.conv1n = nn.Linear(dim, dim, bias=False, device=dev, dtype=dt)
            nn.init.zeros_(m.conv1n.weight)
        elif post == "scale":
            m.scale1n = nn.Parameter(torch.zeros(1, 1, dim, device=dev, dtype=dt))
        elif post == "conv_fuse":
            m.conv1n = nn.Linear(2 * dim, 2 * dim, bias=False, device=dev, dtype=dt)
            nn.init.zeros_(m.conv1n.weight)
        else:
            raise AssertionError(f"Unkown post processing type {post}")
        m.add_norm = False
        m.post = post
        m.joint_scale = 1.0
    dm.invalidate()
    return model


def set_joint_attention(model, enable=True, name_filter=None):
    for name, m in _patched_blocks(model):
        if name_filter is None or name_filter in name:
            m.enable_joint_attention = enable
    return model


def set_joint_scale(model, scale=1.0):
    for _, m in _patched_blocks(model):
        m.joint_scale = scale
    return model


def set_joint_attention_mask(model, joint_attn_mask):
    dm = _model(model)
    mask = torch.tensor(joint_attn_mask, dtype=torch.bool)
    dm._joint_attn_mask = mask
    for _, m in _patched_blocks(model):
        m.joint_attn_mask = mask
    return model


# ---- masked-LoRA plumbing (patch/patch.py:57-92,:872-936); execution: lkgd_amd/lora.py + lkgd_amd/unet.py ----------
def _models(model):
    dm = _model(model)
    return [dm] + ([model.controlnet] if getattr(model, "controlnet", None) is not None else [])


def set_patch_lora_mask(model, lora_name, lora_mask):
    """patch/patch.py:872-896: one bool per batch entry; K / V of the joint attention take the inverted mask"""
    from .lora import Linear
    mask = torch.tensor(lora_mask, dtype=torch.bool)
    for dm in _models(model):
        if not hasattr(dm, "lora_mask"):
            dm.lora_mask = dict()
        dm.lora_mask[lora_name] = mask
        for name, m in dm.named_modules():
            if isinstance(m, Linear):
                if not hasattr(m, "lora_mask"):
                    m.lora_mask = dict()
                m.lora_mask[lora_name] = ~mask if ("attn1n.to_k" in name or "attn1n.to_v" in name) else mask
        if hasattr(dm, "invalidate"):
            dm.invalidate()
    return model


def hack_lora_forward(model):
    """patch/patch.py:912-922: every LoRA-wrapped projection switches to the masked forward (:57-92)"""
    from .lora import Linear
    for dm in _models(model):
        for _, m in dm.named_modules():
            if isinstance(m, Linear):
                m._lkgd_masked = True
        if hasattr(dm, "invalidate"):
            dm.invalidate()
    return model


def initialize_joint_lora(model, adapter_name, joint_adapter_name):
    """patch/patch.py:185-197,:925-936: attn1n's adapter `joint_adapter_name` := attn1's adapter `adapter_name`"""
    from .lora import Linear
    for dm in _models(model):
        for _, blk in dm.named_modules():
            if not getattr(blk, "_lkgd_patched", False) or not hasattr(blk, "attn1n"):
                continue
            for name, m in blk.attn1n.named_modules():
                if not isinstance(m, Linear):
                    continue
                src = blk.attn1.get_submodule(name)
                for layer_name in m.adapter_layer_names:
                    dst_dict, src_dict = getattr(m, layer_name), getattr(src, layer_name)
                    if joint_adapter_name in dst_dict:
                        dst_dict[joint_adapter_name].load_state_dict(src_dict[adapter_name].state_dict())
        if hasattr(dm, "invalidate"):
            dm.invalidate()
    return model


def set_joint_layer_requires_grad(model, adapter_names, requires_grad):
    """patch/patch.py:111-133,:898-910 (a training knob; literal): attn1n's adapter parameters and the post layers"""
    from .lora import Linear
    for dm in _models(model):
        for _, blk in dm.named_modules():
            if not getattr(blk, "_lkgd_patched", False) or not hasattr(blk, "attn1n"):
                continue
            for _, m in blk.attn1n.named_modules():
                if isinstance(m, Linear):
                    for layer_name in m.adapter_layer_names:
                        for key, layer in getattr(m, layer_name).items():
                            if key in adapter_names:
                                layer.requires_grad_(requires_grad)
            for post in ("conv1n", "scale1n"):
                if hasattr(blk, post):
                    getattr(blk, post).requires_grad_(requires_grad)
    return model
