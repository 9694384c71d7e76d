"""fp32 restatement of the reference's masked LoRA forward.  ORACLE - test infrastructure only.

Follows /root/reference/patch/patch.py:57-92 (``lora_forward_hack``: ``result = base(x); result[mask] += B(A(x[mask])) *
scaling`` per active adapter, mask repeat-interleaved over the leading rows of x), the unhacked peft forward
/root/reference/models/lora_layer.py:417-443, ``set_patch_lora_mask`` patch.py:872-896 (inverted mask on
``attn1n.to_k`` / ``attn1n.to_v``), ``hack_lora_forward`` :911-922, and the parameter layout of the vendored peft layer
(lora_layer.py:37-130: ``base_layer``, ``lora_A`` / ``lora_B`` ModuleDicts, ``scaling = lora_alpha / r``).

Pinned: tests/golden/patch_lora.safetensors holds outputs of the reference's own ``lora_forward_hack`` + vendored
``lora_layer.Linear`` run inside its UNet ``forward`` (tests/golden/make_goldens.py::gen_patch_lora).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn


class Linear(nn.Module):
    def __init__(self, base_layer: nn.Linear):
        super().__init__()
        self.base_layer = base_layer
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()
        self.scaling: Dict[str, float] = {}
        self.active_adapters: List[str] = []
        self.lora_mask: Dict[str, torch.Tensor] = {}
        self.masked = False            # hack_lora_forward applied
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features

    def add(self, name: str, r: int, lora_alpha: float) -> None:
        self.lora_A[name] = nn.Linear(self.in_features, r, bias=False)
        self.lora_B[name] = nn.Linear(r, self.out_features, bias=False)
        self.scaling[name] = lora_alpha / r
        if name not in self.active_adapters:
            self.active_adapters.append(name)

    def forward(self, x):
        result = self.base_layer(x)
        for a in self.active_adapters:
            if a not in self.lora_A:
                continue
            if not self.masked:                                     # lora_layer.py:417-443
                result = result + self.lora_B[a](self.lora_A[a](x)) * self.scaling[a]
                continue
            m = self.lora_mask[a]                                   # patch.py:76-90
            m = m.repeat_interleave(x.shape[0] // len(m), dim=0)
            result[m] += self.lora_B[a](self.lora_A[a](x[m])) * self.scaling[a]
        return result


TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


def inject(model: nn.Module, names: Sequence[str], r: int, lora_alpha: float, targets: Sequence[str] = TARGETS) -> None:
    """wrap every projection whose dotted name ends in one of `targets` (peft's suffix rule), one adapter per name"""
    for name, m in list(model.named_modules()):
        if not isinstance(m, nn.Linear) or ".lora_" in name or name.endswith("base_layer"):
            continue
        if not any(name == t or name.endswith("." + t) for t in targets):
            continue
        parent_name, _, child = name.rpartition(".")
        parent = model.get_submodule(parent_name)
        w = Linear(m)
        for n in names:
            w.add(n, r, lora_alpha)
        if isinstance(parent, nn.ModuleList):
            parent[int(child)] = w
        else:
            setattr(parent, child, w)


def hack_lora_forward(model: nn.Module) -> None:
    for m in model.modules():
        if isinstance(m, Linear):
            m.masked = True


def set_patch_lora_mask(model: nn.Module, lora_name: str, lora_mask) -> None:
    mask = torch.tensor(lora_mask, dtype=torch.bool)
    for name, m in model.named_modules():
        if isinstance(m, Linear):
            m.lora_mask[lora_name] = ~mask if ("attn1n.to_k" in name or "attn1n.to_v" in name) else mask


def set_adapters(model: nn.Module, names: Sequence[str]) -> None:
    for m in model.modules():
        if isinstance(m, Linear):
            m.active_adapters = [n for n in names if n in m.lora_A]
