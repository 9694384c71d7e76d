"""LoRA adapters on the attention projections of the lkgd_amd UNet, including the reference's per-batch-entry MASKED forward.

Mirrors
* the peft layer the reference vendors, /root/reference/models/lora_layer.py:268-443 (``Linear``: ``base_layer`` +
  ``lora_A`` / ``lora_B`` ModuleDicts, ``scaling = lora_alpha / r``, ``merge`` / ``get_delta_weight`` :300-415) - here a
  PARAMETER HOLDER with the same attribute and state-dict names (``...to_q.base_layer.weight``,
  ``...to_q.lora_A.<adapter>.weight``), never called;
* /root/reference/patch/patch.py:57-92 (``lora_forward_hack``):
      result = base(x);  result[lora_mask] += lora_B(lora_A(x[lora_mask])) * scaling      per active adapter,
  ``lora_mask`` (bool, one entry per batch entry, :872-896) repeat-interleaved over the leading rows of x; the K / V
  projections of the joint attention ``attn1n`` carry the INVERTED mask (:889-892) because their input is the partner
  entry's hidden states;
* the loader call sequence /root/reference/utils/util.py:570-606 (``lora_state_dict`` -> ``load_lora_into_unet`` per
  adapter, ``set_adapters``, ``hack_lora_forward``, ``set_patch_lora_mask``).

MI355X execution: a low-rank update that applies to whole batch entries is a different WEIGHT for those entries, so the
masked forward costs nothing extra: ``W_eff(entry) = W + sum_{adapters active on the entry} scaling * B @ A`` is folded in
fp32 at pack time (one packed variant per distinct adapter subset, built on demand and cached), and a projection runs as one
GEMM launch per run of consecutive batch entries that share a variant (rows of a batch entry are contiguous in the
[(batch, frame, y, x), C] token layout of both the spatial and the temporal blocks).  All-ones masks / the plain peft
forward are the one-variant case; ``merge_lora`` makes that permanent.
"""
from __future__ import annotations

import math
import os
import re
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ._lib import LkgdHipError


class Linear(nn.Module):
    """peft ``lora.Linear`` as a parameter holder (class NAME as in peft: the reference tests ``isinstance(m, Linear)``)"""

    adapter_layer_names = ("lora_A", "lora_B")

    def __init__(self, base_layer: nn.Linear, adapter_name: Optional[str] = None, r: int = 0, lora_alpha: float = 1.0,
                 init_lora_weights: bool = True):
        super().__init__()
        if not isinstance(base_layer, nn.Linear):
            raise LkgdHipError(f"LoRA targets nn.Linear projections here (got {type(base_layer).__name__})")
        self.base_layer = base_layer
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()
        self.lora_dropout = nn.ModuleDict()
        self.r: Dict[str, int] = {}
        self.lora_alpha: Dict[str, float] = {}
        self.scaling: Dict[str, float] = {}
        self.use_dora: Dict[str, bool] = {}
        self.merged_adapters: List[str] = []
        self.disable_adapters = False
        self._active_adapter: List[str] = []
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        if adapter_name is not None:
            self.update_layer(adapter_name, r, lora_alpha, init_lora_weights)
            self._active_adapter = [adapter_name]

    # the packing code of lkgd_amd.unet reads `.weight` / `.bias` of a projection: the base layer's
    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    @property
    def merged(self) -> bool:
        return bool(self.merged_adapters)

    @property
    def active_adapters(self) -> List[str]:
        return list(self._active_adapter)

    def set_adapter(self, names) -> None:
        self._active_adapter = [names] if isinstance(names, str) else list(names)

    def update_layer(self, adapter_name: str, r: int, lora_alpha: float, init_lora_weights: bool = True) -> None:
        """lora_layer.py:85-130 (no dropout at inference, no rslora / DoRA)"""
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        w = self.base_layer.weight
        self.r[adapter_name], self.lora_alpha[adapter_name] = r, lora_alpha
        self.scaling[adapter_name] = lora_alpha / r
        self.use_dora[adapter_name] = False
        self.lora_dropout[adapter_name] = nn.Identity()
        self.lora_A[adapter_name] = nn.Linear(self.in_features, r, bias=False, device=w.device, dtype=w.dtype)
        self.lora_B[adapter_name] = nn.Linear(r, self.out_features, bias=False, device=w.device, dtype=w.dtype)
        if init_lora_weights:       # lora_layer.py:132-149: A kaiming-uniform, B zero => the adapter starts as identity
            nn.init.kaiming_uniform_(self.lora_A[adapter_name].weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B[adapter_name].weight)

    def get_delta_weight(self, adapter: str) -> torch.Tensor:
        """lora_layer.py:383-415: B @ A * scaling, fp32"""
        a = self.lora_A[adapter].weight.detach().to(torch.float32)
        b = self.lora_B[adapter].weight.detach().to(torch.float32)
        return (b @ a) * float(self.scaling[adapter])

    def effective_weight(self, adapters: Iterable[str]) -> torch.Tensor:
        """fp32 base weight + the deltas of `adapters` (those already merged into the base are skipped)"""
        w = self.base_layer.weight.detach().to(torch.float32)
        if self.disable_adapters:
            return w
        for a in adapters:
            if a in self.lora_A and a not in self.merged_adapters:
                w = w + self.get_delta_weight(a)
        return w

    def merge(self, adapter_names: Optional[Sequence[str]] = None) -> None:
        """lora_layer.py:300-361"""
        for a in (self.active_adapters if adapter_names is None else adapter_names):
            if a in self.lora_A and a not in self.merged_adapters:
                with torch.no_grad():
                    w = self.base_layer.weight
                    w.copy_((w.detach().to(torch.float32) + self.get_delta_weight(a)).to(w.dtype))
                self.merged_adapters.append(a)

    def extra_repr(self) -> str:
        return f"adapters={list(self.lora_A.keys())}, active={self._active_adapter}"


# ------------------------------------------------------------------------------------------------ model-level helpers
def _model(model):
    return model.unet if hasattr(model, "unet") else model


def lora_layers(model) -> List[Tuple[str, Linear]]:
    return [(n, m) for n, m in _model(model).named_modules() if isinstance(m, Linear)]


def _matches(name: str, targets: Sequence[str]) -> bool:
    """peft's target_modules rule: exact module name or dotted suffix"""
    return any(name == t or name.endswith("." + t) for t in targets)


_ATTN_PROJ = re.compile(r"(^|\.)(attn1|attn2|attn1n)\.(to_q|to_k|to_v|to_out\.0)$")


def add_adapter(model, adapter_name: str, r: int = 8, lora_alpha: float = 8.0,
                target_modules: Sequence[str] = ("to_k", "to_q", "to_v", "to_out.0"), init_lora_weights: bool = True,
                ranks: Optional[Dict[str, int]] = None, alphas: Optional[Dict[str, float]] = None) -> List[str]:
    """peft ``inject_adapter`` for the projections the reference targets (train_models/*: ``target_modules`` are the
    attention projections): wraps the matching nn.Linear layers in ``lora.Linear`` (or adds the adapter to an existing
    wrapper).  Returns the wrapped module names."""
    dm = _model(model)
    done = []
    for name, m in list(dm.named_modules()):
        if isinstance(m, Linear):
            key = name
        elif isinstance(m, nn.Linear) and not name.endswith(".base_layer") and ".lora_A." not in name \
                and ".lora_B." not in name:
            key = name
        else:
            continue
        if not _matches(key, target_modules):
            continue
        if not _ATTN_PROJ.search(key):
            raise LkgdHipError(f"LoRA on '{key}': the HIP path folds adapters into the attention projections "
                               "(to_q / to_k / to_v / to_out.0 of attn1, attn2, attn1n) only")
        rr = (ranks or {}).get(key, r)
        aa = (alphas or {}).get(key, lora_alpha)
        if isinstance(m, Linear):
            m.update_layer(adapter_name, rr, aa, init_lora_weights)
        else:
            parent_name, _, child = key.rpartition(".")
            parent = dm.get_submodule(parent_name) if parent_name else dm
            wrapped = Linear(m, adapter_name, rr, aa, init_lora_weights)
            if isinstance(parent, (nn.ModuleList, nn.Sequential)):
                parent[int(child)] = wrapped
            else:
                setattr(parent, child, wrapped)
        done.append(key)
    if not done:
        raise ValueError(f"Target modules {list(target_modules)} not found in the base model.")
    _invalidate(dm)
    return done


def lora_state_dict(path: str, weight_name: Optional[str] = None):
    """``StableDiffusionPipeline.lora_state_dict(dir)`` [EXT diffusers loaders/lora.py]: reads
    ``pytorch_lora_weights.safetensors`` (or ``.bin``) and splits off the ``.alpha`` entries -> (state_dict, network_alphas)"""
    if os.path.isdir(path):
        for cand in ([weight_name] if weight_name else ["pytorch_lora_weights.safetensors", "pytorch_lora_weights.bin"]):
            f = os.path.join(path, cand)
            if os.path.exists(f):
                path = f
                break
        else:
            raise FileNotFoundError(f"no pytorch_lora_weights.safetensors / .bin under {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    alphas = {k: float(v) for k, v in sd.items() if k.endswith(".alpha")}
    sd = {k: v for k, v in sd.items() if not k.endswith(".alpha")}
    return sd, (alphas or None)


def _split_key(k: str, adapter_name: str):
    """-> (module path, 'A' | 'B') for the key formats diffusers / peft write"""
    if k.startswith("unet."):
        k = k[len("unet."):]
    for pat, which in ((f".lora_A.{adapter_name}.weight", "A"), (f".lora_B.{adapter_name}.weight", "B"),
                       (".lora_A.weight", "A"), (".lora_B.weight", "B"),
                       (".lora.down.weight", "A"), (".lora.up.weight", "B"),
                       ("_lora.down.weight", "A"), ("_lora.up.weight", "B")):
        if k.endswith(pat):
            return k[:-len(pat)], which
    return None, None


def load_lora_into_unet(state_dict: Dict[str, torch.Tensor], network_alphas: Optional[Dict[str, float]], unet,
                        adapter_name: str = "default") -> List[str]:
    """``StableDiffusionPipeline.load_lora_into_unet(state_dict, network_alphas, unet=, adapter_name=)`` [EXT] as the
    reference calls it (utils/util.py:574-576): rank per module from the lora_A shapes, alpha from ``network_alphas`` (default
    = rank, i.e. scaling 1), adapters injected into the matching projections and filled.  Keys of other components
    (``text_encoder.``) are ignored."""
    dm = _model(unet)
    per: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in state_dict.items():
        if k.startswith("text_encoder"):
            continue
        mod, which = _split_key(k, adapter_name)
        if mod is None:
            raise ValueError(f"unrecognised LoRA key '{k}'")
        per.setdefault(mod, {})[which] = v
    if not per:
        raise ValueError("empty LoRA state dict")
    ranks, alphas = {}, {}
    for mod, ab in per.items():
        if "A" not in ab or "B" not in ab:
            raise ValueError(f"LoRA module '{mod}' lacks its {'A' if 'A' not in ab else 'B'} matrix")
        ranks[mod] = ab["A"].shape[0]
        alphas[mod] = ranks[mod]
        if network_alphas:
            for cand in (mod + ".alpha", "unet." + mod + ".alpha"):
                if cand in network_alphas:
                    alphas[mod] = float(network_alphas[cand])
    add_adapter(dm, adapter_name, target_modules=list(per.keys()), init_lora_weights=False, ranks=ranks, alphas=alphas)
    with torch.no_grad():
        for mod, ab in per.items():
            w = dm.get_submodule(mod)
            if tuple(w.lora_A[adapter_name].weight.shape) != tuple(ab["A"].shape) or \
                    tuple(w.lora_B[adapter_name].weight.shape) != tuple(ab["B"].shape):
                raise ValueError(f"LoRA shapes of '{mod}' do not fit the projection "
                                 f"({tuple(ab['A'].shape)}, {tuple(ab['B'].shape)})")
            w.lora_A[adapter_name].weight.copy_(ab["A"])
            w.lora_B[adapter_name].weight.copy_(ab["B"])
    set_adapters(dm, [adapter_name])
    return list(per.keys())


def set_adapters(model, adapter_names, weights=None) -> None:
    """``unet.set_adapters(names, weights)`` [EXT diffusers PeftAdapterMixin]: activates the named adapters on every wrapper;
    per-adapter weights scale ``scaling`` (peft ``set_scale``: scaling = weight * lora_alpha / r)"""
    names = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
    ws = [1.0] * len(names) if weights is None else ([weights] * len(names) if isinstance(weights, (int, float)) else list(weights))
    if len(ws) != len(names):
        raise ValueError(f"Length of adapter names {len(names)} is not equal to the length of the weights {len(ws)}")
    dm = _model(model)
    for _, m in lora_layers(dm):
        m.set_adapter([n for n in names if n in m.lora_A])
        for n, w in zip(names, ws):
            if n in m.lora_A:
                m.scaling[n] = float(w) * m.lora_alpha[n] / m.r[n]
    _invalidate(dm)


def merge_lora(model, adapter_names: Optional[Sequence[str]] = None, unload: bool = True) -> None:
    """W += scaling * B @ A for the (active) adapters of every wrapper; with ``unload`` the wrappers are replaced by their
    base layers again (peft ``merge_and_unload``).  Masked adapters merge only where their mask is constant over the entries."""
    dm = _model(model)
    todo = []
    for name, m in lora_layers(dm):
        names = list(m.active_adapters if adapter_names is None else adapter_names)
        if getattr(m, "_lkgd_masked", False):
            keep = []
            for a in names:
                mk = getattr(m, "lora_mask", {}).get(a)
                if mk is None or bool(mk.all()):
                    keep.append(a)
                elif bool(mk.any()):
                    raise LkgdHipError(f"merge_lora: adapter '{a}' on '{name}' carries a per-entry mask that is neither all "
                                       "ones nor all zeros; a masked adapter is not a single weight")
                # an all-zero mask (e.g. attn1n.to_k / to_v under the loader's single-LoRA masks [1,1,1,1], inverted by
                # set_patch_lora_mask) means the adapter never acts on this projection: nothing to merge
            names = keep
        todo.append((name, m, names))
    for name, m, names in todo:
        m.merge(names)
        if unload:
            parent_name, _, child = name.rpartition(".")
            parent = dm.get_submodule(parent_name) if parent_name else dm
            if isinstance(parent, (nn.ModuleList, nn.Sequential)):
                parent[int(child)] = m.base_layer
            else:
                setattr(parent, child, m.base_layer)
    _invalidate(dm)


def _invalidate(dm) -> None:
    if hasattr(dm, "invalidate"):
        dm.invalidate()


# ------------------------------------------------------------------------------------------------ per-entry plan
class EntryPlan:
    """Which adapters act on which batch entry, for one forward.

    ``runs``: [(b0, b1)] maximal runs of consecutive batch entries on which EVERY wrapped projection of the model sees the
    same adapter subset; ``adapters(wrapper, run_index, via_partner)`` -> tuple of adapter names for that run."""

    def __init__(self, layers: List[Tuple[str, Linear]], B: int, partner: Optional[List[int]],
                 B_total: Optional[int] = None, b0: int = 0):
        """``B`` entries of this forward = entries [b0, b0 + B) of a UNet batch of ``B_total`` (a CFG-parallel rank holds one CFG
        half of the call's batch, lkgd_amd/dist_run.py; the lora masks describe the WHOLE batch); ``partner`` is local"""
        self.B = B
        B_total = B if B_total is None else B_total
        self._dec: Dict[int, List[Tuple[str, ...]]] = {}     # id(wrapper) -> per batch entry adapter tuple
        self._dec_p: Dict[int, List[Tuple[str, ...]]] = {}   # the same seen through the partner permutation
        sig = [[] for _ in range(B)]
        for name, m in layers:
            masked = getattr(m, "_lkgd_masked", False)
            per = []
            for b in range(b0, b0 + B):
                act = []
                for a in m.active_adapters:
                    if a not in m.lora_A:
                        continue
                    if not masked:
                        act.append(a)
                        continue
                    masks = getattr(m, "lora_mask", None)
                    if not masks or a not in masks:
                        raise LkgdHipError(f"'{name}': hack_lora_forward is active but adapter '{a}' has no lora_mask "
                                           "(patch.set_patch_lora_mask) - the reference raises KeyError here")
                    mk = masks[a]
                    L = len(mk)
                    if B_total % L:
                        raise LkgdHipError(f"lora_mask of length {L} does not divide the UNet batch of {B_total} entries")
                    if bool(mk[b // (B_total // L)]):
                        act.append(a)
                per.append(tuple(act))
            self._dec[id(m)] = per
            pp = [per[partner[b]] for b in range(B)] if partner is not None else per
            self._dec_p[id(m)] = pp
            for b in range(B):
                sig[b].append((per[b], pp[b]))
        self.runs: List[Tuple[int, int]] = []
        b0 = 0
        for b in range(1, B + 1):
            if b == B or sig[b] != sig[b0]:
                self.runs.append((b0, b))
                b0 = b

    def adapters(self, wrapper, run: int, via_partner: bool = False) -> Tuple[str, ...]:
        if not isinstance(wrapper, Linear):
            return ()
        return (self._dec_p if via_partner else self._dec)[id(wrapper)][self.runs[run][0]]


def entry_plan(unet, B: int, partner: Optional[List[int]], B_total: Optional[int] = None, b0: int = 0) -> Optional[EntryPlan]:
    """None when the model has no LoRA wrappers (the common case: nothing changes on the hot path)"""
    layers = lora_layers(unet)
    if not layers:
        return None
    return EntryPlan(layers, B, partner, B_total, b0)
