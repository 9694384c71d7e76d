"""``from_pretrained`` / ``save_pretrained`` for the lkgd_amd models: the diffusers on-disk layout read directly.

The reference loads everything through diffusers' ``ModelMixin.from_pretrained`` / ``DiffusionPipeline.from_pretrained``
[EXT] (call sites: /root/reference/run_models/run_inference_svd.py:166-168, utils/util.py:536,607-616).  A model directory
is ``config.json`` + ``diffusion_pytorch_model[.<variant>].safetensors`` (or a sharded ``*.safetensors.index.json``, or
``.bin``); a pipeline directory holds one sub-folder per component (``unet/``, ``vae/``, ``image_encoder/``,
``feature_extractor/``, ``scheduler/scheduler_config.json``) and a ``model_index.json``.  Reading that needs no diffusers:
json + safetensors.  Weights go host -> HBM once; nothing here is on the hot path.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

WEIGHTS_NAME = "diffusion_pytorch_model"


def _dir(path: str, subfolder: Optional[str]) -> str:
    d = os.path.join(path, subfolder) if subfolder else path
    if not os.path.isdir(d):
        raise OSError(f"{d} is not a directory (only local directories are supported: there is no hub access)")
    return d


def load_config(path: str, subfolder: Optional[str] = None, name: str = "config.json") -> Dict:
    f = os.path.join(_dir(path, subfolder), name)
    if not os.path.exists(f):
        raise OSError(f"no {name} under {os.path.dirname(f)}")
    with open(f) as fh:
        return json.load(fh)


def load_state_dict(path: str, subfolder: Optional[str] = None, variant: Optional[str] = None,
                    weights_name: str = WEIGHTS_NAME) -> Dict[str, torch.Tensor]:
    """safetensors first (single file, then sharded index), ``.bin`` last - diffusers' own order; ``variant`` ('fp16')
    selects ``<name>.<variant>.safetensors`` and falls back to the plain name"""
    d = _dir(path, subfolder)
    stems = ([f"{weights_name}.{variant}"] if variant else []) + [weights_name]
    for stem in stems:
        f = os.path.join(d, stem + ".safetensors")
        if os.path.exists(f):
            from safetensors.torch import load_file
            return load_file(f)
        idx = os.path.join(d, stem + ".safetensors.index.json")
        if os.path.exists(idx):
            from safetensors.torch import load_file
            with open(idx) as fh:
                shards = sorted(set(json.load(fh)["weight_map"].values()))
            sd: Dict[str, torch.Tensor] = {}
            for s in shards:
                sd.update(load_file(os.path.join(d, s)))
            return sd
        f = os.path.join(d, stem + ".bin")
        if os.path.exists(f):
            return torch.load(f, map_location="cpu", weights_only=True)
    raise OSError(f"no {weights_name}[.{variant}].safetensors / .bin under {d}")


def save_pretrained(model: torch.nn.Module, path: str, config: Dict, class_name: str, variant: Optional[str] = None) -> None:
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = {"_class_name": class_name, "_lkgd_amd": True}
    cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in config.items()})
    with open(os.path.join(path, "config.json"), "w") as fh:
        json.dump(cfg, fh, indent=2)
    name = WEIGHTS_NAME + (f".{variant}" if variant else "") + ".safetensors"
    save_file({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}, os.path.join(path, name))


def build_from_pretrained(cls, config_cls, path: str, subfolder: Optional[str] = None, torch_dtype=None,
                          variant: Optional[str] = None, strict: bool = True, **extra_ctor):
    """shared body of the models' ``from_pretrained``: config keys the dataclass knows are used, the rest (diffusers
    bookkeeping such as ``_class_name`` / ``_diffusers_version``) ignored; parameters are created on the meta device and
    filled from the checkpoint (no random init of 1.5 B parameters)"""
    raw = load_config(path, subfolder)
    known = {k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items()
             if k in config_cls.__dataclass_fields__}
    cfg = config_cls(**known)
    extra = {k: (tuple(raw[k]) if isinstance(raw[k], list) else raw[k]) for k in extra_ctor if k in raw}
    extra_ctor = {**extra_ctor, **extra}
    sd = load_state_dict(path, subfolder, variant)
    with torch.device("meta"):
        m = cls(cfg, **extra_ctor)
    m = m.to_empty(device="cpu")
    if torch_dtype is not None:
        m = m.to(torch_dtype)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError(f"{cls.__name__}.from_pretrained({path!r}): missing keys {list(missing)[:5]}"
                           f"{'...' if len(missing) > 5 else ''}, unexpected keys {list(unexpected)[:5]}"
                           f"{'...' if len(unexpected) > 5 else ''}")
    m._name_or_path = os.path.join(path, subfolder) if subfolder else path
    return m
