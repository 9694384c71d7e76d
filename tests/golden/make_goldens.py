#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by EXECUTING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference, read-only; never on the GPU box).  The reference's hot-path
files import `diffusers`, `peft` and `core_qnn`, none of which is installed or installable here, so this script
registers NAME-ONLY stubs for those packages in ``sys.modules`` (config plumbing, base classes, logging) and binds the
block classes the reference asks diffusers for to the restatements in ``oracle/blocks.py``.  What the fixtures pin is
therefore everything the reference itself implements on the path:

* scheduler_kat.json      <- utils/scheduling_euler_discrete_karras_fix.py (whole file, unmodified)
* unet_wiring.safetensors <- models/unet_spatio_temporal_condition_controlnet.py ``forward`` (stock signature, with
                             and without ControlNet residuals) and models/unet_spatio_temporal_condition.py
                             ``forward`` (LK signature incl. the latent-knowledge fuse), tiny config
* patch_joint.safetensors <- patch/patch.py ``apply_patch`` + ``ToMeBlock.forward`` / ``forward_temporal`` joint branch
* patch_fsm.safetensors   <- patch/patch_FSM.py ``apply_patch`` + ``ToMeBlock.forward`` FSM branch (:380-441)
* controlnet.safetensors  <- models/controlnet_sdv.py ``ControlNetSDVModel.forward`` (conditioning embedding, encoder,
                             zero convolutions, conditioning_scale) and its residuals fed to the stock UNet
* loop.safetensors        <- pipeline/pipeline_stable_video_diffusion_trans.py ``__call__`` (output_type="latent")
                             with stand-in CLIP/VAE stages (boundary stages, outside the hot path)

Weights are NOT stored: they are regenerated from ``oracle.unet.init_weights_(seed)`` (a checksum is stored).
No reference source text is copied into the repo; fixtures are tensors and scalars only.

Usage:  python tests/golden/make_goldens.py
"""
from __future__ import annotations

import enum
import functools
import importlib.util
import inspect
import json
import logging as _pylogging
import os
import sys
import types
from types import SimpleNamespace

sys.dont_write_bytecode = True   # /root/reference is read-only: never drop __pycache__ there

import numpy as np
import torch
import torch.nn as nn
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import blocks as ob            # noqa: E402
from oracle import unet as ou              # noqa: E402


# ------------------------------------------------------------------------------------------------ name-only stubs
def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


class _FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    """diffusers ConfigMixin, reduced to: ``register_to_config`` storage + attribute fallback to the config."""
    config_name = None

    def register_to_config(self, **kw):
        object.__setattr__(self, "_internal_dict", _FrozenDict(kw))

    @property
    def config(self):
        return self._internal_dict

    def __getattr__(self, name):
        d = self.__dict__.get("_internal_dict")
        if d is not None and name in d:
            return d[name]
        sup = super()
        if hasattr(sup, "__getattr__"):
            return sup.__getattr__(name)      # nn.Module parameter/buffer/submodule lookup
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        params = {n: p.default for i, (n, p) in enumerate(inspect.signature(init).parameters.items()) if i > 0}
        new = {}
        for a, n in zip(args, params.keys()):
            new[n] = a
        new.update({k: kwargs.get(k, d) for k, d in params.items() if k not in new})
        self.register_to_config(**new)
        init(self, *args, **kwargs)
    return inner


class BaseOutput:
    pass


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)


def install_stubs():
    d = _mod("diffusers")
    cu = _mod("diffusers.configuration_utils")
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    ut = _mod("diffusers.utils")
    ut.BaseOutput, ut.logging = BaseOutput, _Logging
    ut.replace_example_docstring = lambda s: (lambda f: f)
    tu = _mod("diffusers.utils.torch_utils")
    tu.randn_tensor = randn_tensor
    tu.is_compiled_module = lambda m: False
    _mod("diffusers.schedulers")
    su = _mod("diffusers.schedulers.scheduling_utils")
    su.KarrasDiffusionSchedulers = enum.Enum("KarrasDiffusionSchedulers", "EulerDiscreteScheduler DDIMScheduler")
    su.SchedulerMixin = type("SchedulerMixin", (), {})
    ld = _mod("diffusers.loaders")
    for n in ("UNet2DConditionLoadersMixin", "PeftAdapterMixin", "FromSingleFileMixin", "FromOriginalModelMixin"):
        setattr(ld, n, type(n, (), {}))
    dm = _mod("diffusers.models")
    dm.AutoencoderKLTemporalDecoder = type("AutoencoderKLTemporalDecoder", (), {})
    dm.UNetSpatioTemporalConditionModel = type("UNetSpatioTemporalConditionModel", (), {})
    ap = _mod("diffusers.models.attention_processor")
    ap.CROSS_ATTENTION_PROCESSORS = ()
    ap.AttentionProcessor = type("AttentionProcessor", (), {})
    ap.AttnProcessor = type("AttnProcessor", (), {})
    ap.Attention = ob.Attention
    ap.ADDED_KV_ATTENTION_PROCESSORS = ()
    ap.AttnAddedKVProcessor = type("AttnAddedKVProcessor", (), {})
    em = _mod("diffusers.models.embeddings")
    em.TimestepEmbedding, em.Timesteps = ob.TimestepEmbedding, ob.Timesteps
    for n in ("TextImageProjection", "TextImageTimeEmbedding", "TextTimeEmbedding"):
        setattr(em, n, type(n, (nn.Module,), {}))
    mu = _mod("diffusers.models.modeling_utils")
    mu.ModelMixin = type("ModelMixin", (nn.Module,), {})
    _mod("diffusers.models.unets")
    b3 = _mod("diffusers.models.unets.unet_3d_blocks")
    b3.UNetMidBlockSpatioTemporal = ob.UNetMidBlockSpatioTemporal
    b3.get_down_block, b3.get_up_block = ob.get_down_block, ob.get_up_block
    nm = _mod("diffusers.models.normalization")
    for n in ("AdaLayerNorm", "AdaLayerNormContinuous", "AdaLayerNormZero", "RMSNorm"):
        setattr(nm, n, type(n, (nn.Module,), {}))
    ip = _mod("diffusers.image_processor")

    class VaeImageProcessor:        # stand-in for a BOUNDARY stage (outside the hot path)
        def __init__(self, vae_scale_factor=8):
            self.vae_scale_factor = vae_scale_factor

        def preprocess(self, image, height=None, width=None):
            return 2.0 * image - 1.0
    ip.VaeImageProcessor, ip.PipelineImageInput = VaeImageProcessor, object
    _mod("diffusers.pipelines")
    pu = _mod("diffusers.pipelines.pipeline_utils")

    class DiffusionPipeline:
        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def _execution_device(self):
            return torch.device("cpu")

        def progress_bar(self, total=None):
            class _PB:
                def __enter__(s): return s
                def __exit__(s, *a): return False
                def update(s): pass
            return _PB()

        def maybe_free_model_hooks(self):
            pass
    pu.DiffusionPipeline = DiffusionPipeline

    _mod("peft"); _mod("peft.tuners"); _mod("peft.tuners.lora")
    # peft is not installed: bookkeeping-only stand-in for its BaseTunerLayer [EXT peft tuners_utils.py] (which adapters
    # are active / merged, the base-layer accessor) so that the reference's VENDORED peft layer
    # (/root/reference/models/lora_layer.py - the arithmetic: update_layer, get_delta_weight, merge, forward) can be
    # executed and patch.py's `Linear` / `lora_forward_hack` (:57-92) run over it
    tu_ = _mod("peft.tuners.tuners_utils")

    class BaseTunerLayer:
        adapter_layer_names = ()
        other_param_names = ()
        _disable_adapters = False
        _active_adapter = "default"
        merged_adapters = []

        def get_base_layer(self):
            base = self
            while hasattr(base, "base_layer"):
                base = base.base_layer
            return base

        @property
        def weight(self):
            return self.get_base_layer().weight

        @property
        def bias(self):
            return self.get_base_layer().bias

        @property
        def merged(self):
            return bool(self.merged_adapters)

        @property
        def disable_adapters(self):
            return self._disable_adapters

        @property
        def active_adapter(self):
            return self._active_adapter

        @property
        def active_adapters(self):
            return [self._active_adapter] if isinstance(self._active_adapter, str) else self._active_adapter

        def set_adapter(self, adapter_names):
            self._active_adapter = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
    tu_.BaseTunerLayer = BaseTunerLayer
    tu_.check_adapters_to_merge = lambda module, adapter_names=None: list(adapter_names or module.active_adapters)
    _mod("peft.utils")
    import contextlib
    _mod("peft.utils.integrations").gather_params_ctx = lambda *a, **k: contextlib.nullcontext()
    _mod("peft.utils.other").transpose = lambda w, fan_in_fan_out: w.T if fan_in_fan_out else w
    _mod("peft.tuners.lora.config").LoraConfig = type("LoraConfig", (), {})
    pl = _mod("peft.tuners.lora.layer")
    vendored = load_ref("models/lora_layer.py", "ref_lora_layer")
    pl.Linear = vendored.Linear
    pl.BaseTunerLayer = BaseTunerLayer
    _mod("core_qnn")
    q = _mod("core_qnn.quaternion_layers")
    q.QuaternionLinearAutograd = ob.QuaternionLinearAutograd
    q.__all__ = ["QuaternionLinearAutograd"]


def load_ref(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def checksum(model) -> float:
    return float(sum(p.double().abs().sum() for p in model.parameters()))


# ------------------------------------------------------------------------------------------------ 1. scheduler
def gen_scheduler(sched_mod):
    from oracle.scheduler import SchedulerConfig
    cfg = SchedulerConfig().__dict__
    out = {"config": cfg, "cases": []}
    for n in (2, 3, 25):
        s = sched_mod.EulerDiscreteScheduler(**cfg)
        s.set_timesteps(n)
        g = torch.Generator().manual_seed(100 + n)
        x = torch.randn(1, 2, 4, 3, 3, generator=g) * float(s.init_noise_sigma)
        case = {"n": n, "sigmas": s.sigmas.tolist(), "timesteps": s.timesteps.tolist(),
                "init_noise_sigma": float(s.init_noise_sigma), "seed": 100 + n,
                "x0": x.flatten().tolist(), "steps": []}
        for t in s.timesteps:
            v = torch.randn(x.shape, generator=g)
            scaled = s.scale_model_input(x, t)
            x = s.step(v, t, x).prev_sample
            case["steps"].append({"v": v.flatten().tolist(), "scaled": scaled.flatten().tolist(),
                                  "prev": x.flatten().tolist()})
        out["cases"].append(case)
    # the scalar KAT quoted in SURVEY.md 8c: x=1, v=0.5, first of 25 steps
    s = sched_mod.EulerDiscreteScheduler(**cfg)
    s.set_timesteps(25)
    x = torch.ones(1)
    out["scalar_kat"] = {"scaled": float(s.scale_model_input(x, s.timesteps[0])),
                         "prev": float(s.step(torch.full((1,), 0.5), s.timesteps[0], x).prev_sample)}
    with open(os.path.join(HERE, "scheduler_kat.json"), "w") as f:
        json.dump(out, f)
    print("scheduler: sigmas[25] head", out["cases"][2]["sigmas"][:3], "scalar", out["scalar_kat"])


CHURN = dict(s_churn=20.0, s_tmin=0.05, s_tmax=200.0, s_noise=1.003)


def gen_scheduler_churn(sched_mod):
    """the stochastic step (s_churn > 0, scheduling_euler_discrete_karras_fix.py:485-497) of the reference scheduler over a
    whole 25-step schedule of the SVD scheduler config (v_prediction), fp16 model outputs as on the GPU path; sigma > s_tmax
    on the first steps (gamma = 0 there), gamma = sqrt(2) - 1 afterwards"""
    from oracle.scheduler import SchedulerConfig
    out = {"churn": CHURN, "cases": []}
    for pt in ("v_prediction",):
        cfg = dict(SchedulerConfig().__dict__, prediction_type=pt)
        s = sched_mod.EulerDiscreteScheduler(**cfg)
        s.set_timesteps(25)
        g = torch.Generator().manual_seed(300)
        gn = torch.Generator().manual_seed(301)          # the generator handed to step()
        x = torch.randn(1, 2, 4, 3, 3, generator=g) * float(s.init_noise_sigma)
        case = {"prediction_type": pt, "seed": 300, "noise_seed": 301, "x0": x.flatten().tolist(), "steps": []}
        for t in s.timesteps:
            v = torch.randn(x.shape, generator=g).half()
            s.scale_model_input(x, t)
            x = s.step(v, t, x, generator=gn, **CHURN).prev_sample
            case["steps"].append({"v": v.float().flatten().tolist(), "prev": x.float().flatten().tolist()})
        out["cases"].append(case)
    with open(os.path.join(HERE, "scheduler_churn_kat.json"), "w") as f:
        json.dump(out, f)
    print("scheduler churn: last prev head", out["cases"][0]["steps"][-1]["prev"][:3])


# ------------------------------------------------------------------------------------------------ 2. UNet wiring
TINY = ou.TINY_CONFIG
WSEED = 7


def tiny_inputs(seed=11, frames=4, hw=8):
    g = torch.Generator().manual_seed(seed)
    return dict(
        sample=torch.randn(2, frames, 8, hw, hw, generator=g),
        enc=torch.randn(2, 1, 1024, generator=g),
        ids=torch.tensor([[6.0, 127.0, 0.02]] * 2),
        t=torch.tensor(1.6378),
        domain=torch.randn(1, 1, 1000, generator=g),
        flow=torch.randn(1, 1, 1000, generator=g),
    )


def skip_shapes(cfg, frames, hw):
    boc = cfg.block_out_channels
    n = 2 * frames
    shapes = [(n, boc[0], hw, hw)]
    r = hw
    for i, c in enumerate(boc):
        shapes += [(n, c, r, r)] * cfg.layers_per_block
        if i != len(boc) - 1:
            r //= 2
            shapes.append((n, c, r, r))
    return shapes, (n, boc[-1], r, r)


def gen_unet(ref_stock, ref_lk):
    kw = {k: v for k, v in TINY.__dict__.items()}
    out = {}
    inp = tiny_inputs()
    with torch.no_grad():
        m = ref_stock.UNetSpatioTemporalConditionControlNetModel(**kw)
        ou.init_weights_(m, WSEED)
        o = ou.UNetSpatioTemporalConditionControlNetModel(TINY)
        o.load_state_dict(m.state_dict())          # name-for-name identical parameter tree
        out["stock_checksum"] = torch.tensor(checksum(m), dtype=torch.float64)
        out["stock_out"] = m(inp["sample"], inp["t"], inp["enc"], added_time_ids=inp["ids"], return_dict=False)[0]
        # with ControlNet residuals (repeated-add quirk, App. C1)
        g = torch.Generator().manual_seed(12)
        shapes, mid_shape = skip_shapes(TINY, 4, 8)
        down = tuple(0.1 * torch.randn(s, generator=g) for s in shapes)
        mid = 0.1 * torch.randn(mid_shape, generator=g)
        out["stock_out_ctrl"] = m(inp["sample"], inp["t"], inp["enc"], down_block_additional_residuals=down,
                                  mid_block_additional_residual=mid, added_time_ids=inp["ids"],
                                  return_dict=False)[0]
        # python-float / int timesteps take the promotion branch (:390-404)
        out["stock_out_tfloat"] = m(inp["sample"], 0.5, inp["enc"], added_time_ids=inp["ids"], return_dict=False)[0]

        lk = ref_lk.UNetSpatioTemporalConditionModel(**kw)
        ou.init_weights_(lk, WSEED + 1)
        out["lk_checksum"] = torch.tensor(checksum(lk), dtype=torch.float64)
        out["lk_out"] = lk(inp["sample"], inp["t"], inp["enc"], inp["domain"], inp["flow"],
                           added_time_ids=inp["ids"], return_dict=False)[0]
        # the fused embedding itself: capture what the first down block receives
        cap = {}
        def _cap(mod, a, k):
            cap.setdefault("enc", k["encoder_hidden_states"])
            return None
        h = lk.down_blocks[0].register_forward_pre_hook(_cap, with_kwargs=True)
        lk(inp["sample"], inp["t"], inp["enc"], inp["domain"], inp["flow"], added_time_ids=inp["ids"])
        h.remove()
        out["lk_fused_enc"] = cap["enc"][::4].contiguous()   # repeat_interleave(4) undone -> [2,1,1024]
    for k, v in inp.items():
        out["in_" + k] = v
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_wiring.safetensors"))
    print("unet wiring: stock std %.4f ctrl std %.4f lk std %.4f" % (
        out["stock_out"].std(), out["stock_out_ctrl"].std(), out["lk_out"].std()))


C1_SEED = 31


def c1_inputs():
    """BASELINE.json configs[0] geometry: CFG batch 2 x 4 frames x 32x32 latent, the real SVD channel widths"""
    g = torch.Generator().manual_seed(C1_SEED + 1)
    return {"sample": torch.randn(2, 4, 8, 32, 32, generator=g), "t": torch.tensor(1.25),
            "enc": torch.randn(2, 1, 1024, generator=g), "ids": torch.tensor([[6.0, 127.0, 0.02]] * 2)}


def gen_unet_c1(ref_stock):
    """ONE forward of the reference UNet at its real width (block_out_channels (320,640,1280,1280), heads (5,10,20,20),
    1.52 B parameters, fp32 on the CPU) on the configs[0] geometry.  Weights: oracle.init_weights_(seed C1_SEED) rounded to
    fp16 (what the HIP model holds), regenerated from the seed by the test; only inputs, output and a checksum are stored."""
    from oracle.unet import SVD_CONFIG
    inp = c1_inputs()
    with torch.no_grad():
        m = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(m, C1_SEED)
        for p in m.parameters():
            p.copy_(p.half().float())
        out = {"checksum": torch.tensor(checksum(m), dtype=torch.float64),
               "out": m(inp["sample"], inp["t"], inp["enc"], added_time_ids=inp["ids"], return_dict=False)[0]}
    for k, v in inp.items():
        out["in_" + k] = v
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_c1_realwidth.safetensors"))
    print("unet c1 real width: out std %.4f, checksum %.6e" % (out["out"].std(), out["checksum"]))


def gen_controlnet(ref_ctrl, ref_stock):
    """ControlNet-SVD encoder: the reference class over the restated blocks, tiny config"""
    from oracle import controlnet as oc
    kw = {k: v for k, v in TINY.__dict__.items()}
    inp = tiny_inputs()
    g = torch.Generator().manual_seed(71)
    cond = torch.rand(2, 4, 3, 64, 64, generator=g)          # pixel-space conditioning: 8x the 8x8 latent grid
    out = {"in_cond": cond}
    with torch.no_grad():
        m = ref_ctrl.ControlNetSDVModel(**kw, conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256))
        ou.init_weights_(m, WSEED + 5)                       # also fills the zero-initialised convolutions
        o = oc.ControlNetSDVModel(TINY)
        o.load_state_dict(m.state_dict())                    # name-for-name identical parameter tree
        out["checksum"] = torch.tensor(checksum(m), dtype=torch.float64)
        down, mid = m(inp["sample"], inp["t"], inp["enc"], inp["ids"], controlnet_cond=cond, return_dict=False,
                      conditioning_scale=0.75)
        for i, d in enumerate(down):
            out[f"down_{i}"] = d
        out["mid"] = mid
        out["cond_embedding"] = m.controlnet_cond_embedding(cond)
        down0, mid0 = m(inp["sample"], inp["t"], inp["enc"], inp["ids"], controlnet_cond=None, return_dict=False)
        out["nocond_down_3"], out["nocond_mid"] = down0[3], mid0
        # the residuals through the reference's UNet (pipeline_..._controlnet.py:582-607)
        u = ref_stock.UNetSpatioTemporalConditionControlNetModel(**kw)
        ou.init_weights_(u, WSEED)
        out["unet_out"] = u(inp["sample"], inp["t"], inp["enc"], down_block_additional_residuals=down,
                            mid_block_additional_residual=mid, added_time_ids=inp["ids"], return_dict=False)[0]
    for k, v in inp.items():
        out["in_" + k] = v
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "controlnet.safetensors"))
    print("controlnet: %d down residuals, mid std %.4f, cond-embedding std %.4f, unet std %.4f" % (
        len(down), mid.std(), out["cond_embedding"].std(), out["unet_out"].std()))


# ------------------------------------------------------------------------------------------------ 3. patch hooks
class _Holder(nn.Module):
    def __init__(self):
        super().__init__()
        self.spatial = ob.BasicTransformerBlock(128, 2, 64, 1024)
        self.temporal = ob.TemporalBasicTransformerBlock(128, 128, 2, 64, 1024)

    def forward(self, x):
        return x


def gen_patch(patch_mod):
    torch.manual_seed(21)
    holder = _Holder()
    ou.init_weights_(holder, 21)
    # `apply_patch` wants a ModelMixin-named class in the MRO
    holder.__class__ = type("Holder", (_Holder, sys.modules["diffusers.models.modeling_utils"].ModelMixin), {})
    out = {}
    frames, S, C = 3, 16, 128
    g = torch.Generator().manual_seed(22)
    x = torch.randn(4 * frames, S, C, generator=g)          # batch [start,end] x CFG, masks [0,1,0,1]
    enc = torch.randn(4 * frames, 1, 1024, generator=g)
    tctx = torch.randn(4 * S, 1, 1024, generator=g)
    mask = [False, True, False, True]
    with torch.no_grad():
        for flip in (False, True):
            patch_mod.apply_patch(holder, flip=flip, with_spatial_block=True, with_temporal_block=True)
            patch_mod.initialize_joint_layers(holder, post="conv")
            patch_mod.set_joint_attention_mask(holder, mask)
            # zero-init conv1n => identity; give it weights so the joint branch is exercised
            gg = torch.Generator().manual_seed(23)
            for blk in (holder.spatial, holder.temporal):
                blk.conv1n.weight.copy_(torch.randn(C, C, generator=gg) / C ** 0.5)
                for p in blk.attn1n.parameters():
                    p.add_(0.05 * torch.randn(p.shape, generator=gg))
            holder._tome_info["size"] = (4, frames, C, 4, 4)
            tag = "flip" if flip else "noflip"
            patch_mod.set_joint_attention(holder, True)
            out[f"spatial_joint_{tag}"] = holder.spatial(x, encoder_hidden_states=enc)
            if not flip:
                out["temporal_joint"] = holder.temporal(x, num_frames=frames, encoder_hidden_states=tctx)
                patch_mod.set_joint_scale(holder, 0.5)
                out["spatial_joint_scale05"] = holder.spatial(x, encoder_hidden_states=enc)
                patch_mod.set_joint_scale(holder, 1.0)
                patch_mod.set_joint_attention(holder, False)
                out["spatial_nojoint"] = holder.spatial(x, encoder_hidden_states=enc)
                out["temporal_nojoint"] = holder.temporal(x, num_frames=frames, encoder_hidden_states=tctx)
                out["conv1n_spatial"] = holder.spatial.conv1n.weight.clone()
                out["conv1n_temporal"] = holder.temporal.conv1n.weight.clone()
                for n, p in holder.spatial.attn1n.named_parameters():
                    out["attn1n_spatial." + n] = p.clone()
                for n, p in holder.temporal.attn1n.named_parameters():
                    out["attn1n_temporal." + n] = p.clone()
            patch_mod.remove_patch(holder)
        # the other post-processing variants of initialize_joint_layers (patch.py:146-158, :484-494)
        for post in ("scale", "conv_fuse"):
            patch_mod.apply_patch(holder, flip=False, with_spatial_block=True, with_temporal_block=True)
            patch_mod.initialize_joint_layers(holder, post=post)
            patch_mod.set_joint_attention_mask(holder, mask)
            gg = torch.Generator().manual_seed(31 if post == "scale" else 32)
            for name, blk in (("spatial", holder.spatial), ("temporal", holder.temporal)):
                if post == "scale":
                    blk.scale1n.copy_(torch.randn(1, 1, C, generator=gg))
                    out[f"{post}.scale1n_{name}"] = blk.scale1n.detach().clone()
                else:
                    blk.conv1n.weight.copy_(torch.randn(2 * C, 2 * C, generator=gg) / (2 * C) ** 0.5)
                    out[f"{post}.conv1n_{name}"] = blk.conv1n.weight.clone()
                for n, p in blk.attn1n.named_parameters():
                    p.add_(0.05 * torch.randn(p.shape, generator=gg))
                    out[f"{post}.attn1n_{name}." + n] = p.detach().clone()
            holder._tome_info["size"] = (4, frames, C, 4, 4)
            patch_mod.set_joint_attention(holder, True)
            patch_mod.set_joint_scale(holder, 0.75)
            out[f"{post}.spatial_joint"] = holder.spatial(x, encoder_hidden_states=enc)
            out[f"{post}.temporal_joint"] = holder.temporal(x, num_frames=frames, encoder_hidden_states=tctx)
            patch_mod.remove_patch(holder)
    out["in_x"], out["in_enc"], out["in_tctx"] = x, enc, tctx
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "patch_joint.safetensors"))
    print("patch: joint vs nojoint delta %.4f" % (out["spatial_joint_noflip"] - out["spatial_nojoint"]).abs().max())


def gen_patch_fsm(fsm_mod):
    torch.manual_seed(51)
    holder = _Holder()
    ou.init_weights_(holder, 51)
    holder.__class__ = type("Holder", (_Holder, sys.modules["diffusers.models.modeling_utils"].ModelMixin), {})
    C, fh, fw = 128, 6, 8
    S = fh * fw
    nb, P = 6, 40                      # 3 (src, dst) pairs, 40 tracked points each
    g = torch.Generator().manual_seed(52)
    x = torch.randn(nb, S, C, generator=g)
    enc = torch.randn(nb, 1, 1024, generator=g)
    track_res = (2 * fh, 2 * fw)       # tracks live on a 2x finer grid -> downsample 2
    src = torch.stack([torch.randint(0, 2 * fw, (nb // 2, P), generator=g),
                       torch.randint(0, 2 * fh, (nb // 2, P), generator=g)], dim=-1).float()
    dst = torch.stack([torch.randint(-3, 2 * fw + 3, (nb // 2, P), generator=g),      # dst may leave the image: clamped
                       torch.randint(-3, 2 * fh + 3, (nb // 2, P), generator=g)], dim=-1).float()
    vis = (torch.rand(nb // 2, P, generator=g) > 0.25).float()
    out = {"in_x": x, "in_enc": enc, "src_tracks": src, "dst_tracks": dst, "vis": vis,
           "track_res": torch.tensor(track_res)}
    with torch.no_grad():
        fsm_mod.apply_patch(holder, with_spatial_block=True, with_temporal_block=False)
        fsm_mod.initialize_joint_layers(holder)
        fsm_mod.update_patch(holder, track=(src.clone(), dst.clone(), vis.clone()), track_res=track_res)
        fsm_mod.set_joint_attention(holder, True)
        out["fsm_zero_init"] = holder.spatial(x, encoder_hidden_states=enc)          # zero conv_fuse => identity
        gg = torch.Generator().manual_seed(53)
        holder.spatial.conv_fuse.weight.copy_(torch.randn(2 * C, 2 * C, 3, 3, generator=gg) / (18 * C) ** 0.5)
        holder.spatial.conv_fuse.bias.copy_(0.1 * torch.randn(2 * C, generator=gg))
        out["conv_fuse_w"] = holder.spatial.conv_fuse.weight.clone()
        out["conv_fuse_b"] = holder.spatial.conv_fuse.bias.clone()
        fsm_mod.update_patch(holder, track=(src.clone(), dst.clone(), vis.clone()), track_res=track_res)
        out["fsm_on"] = holder.spatial(x, encoder_hidden_states=enc)
        fsm_mod.set_joint_attention(holder, False)
        out["fsm_off"] = holder.spatial(x, encoder_hidden_states=enc)
        fsm_mod.remove_patch(holder)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "patch_fsm.safetensors"))
    print("patch_FSM: on vs off delta %.4f, zero-init delta %.2e" % (
        (out["fsm_on"] - out["fsm_off"]).abs().max(), (out["fsm_zero_init"] - out["fsm_off"]).abs().max()))


# ------------------------------------------------------------------------------------------------ 4. pipeline loop
class _FakeVAE(nn.Module):
    """Boundary stand-in: 8x average pool + fixed channel mix -> 4 latent channels."""
    def __init__(self):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=(1, 1, 1, 1), force_upcast=False, scaling_factor=0.18215)
        g = torch.Generator().manual_seed(31)
        self.mix = nn.Parameter(torch.randn(4, 3, generator=g))

    @property
    def dtype(self):
        return self.mix.dtype

    def encode(self, image):
        z = torch.nn.functional.avg_pool2d(image, 8)
        z = torch.einsum("oc,bchw->bohw", self.mix, z) * 0.18215
        return SimpleNamespace(latent_dist=SimpleNamespace(mode=lambda: z))


class _FakeCLIP(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(32)
        self.w = nn.Parameter(torch.randn(1024, 3, generator=g))

    def forward(self, image):
        return SimpleNamespace(image_embeds=torch.einsum("oc,bc->bo", self.w, image.mean(dim=(2, 3))))


def gen_loop(pipe_mod, ref_stock, sched_mod):
    from oracle.scheduler import SchedulerConfig
    kw = {k: v for k, v in TINY.__dict__.items()}
    unet = ref_stock.UNetSpatioTemporalConditionControlNetModel(**kw)
    ou.init_weights_(unet, WSEED)
    sched = sched_mod.EulerDiscreteScheduler(**SchedulerConfig().__dict__)
    fe = lambda images, **k: SimpleNamespace(pixel_values=images)   # noqa: E731
    pipe = pipe_mod.StableVideoDiffusionPipeline(vae=_FakeVAE(), image_encoder=_FakeCLIP(), unet=unet,
                                                 scheduler=sched, feature_extractor=fe)
    g = torch.Generator().manual_seed(41)
    image = torch.rand(1, 3, 64, 64, generator=g)
    lat0 = torch.randn(1, 4, 4, 8, 8, generator=g)
    rec = {}
    orig_forward = unet.forward

    def spy(sample, t, **k):
        y = orig_forward(sample, t, **k)
        if "in0" not in rec:
            rec["in0"], rec["out0"] = sample.clone(), y[0].clone()
            rec["enc"], rec["ids"] = k["encoder_hidden_states"].clone(), k["added_time_ids"].clone()
        return y
    unet.forward = spy
    steps = []
    res = pipe(image, height=64, width=64, num_frames=4, num_inference_steps=3, latents=lat0.clone(),
               output_type="latent", generator=torch.Generator().manual_seed(42),
               callback_on_step_end=lambda p, i, t, kw_: (steps.append(kw_["latents"].clone()), {})[1])
    out = {"latents0": lat0, "final": res.frames, "unet_in0": rec["in0"], "unet_out0": rec["out0"],
           "image_embeddings": rec["enc"], "added_time_ids": rec["ids"],
           "image_latents": rec["in0"][:, :, 4:].contiguous(),
           "step_latents": torch.stack(steps)}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "loop.safetensors"))
    print("loop: final std %.4f (3 steps), ids %s" % (out["final"].std(), rec["ids"].tolist()))


from image_cases import IMAGE_CASES, image_case_input   # noqa: E402  (shared with tests/test_image_ops*.py)


def gen_image_ops(pipe_mod):
    """`_resize_with_antialiasing` (reference pipeline_stable_video_diffusion_trans.py:661-765) on seeded images; inputs
    are regenerated from the seed by the tests (image_case_input), the full-size case stores every 4th output pixel"""
    out = {}
    for i, (name, shape, size, sub) in enumerate(IMAGE_CASES):
        y = pipe_mod._resize_with_antialiasing(image_case_input(shape, 900 + i), size)
        out[name] = y[..., ::sub, ::sub].contiguous()
        out[name + "_mean"] = y.mean().reshape(1)
    save_file(out, os.path.join(HERE, "image_ops.safetensors"))
    print("image_ops:", {k: tuple(v.shape) for k, v in out.items() if not k.endswith("_mean")})


def gen_loop_c1(pipe_mod, ref_stock, sched_mod):
    """BASELINE.json configs[0] end to end through the reference's own pipeline `__call__`: real-width UNet (weights as in
    gen_unet_c1), 1 clip x 4 frames x 256x256 pixels (32x32 latent), 2 Euler steps, CFG 1 -> 3, output_type="latent"."""
    from oracle.scheduler import SchedulerConfig
    from oracle.unet import SVD_CONFIG
    with torch.no_grad():
        unet = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(unet, C1_SEED)
        for p in unet.parameters():
            p.copy_(p.half().float())
    sched = sched_mod.EulerDiscreteScheduler(**SchedulerConfig().__dict__)
    fe = lambda images, **k: SimpleNamespace(pixel_values=images)   # noqa: E731
    pipe = pipe_mod.StableVideoDiffusionPipeline(vae=_FakeVAE(), image_encoder=_FakeCLIP(), unet=unet,
                                                 scheduler=sched, feature_extractor=fe)
    g = torch.Generator().manual_seed(C1_SEED + 2)
    image = torch.rand(1, 3, 256, 256, generator=g)
    lat0 = torch.randn(1, 4, 4, 32, 32, generator=g)
    rec, steps = {}, []
    orig_forward = unet.forward

    def spy(sample, t, **k):
        if "enc" not in rec:
            rec["enc"], rec["ids"] = k["encoder_hidden_states"].clone(), k["added_time_ids"].clone()
            rec["image_latents"] = sample[:, :, 4:].clone()
        return orig_forward(sample, t, **k)
    unet.forward = spy
    res = pipe(image, height=256, width=256, num_frames=4, num_inference_steps=2, latents=lat0.clone(),
               output_type="latent", generator=torch.Generator().manual_seed(C1_SEED + 3),
               callback_on_step_end=lambda p, i, t, kw_: (steps.append(kw_["latents"].clone()), {})[1])
    out = {"latents0": lat0, "final": res.frames, "image_embeddings": rec["enc"], "added_time_ids": rec["ids"],
           "image_latents": rec["image_latents"].contiguous(), "step_latents": torch.stack(steps),
           "checksum": torch.tensor(checksum(unet), dtype=torch.float64)}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "loop_c1_realwidth.safetensors"))
    print("loop c1 real width: final std %.4f (2 steps)" % out["final"].std())


from fullres_cases import (F14_SEED, FULLRES_LK_SEED, FULLRES_SEED, LOOP25_SEED, fullres_inputs, fullres_tracks,   # noqa: E402
                           seed_conv_fuse_)


def gen_patch_lora(patch_mod, ref_stock):
    """the inference loader's LoRA sequence (utils/util.py:560-606) on the reference's own classes: `patch.apply_patch` +
    `initialize_joint_layers` on the stock UNet, adapters xy_lora / yx_lora as vendored `models/lora_layer.py::Linear`
    wrappers on every attention projection, `set_adapter`, `patch.hack_lora_forward` (patch.py:57-92),
    `set_patch_lora_mask` ([0,1,0,1] / [1,0,1,0], inverted on attn1n.to_k / to_v :889-892), `set_joint_attention_mask`;
    one forward of the batch [u_x, u_y, c_x, c_y] per variant"""
    from lora_cases import ADAPTERS, ALPHA, JOINT_MASK, MASKS, RANK, lora_inputs, seed_joint_and_lora_
    LoraLinear = sys.modules["peft.tuners.lora.layer"].Linear
    kw = {k: v for k, v in TINY.__dict__.items()}
    inp = lora_inputs()
    out = {}
    with torch.no_grad():
        m = ref_stock.UNetSpatioTemporalConditionControlNetModel(**kw)
        ou.init_weights_(m, WSEED + 9)
        for p in m.parameters():
            p.copy_(p.half().float())
        out["checksum_base"] = torch.tensor(checksum(m), dtype=torch.float64)
        patch_mod.apply_patch(m, flip=False, with_temporal_block=True, with_spatial_block=True)
        patch_mod.initialize_joint_layers(m)
        for name, mod in list(m.named_modules()):          # peft inject_adapter: suffix match on the target names
            if isinstance(mod, nn.Linear) and any(name.endswith("." + t) for t in ("to_q", "to_k", "to_v", "to_out.0")) \
                    and ".lora_" not in name and not name.endswith("base_layer"):
                parent = m.get_submodule(name.rpartition(".")[0])
                child = name.rpartition(".")[2]
                w = LoraLinear(mod, ADAPTERS[0], r=RANK, lora_alpha=ALPHA)
                for a in ADAPTERS[1:]:
                    w.update_layer(a, RANK, lora_alpha=ALPHA, lora_dropout=0.0, init_lora_weights=True, use_rslora=False)
                if isinstance(parent, nn.ModuleList):
                    parent[int(child)] = w
                else:
                    setattr(parent, child, w)
        names = seed_joint_and_lora_(m)
        out["n_seeded"] = torch.tensor(len(names))
        out["checksum"] = torch.tensor(checksum(m), dtype=torch.float64)
        for mod in m.modules():
            if isinstance(mod, LoraLinear):
                mod.set_adapter(list(ADAPTERS))
        call = lambda: m(inp["sample"], inp["t"], inp["enc"], added_time_ids=inp["ids"], return_dict=False)[0]  # noqa: E731
        patch_mod.set_joint_attention_mask(m, JOINT_MASK)
        patch_mod.set_joint_attention(m, True)
        out["plain_peft"] = call()                          # unhacked peft forward: both adapters on every entry
        patch_mod.hack_lora_forward(m)
        for a, mk in MASKS.items():
            patch_mod.set_patch_lora_mask(m, a, mk)
        out["masked"] = call()
        patch_mod.set_joint_attention(m, False)
        out["masked_nojoint"] = call()
        patch_mod.set_joint_attention(m, True)
        for mod in m.modules():                             # single_lora branch of the loader (:598-599)
            if isinstance(mod, LoraLinear):
                mod.set_adapter(["xy_lora"])
        patch_mod.set_patch_lora_mask(m, "xy_lora", [1, 1, 1, 1])
        out["single_all_ones"] = call()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "patch_lora.safetensors"))
    print("patch_lora: %d seeded tensors; masked vs plain delta %.4f, masked vs nojoint %.4f, single %.4f" % (
        len(names), (out["masked"] - out["plain_peft"]).abs().max(), (out["masked"] - out["masked_nojoint"]).abs().max(),
        (out["single_all_ones"] - out["masked"]).abs().max()))


DIT_SEED = 191


def dit_inputs(cfg, seed=DIT_SEED + 1, batch=2):
    g = torch.Generator().manual_seed(seed)
    f = (cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1
    return dict(hidden=torch.randn(batch, f, cfg.in_channels, cfg.sample_height, cfg.sample_width, generator=g).half().float(),
                text=torch.randn(batch, cfg.max_text_seq_length, cfg.text_embed_dim, generator=g).half().float(),
                t=torch.tensor([721] * batch), domain=torch.randn(1, 1, 1000, generator=g),
                flow=torch.randn(1, 1, 1000, generator=g))


def gen_cogvideox():
    """the reference's in-tree CogVideoX DiT (CogVideo-main/finetune/models/cogvideox_i2v/cogvideox_transformer_3d.py: block
    :41-160, forward :473-638 incl. the latent-knowledge fuse on the text embeddings :519-582 and the un-patchify) executed over
    the restated diffusers >= 0.32 pieces of oracle/cogvideox.py (bound by name: Attention, FeedForward, CogVideoXPatchEmbed,
    CogVideoXLayerNormZero, AdaLayerNorm, the attention processor), tiny config"""
    from oracle import cogvideox as oc
    du = sys.modules["diffusers.utils"]
    du.USE_PEFT_BACKEND = False
    du.scale_lora_layers = lambda *a, **k: None
    du.unscale_lora_layers = lambda *a, **k: None
    sys.modules["diffusers.utils.torch_utils"].maybe_allow_in_graph = lambda c: c
    at = _mod("diffusers.models.attention")
    at.Attention, at.FeedForward = oc.Attention, oc.FeedForward
    ap = sys.modules["diffusers.models.attention_processor"]
    ap.CogVideoXAttnProcessor2_0, ap.FusedCogVideoXAttnProcessor2_0 = oc.CogVideoXAttnProcessor2_0, oc.CogVideoXAttnProcessor2_0
    _mod("diffusers.models.cache_utils").CacheMixin = type("CacheMixin", (), {})
    sys.modules["diffusers.models.embeddings"].CogVideoXPatchEmbed = oc.CogVideoXPatchEmbed

    class _TE(ob.TimestepEmbedding):          # diffusers' signature (in_channels, time_embed_dim, act_fn, out_dim) / forward(x, cond)
        def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
            assert act_fn == "silu"
            super().__init__(in_channels, time_embed_dim, out_dim)

        def forward(self, sample, condition=None):
            assert condition is None
            return super().forward(sample)
    sys.modules["diffusers.models.embeddings"].TimestepEmbedding = _TE
    _mod("diffusers.models.modeling_outputs").Transformer2DModelOutput = type("Transformer2DModelOutput", (), {})
    nm = sys.modules["diffusers.models.normalization"]
    nm.AdaLayerNorm, nm.CogVideoXLayerNormZero = oc.AdaLayerNorm, oc.CogVideoXLayerNormZero
    ref = load_ref("CogVideo-main/finetune/models/cogvideox_i2v/cogvideox_transformer_3d.py", "ref_cogvideox_transformer_3d")
    cfg = oc.TINY_DIT
    kw = {k: v for k, v in cfg.__dict__.items()}
    with torch.no_grad():
        m = ref.CogVideoXTransformer3DModel(**kw)
        m.init_quaternion_modules()
        o = oc.CogVideoXTransformer3DModel(cfg)
        assert sorted(k for k, _ in m.named_parameters()) == sorted(k for k, _ in o.named_parameters())
        oc.init_weights_(m, DIT_SEED)
        for p in m.parameters():
            p.copy_(p.half().float())
        inp = dit_inputs(cfg)
        out = {"checksum": torch.tensor(checksum(m), dtype=torch.float64)}
        out["out"] = m(inp["hidden"], inp["text"], inp["t"], inp["domain"], inp["flow"], return_dict=False)[0]
        cap = {}
        h = m.patch_embed.register_forward_pre_hook(lambda mod, a: (cap.setdefault("text", a[0].clone()), None)[1])
        m(inp["hidden"], inp["text"], inp["t"], inp["domain"], inp["flow"], return_dict=False)
        h.remove()
        out["fused_text"] = cap["text"]
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "cogvideox.safetensors"))
    print("cogvideox: out %s std %.4f, fused text std %.4f" % (tuple(out["out"].shape), out["out"].std(), out["fused_text"].std()))


def _run_pipeline_loop(pipe_mod, unet, sched_mod, image, lat0, px, frames, steps_n, gen_seed):
    """the reference `__call__` (output_type="latent") with stand-in CLIP / VAE; returns what the loop tests need"""
    from oracle.scheduler import SchedulerConfig
    sched = sched_mod.EulerDiscreteScheduler(**SchedulerConfig().__dict__)
    fe = lambda images, **k: SimpleNamespace(pixel_values=images)   # noqa: E731
    pipe = pipe_mod.StableVideoDiffusionPipeline(vae=_FakeVAE(), image_encoder=_FakeCLIP(), unet=unet,
                                                 scheduler=sched, feature_extractor=fe)
    rec, steps = {}, []
    orig_forward = unet.forward
    lc = lat0.shape[2]

    def spy(sample, t, **k):
        if "enc" not in rec:
            rec["enc"], rec["ids"] = k["encoder_hidden_states"].clone(), k["added_time_ids"].clone()
            rec["image_latents"] = sample[:, :, lc:].clone()
        y = orig_forward(sample, t, **k)
        print("  step", len(steps), flush=True)
        return y
    unet.forward = spy
    res = pipe(image, height=px, width=px, num_frames=frames, num_inference_steps=steps_n, latents=lat0.clone(),
               output_type="latent", generator=torch.Generator().manual_seed(gen_seed),
               callback_on_step_end=lambda p, i, t, kw_: (steps.append(kw_["latents"].clone()), {})[1])
    unet.forward = orig_forward
    return {"latents0": lat0, "final": res.frames, "image_embeddings": rec["enc"], "added_time_ids": rec["ids"],
            "image_latents": rec["image_latents"].contiguous(), "step_latents": torch.stack(steps)}


def gen_loop25(pipe_mod, ref_stock, sched_mod):
    """the 25-step Euler loop of the headline metric at the tiny width: reference `__call__`
    (pipeline_stable_video_diffusion_trans.py:544-640, scheduler :418-528), every step's latents stored"""
    kw = {k: v for k, v in TINY.__dict__.items()}
    with torch.no_grad():
        unet = ref_stock.UNetSpatioTemporalConditionControlNetModel(**kw)
        ou.init_weights_(unet, LOOP25_SEED)
        for p in unet.parameters():
            p.copy_(p.half().float())
    g = torch.Generator().manual_seed(LOOP25_SEED + 1)
    image = torch.rand(1, 3, 64, 64, generator=g)
    lat0 = torch.randn(1, 4, 4, 8, 8, generator=g)
    out = _run_pipeline_loop(pipe_mod, unet, sched_mod, image, lat0, 64, 4, 25, LOOP25_SEED + 2)
    out["checksum"] = torch.tensor(checksum(unet), dtype=torch.float64)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "loop25.safetensors"))
    print("loop25 (tiny): final std %.4f, step stds %s" % (
        out["final"].std(), [round(float(s.std()), 3) for s in out["step_latents"][::6]]))


def gen_loop25_c1(pipe_mod, ref_stock, sched_mod):
    """the 25-step loop with the REAL-width UNet on the configs[0] geometry (1 clip x 4 frames x 256x256 px): 25 fp32
    CPU forwards of the reference (weights as gen_unet_c1), every step's latents stored"""
    from oracle.unet import SVD_CONFIG
    with torch.no_grad():
        unet = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(unet, C1_SEED)
        for p in unet.parameters():
            p.copy_(p.half().float())
    g = torch.Generator().manual_seed(C1_SEED + 2)
    image = torch.rand(1, 3, 256, 256, generator=g)
    lat0 = torch.randn(1, 4, 4, 32, 32, generator=g)
    out = _run_pipeline_loop(pipe_mod, unet, sched_mod, image, lat0, 256, 4, 25, C1_SEED + 3)
    out["checksum"] = torch.tensor(checksum(unet), dtype=torch.float64)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "loop25_c1_realwidth.safetensors"))
    print("loop25 c1 real width: final std %.4f" % out["final"].std())


def gen_loop_f14(pipe_mod, ref_stock, sched_mod, guidance: bool):
    """round 3: the reference `__call__` with the REAL-width UNet on ALL 14 frames of a clip at a 32 x 64 latent
    (256 x 512 px; S = 2048 - every UNet level needs latent sides divisible by 8): 2 Euler steps, every step's latents stored - the 14-frame temporal attention, the F = 14
    Conv3d chain and the temporal GroupNorm across 14 frames, end to end, and the oracle the (7,7) / (4,4,3,3) frame-sharded
    runs are checked against.  guidance: CFG 1 -> 3 (batch 2 x 14) or guidance off (max_guidance_scale 1: batch 1 x 14, the
    layout a 4-way frame split without CFG-parallel runs).  fp32 on the CPU: ~20 TFLOP per forward of the CFG batch."""
    from oracle.unet import SVD_CONFIG
    with torch.no_grad():
        unet = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(unet, C1_SEED)
        for p in unet.parameters():
            p.copy_(p.half().float())
    g = torch.Generator().manual_seed(F14_SEED)
    image = torch.rand(1, 3, 256, 512, generator=g)
    lat0 = torch.randn(1, 14, 4, 32, 64, generator=g)
    from oracle.scheduler import SchedulerConfig
    sched = sched_mod.EulerDiscreteScheduler(**SchedulerConfig().__dict__)
    fe = lambda images, **k: SimpleNamespace(pixel_values=images)   # noqa: E731
    pipe = pipe_mod.StableVideoDiffusionPipeline(vae=_FakeVAE(), image_encoder=_FakeCLIP(), unet=unet, scheduler=sched,
                                                 feature_extractor=fe)
    rec, steps = {}, []
    orig_forward = unet.forward

    def spy(sample, t, **k):
        if "enc" not in rec:
            rec["enc"], rec["ids"] = k["encoder_hidden_states"].clone(), k["added_time_ids"].clone()
            rec["image_latents"] = sample[:, :, 4:].clone()
        y = orig_forward(sample, t, **k)
        print("  step", len(steps), flush=True)
        return y
    unet.forward = spy
    res = pipe(image, height=256, width=512, num_frames=14, num_inference_steps=2, latents=lat0.clone(),
               min_guidance_scale=1.0, max_guidance_scale=3.0 if guidance else 1.0,
               output_type="latent", generator=torch.Generator().manual_seed(F14_SEED + 1),
               callback_on_step_end=lambda p_, i, t, kw_: (steps.append(kw_["latents"].clone()), {})[1])
    out = {"latents0": lat0, "final": res.frames, "image_embeddings": rec["enc"], "added_time_ids": rec["ids"],
           "image_latents": rec["image_latents"].contiguous(), "step_latents": torch.stack(steps),
           "checksum": torch.tensor(checksum(unet), dtype=torch.float64)}
    name = "loop_f14_cfg.safetensors" if guidance else "loop_f14_nocfg.safetensors"
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, name))
    print("loop f14 (%s): final std %.4f" % ("cfg" if guidance else "no cfg", out["final"].std()))


def gen_loop25_headline(pipe_mod, ref_stock, sched_mod):
    """round 5: the reference `__call__` (pipeline_stable_video_diffusion_trans.py:544-640) through ALL 25 Euler steps at the
    geometry of BASELINE.json configs[1] itself - one clip x 14 frames x 576 x 1024 px (latent 72 x 128), CFG 1 -> 3, real-width
    UNet (weights of the other real-width fixtures) - in fp32 on the CPU: 25 x 89.7 TFLOP, about two hours here.  Stored: the
    latents after steps 5 / 10 / 15 / 20 and the returned latents (= step 25) in fp16 (their rounding, 5e-4 relative, is far inside the
    5e-2 gate), fp64 sum / abs-sum / std of EVERY step's fp32 latents, and the loop's boundary inputs.  Progress is written to
    /tmp after every step so that a killed run leaves its partial curve behind."""
    from fullres_cases import HEADLINE_SEED, headline_inputs
    from oracle.unet import SVD_CONFIG
    with torch.no_grad():
        unet = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(unet, C1_SEED)
        for p in unet.parameters():
            p.copy_(p.half().float())
    image, lat0 = headline_inputs()
    from oracle.scheduler import SchedulerConfig
    sched = sched_mod.EulerDiscreteScheduler(**SchedulerConfig().__dict__)
    fe = lambda images, **k: SimpleNamespace(pixel_values=images)   # noqa: E731
    pipe = pipe_mod.StableVideoDiffusionPipeline(vae=_FakeVAE(), image_encoder=_FakeCLIP(), unet=unet, scheduler=sched,
                                                 feature_extractor=fe)
    rec, kept, stats = {}, {}, []
    orig_forward = unet.forward
    import time
    t00 = time.time()

    def spy(sample, t, **k):
        if "enc" not in rec:
            rec["enc"], rec["ids"] = k["encoder_hidden_states"].clone(), k["added_time_ids"].clone()
            il = sample[:, :, 4:]
            assert bool((il == il[:, :1]).all()), "image latents differ between frames"
            rec["image_latents"] = il[:, 0].clone()
        return orig_forward(sample, t, **k)

    def on_step(p_, i, t, kw_):
        lat = kw_["latents"]
        d = lat.double()
        stats.append([float(d.sum()), float(d.abs().sum()), float(d.std())])
        if (i + 1) % 5 == 0 and i + 1 < 25:      # (step 25 is the returned tensor)
            kept[i] = lat.half().clone()
        print("  step %d done at %.0f s, latents std %.4f" % (i, time.time() - t00, stats[-1][2]), flush=True)
        torch.save({"stats": stats, "kept": kept}, "/tmp/loop25_headline_progress.pt")
        return {}
    unet.forward = spy
    res = pipe(image, height=576, width=1024, num_frames=14, num_inference_steps=25, latents=lat0.clone(),
               min_guidance_scale=1.0, max_guidance_scale=3.0, output_type="latent",
               generator=torch.Generator().manual_seed(HEADLINE_SEED + 1), callback_on_step_end=on_step)
    assert sorted(kept) == [4, 9, 14, 19]
    out = {"final_f16": res.frames.half(), "kept_steps": torch.tensor(sorted(kept)),
           "step_latents_f16": torch.stack([kept[i] for i in sorted(kept)]),
           "step_stats": torch.tensor(stats, dtype=torch.float64),            # [25, 3]: sum, abs-sum, std of the fp32 latents
           "image_embeddings": rec["enc"], "added_time_ids": rec["ids"], "image_latents": rec["image_latents"].contiguous(),
           "checksum": torch.tensor(checksum(unet), dtype=torch.float64),
           "latents0_sum": lat0.double().sum().reshape(1)}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "loop25_headline.safetensors"))
    print("loop25 headline: final std %.4f, %.0f s" % (res.frames.std(), time.time() - t00))


def gen_unet_fullres(ref_stock):
    """ONE forward of the reference's stock UNet (unet_spatio_temporal_condition_controlnet.py:358-508) at the REAL width
    and the FULL latent resolution of configs[1] (72 x 128, S = 9216) with CFG 2 x 2 frames, fp32 on the CPU"""
    from oracle.unet import SVD_CONFIG
    inp = fullres_inputs()
    with torch.no_grad():
        m = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(m, C1_SEED)          # the weights of the other real-width fixtures (one 1.5 B-parameter init per test session)
        for p in m.parameters():
            p.copy_(p.half().float())
        y = m(inp["sample"], inp["t"], inp["enc"], added_time_ids=inp["ids"], return_dict=False)[0]
        out = {"checksum": torch.tensor(checksum(m), dtype=torch.float64), "out": y,
               "out_sum": y.double().sum().reshape(1), "out_abs_sum": y.double().abs().sum().reshape(1)}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_fullres.safetensors"))
    print("unet fullres: out std %.4f, checksum %.6e" % (out["out"].std(), out["checksum"]))


def gen_unet_fullres_f14(ref_stock):
    """ONE forward of the reference's stock UNet at the geometry of BASELINE.json configs[1] ITSELF: CFG 2 x 14 frames x 72 x 128
    latent (N = 28 frame-images, S = 9216; 89.7 TFLOP in fp32 on the CPU, minutes).  The output is stored in fp16 (2 MB; its
    rounding, 5e-4 relative, is far inside the 1e-2 gate) beside fp64 sums of the fp32 result."""
    from fullres_cases import FULLRES_F14_SEED
    from oracle.unet import SVD_CONFIG
    inp = fullres_inputs(seed=FULLRES_F14_SEED, frames=14)
    with torch.no_grad():
        m = ref_stock.UNetSpatioTemporalConditionControlNetModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(m, C1_SEED)
        for p in m.parameters():
            p.copy_(p.half().float())
        import time
        t0 = time.time()
        y = m(inp["sample"], inp["t"], inp["enc"], added_time_ids=inp["ids"], return_dict=False)[0]
        print("reference forward at 2 x 14 x 72 x 128: %.0f s" % (time.time() - t0))
        out = {"checksum": torch.tensor(checksum(m), dtype=torch.float64), "out_f16": y.half(),
               "out_sum": y.double().sum().reshape(1), "out_abs_sum": y.double().abs().sum().reshape(1),
               "out_std": y.double().std().reshape(1)}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_fullres_f14.safetensors"))
    print("unet fullres f14: out std %.4f, checksum %.6e" % (y.std(), out["checksum"]))


def gen_unet_fullres_lk(ref_lk, fsm_mod):
    """configs[2] at full resolution: ONE forward of the reference's LKGD UNet (unet_spatio_temporal_condition.py:448-693,
    domain / flow features) with the patch_FSM hook active in every spatial transformer block (patch_FSM.py:380-441),
    real width, 72 x 128 latent, CFG 2 x 2 frames, fp32 on the CPU"""
    from oracle.unet import SVD_CONFIG
    inp = fullres_inputs(lk=True)
    track, res = fullres_tracks()
    with torch.no_grad():
        m = ref_lk.UNetSpatioTemporalConditionModel(**SVD_CONFIG.__dict__)
        ou.init_weights_(m, FULLRES_LK_SEED)
        for p in m.parameters():
            p.copy_(p.half().float())
        out = {"checksum": torch.tensor(checksum(m), dtype=torch.float64)}
        out["lk_nohook"] = m(inp["sample"], inp["t"], inp["enc"], inp["domain"], inp["flow"],
                             added_time_ids=inp["ids"], return_dict=False)[0]
        fsm_mod.apply_patch(m, with_spatial_block=True, with_temporal_block=False)
        fsm_mod.initialize_joint_layers(m)
        seed_conv_fuse_([b for _, b in m.named_modules() if hasattr(b, "conv_fuse")])
        fsm_mod.update_patch(m, track=tuple(t.clone() for t in track), track_res=res)
        fsm_mod.set_joint_attention(m, True)
        out["lk_fsm"] = m(inp["sample"], inp["t"], inp["enc"], inp["domain"], inp["flow"],
                          added_time_ids=inp["ids"], return_dict=False)[0]
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "unet_fullres_lk.safetensors"))
    print("unet fullres lk: nohook std %.4f, fsm std %.4f, delta %.4f" % (
        out["lk_nohook"].std(), out["lk_fsm"].std(), (out["lk_fsm"] - out["lk_nohook"]).abs().max()))


def main():
    assert os.path.isdir(REF), "runs only where /root/reference is mounted"
    install_stubs()
    sys.path.insert(0, REF)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "cogvideox":
        return gen_cogvideox()
    if only == "scheduler_churn":
        _mod("utils")
        return gen_scheduler_churn(load_ref("utils/scheduling_euler_discrete_karras_fix.py",
                                            "utils.scheduling_euler_discrete_karras_fix"))
    if only == "patch_lora":
        for m in ("models", "utils"):
            _mod(m)
        ref_stock = load_ref("models/unet_spatio_temporal_condition_controlnet.py",
                             "models.unet_spatio_temporal_condition_controlnet")
        _mod("patch")
        load_ref("patch/utils.py", "patch.utils")
        return gen_patch_lora(load_ref("patch/patch.py", "patch.patch"), ref_stock)
    if only in ("loop_f14_cfg", "loop_f14_nocfg", "loop25_headline"):           # round-3 / round-5 fixtures
        for m in ("models", "utils"):
            _mod(m)
        sched_mod = load_ref("utils/scheduling_euler_discrete_karras_fix.py", "utils.scheduling_euler_discrete_karras_fix")
        ref_stock = load_ref("models/unet_spatio_temporal_condition_controlnet.py",
                             "models.unet_spatio_temporal_condition_controlnet")
        pipe_mod = load_ref("pipeline/pipeline_stable_video_diffusion_trans.py", "ref_pipeline_trans")
        if only == "loop25_headline":
            return gen_loop25_headline(pipe_mod, ref_stock, sched_mod)
        return gen_loop_f14(pipe_mod, ref_stock, sched_mod, only == "loop_f14_cfg")
    if only in ("loop25", "loop25_c1", "unet_fullres", "unet_fullres_lk", "unet_fullres_f14"):     # one at a time
        for m in ("models", "utils"):
            _mod(m)
        sched_mod = load_ref("utils/scheduling_euler_discrete_karras_fix.py", "utils.scheduling_euler_discrete_karras_fix")
        ref_stock = load_ref("models/unet_spatio_temporal_condition_controlnet.py",
                             "models.unet_spatio_temporal_condition_controlnet")
        if only == "unet_fullres":
            return gen_unet_fullres(ref_stock)
        if only == "unet_fullres_f14":
            return gen_unet_fullres_f14(ref_stock)
        if only == "unet_fullres_lk":
            ref_lk = load_ref("models/unet_spatio_temporal_condition.py", "models.unet_spatio_temporal_condition")
            _mod("patch")
            load_ref("patch/utils.py", "patch.utils")
            uo = _mod("utils.optical_flow")
            uo.warp_frames = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("name-only stub"))
            return gen_unet_fullres_lk(ref_lk, load_ref("patch/patch_FSM.py", "patch.patch_FSM"))
        pipe_mod = load_ref("pipeline/pipeline_stable_video_diffusion_trans.py", "ref_pipeline_trans")
        return (gen_loop25 if only == "loop25" else gen_loop25_c1)(pipe_mod, ref_stock, sched_mod)
    if len(sys.argv) > 1 and sys.argv[1] in ("unet_c1", "loop_c1"):   # only these fixtures (6 GB of fp32 weights, ~1 min each)
        for m in ("models", "utils"):
            _mod(m)
        sched_mod = load_ref("utils/scheduling_euler_discrete_karras_fix.py", "utils.scheduling_euler_discrete_karras_fix")
        ref_stock = load_ref("models/unet_spatio_temporal_condition_controlnet.py",
                             "models.unet_spatio_temporal_condition_controlnet")
        if sys.argv[1] == "unet_c1":
            gen_unet_c1(ref_stock)
        else:
            sys.modules["models.unet_spatio_temporal_condition_controlnet"] = ref_stock
            sys.modules["utils.scheduling_euler_discrete_karras_fix"] = sched_mod
            gen_loop_c1(load_ref("pipeline/pipeline_stable_video_diffusion_trans.py", "ref_pipeline_trans"), ref_stock,
                        sched_mod)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "image_ops":       # only this fixture (the others are unchanged)
        for m in ("models", "utils"):
            _mod(m)
        sched_mod = load_ref("utils/scheduling_euler_discrete_karras_fix.py", "utils.scheduling_euler_discrete_karras_fix")
        ref_stock = load_ref("models/unet_spatio_temporal_condition_controlnet.py",
                             "models.unet_spatio_temporal_condition_controlnet")
        sys.modules["models.unet_spatio_temporal_condition_controlnet"] = ref_stock
        sys.modules["utils.scheduling_euler_discrete_karras_fix"] = sched_mod
        gen_image_ops(load_ref("pipeline/pipeline_stable_video_diffusion_trans.py", "ref_pipeline_trans"))
        return
    sched_mod = load_ref("utils/scheduling_euler_discrete_karras_fix.py", "utils.scheduling_euler_discrete_karras_fix")
    gen_scheduler(sched_mod)
    gen_scheduler_churn(sched_mod)
    ref_stock = load_ref("models/unet_spatio_temporal_condition_controlnet.py",
                         "models.unet_spatio_temporal_condition_controlnet")
    ref_lk = load_ref("models/unet_spatio_temporal_condition.py", "models.unet_spatio_temporal_condition")
    gen_unet(ref_stock, ref_lk)
    gen_unet_c1(ref_stock)
    sys.modules["diffusers.models"].UNetSpatioTemporalConditionModel = ref_lk.UNetSpatioTemporalConditionModel
    ref_ctrl = load_ref("models/controlnet_sdv.py", "models.controlnet_sdv")
    gen_controlnet(ref_ctrl, ref_stock)
    _mod("patch")
    load_ref("patch/utils.py", "patch.utils")
    patch_mod = load_ref("patch/patch.py", "patch.patch")
    sys.modules["patch"].patch = patch_mod
    gen_patch(patch_mod)
    gen_patch_lora(patch_mod, ref_stock)
    uo = _mod("utils.optical_flow") if "utils" in sys.modules else None
    if uo is None:
        _mod("utils")
        uo = _mod("utils.optical_flow")
    uo.warp_frames = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("name-only stub"))   # imported, never called
    fsm_mod = load_ref("patch/patch_FSM.py", "patch.patch_FSM")
    gen_patch_fsm(fsm_mod)
    _mod("models")
    if "utils" not in sys.modules:
        _mod("utils")
    sys.modules["models.unet_spatio_temporal_condition_controlnet"] = ref_stock
    sys.modules["utils.scheduling_euler_discrete_karras_fix"] = sched_mod
    pipe_mod = load_ref("pipeline/pipeline_stable_video_diffusion_trans.py", "ref_pipeline_trans")
    gen_loop(pipe_mod, ref_stock, sched_mod)
    gen_loop_c1(pipe_mod, ref_stock, sched_mod)
    gen_image_ops(pipe_mod)
    gen_loop25(pipe_mod, ref_stock, sched_mod)
    gen_loop25_c1(pipe_mod, ref_stock, sched_mod)
    gen_unet_fullres(ref_stock)
    gen_unet_fullres_lk(ref_lk, fsm_mod)
    gen_cogvideox()


if __name__ == "__main__":
    main()
