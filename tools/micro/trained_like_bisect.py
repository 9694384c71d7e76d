#!/usr/bin/env python3
"""which ingredient of the 'trained-like' weight statistics (tests/test_unet_gpu.py) separates the HIP forward from the oracle?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from oracle import unet as ou
from lkgd_amd import unet as pu
DEV = "cuda:0"

def make(flags, seed=48):
    o = ou.init_weights_(ou.UNetSpatioTemporalConditionControlNetModel(ou.TINY_CONFIG), 47)
    g = torch.Generator().manual_seed(seed)
    t3 = torch.distributions.StudentT(3.0)
    with torch.no_grad():
        for name, m in o.named_modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                if "gain" in flags: m.weight.copy_(torch.exp(0.6 * torch.randn(m.weight.shape, generator=g)))
                if "nbias" in flags: m.bias.copy_(0.5 * torch.randn(m.bias.shape, generator=g))
            elif isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)):
                fan_in = m.weight[0].numel()
                if "tails" in flags:
                    torch.manual_seed(int(torch.randint(0, 2 ** 31, (1,), generator=g)))
                    m.weight.copy_((t3.sample(m.weight.shape) / 3 ** 0.5).clamp(-12, 12) / fan_in ** 0.5)
                if (name.endswith("to_q") or name.endswith("to_k")) and ".attn1" in name:
                    for f in flags:
                        if f.startswith("qk"):
                            if "spatial" in f and "temporal_transformer" in name: continue
                            if "temporal" in f and "temporal_transformer" not in name: continue
                            m.weight.mul_(float(f.split("x")[1]))
                if "bias" in flags and m.bias is not None:
                    m.bias.copy_(0.3 * torch.randn(m.bias.shape, generator=g))
        for p in o.parameters():
            p.copy_(p.half().float())
    m = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    m.load_state_dict(o.state_dict(), strict=True)
    return o, m.half().to(DEV)

g = torch.Generator().manual_seed(49)
cfg = ou.TINY_CONFIG
x = torch.randn(2, 4, cfg.in_channels, 16, 16, generator=g).half().float()
enc = torch.randn(2, 1, cfg.cross_attention_dim, generator=g).half().float()
ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
for flags in (("qkx1.5",), ("qkx2",), ("qkx4",), ("qk_spatialx4",), ("qk_temporalx4",)):
    o, m = make(flags)
    with torch.no_grad():
        ref = o(x, torch.tensor(1.25), enc, added_time_ids=ids, return_dict=False)[0]
    with torch.no_grad():
        floors = []
        for s_ in range(3):
            gg = torch.Generator().manual_seed(100 + s_)
            xp = (x * (1 + 4.9e-4 * (2 * torch.rand(x.shape, generator=gg) - 1))).half().float()     # one fp16 rounding of the input
            r2 = o(xp, torch.tensor(1.25), enc, added_time_ids=ids, return_dict=False)[0]
            floors.append(((r2 - ref).norm() / ref.norm()).item())
    print("   oracle's own sensitivity to ONE fp16 rounding of its input:", " ".join(f"{f:.2e}" for f in floors))
    out = m(x.to(DEV), torch.tensor(1.25).to(DEV), enc.to(DEV), added_time_ids=ids.to(DEV), return_dict=False)[0].float().cpu()
    print(f"{'+'.join(flags) or 'baseline':30s} rel L2 {((out-ref).norm()/ref.norm()).item():.3e}  ref std {ref.std():.3f} max {ref.abs().max():.2f} finite {bool(torch.isfinite(out).all())}", flush=True)
