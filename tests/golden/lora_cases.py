"""Seeded weights / inputs of the masked-LoRA fixture (tests/golden/patch_lora.safetensors): the generator script and the
tests fill the joint layers and the LoRA matrices of name-for-name identical parameter trees from here."""
import torch

LORA_SEED = 171
ADAPTERS = ("xy_lora", "yx_lora")           # utils/util.py:570-604
RANK, ALPHA = 4, 8.0
JOINT_MASK = [0, 1, 0, 1]
MASKS = {"yx_lora": [0, 1, 0, 1], "xy_lora": [1, 0, 1, 0]}


def seed_joint_and_lora_(model, seed=LORA_SEED + 1):
    """fill every parameter whose name contains attn1n / conv1n / lora_A / lora_B, in sorted-name order"""
    g = torch.Generator().manual_seed(seed)
    ps = sorted((n, p) for n, p in model.named_parameters()
                if any(k in n for k in ("attn1n", "conv1n", "lora_A", "lora_B")))
    with torch.no_grad():
        for n, p in ps:
            if "lora_" in n:
                v = torch.randn(p.shape, generator=g) * (0.5 / p.shape[-1] ** 0.5)
            elif "conv1n" in n:
                v = torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5
            elif p.ndim >= 2:
                v = torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5
            else:
                v = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(v.half().float().to(p.dtype))
    return [n for n, _ in ps]


def lora_inputs(seed=LORA_SEED + 2, frames=4, hw=8):
    g = torch.Generator().manual_seed(seed)
    return dict(sample=torch.randn(4, frames, 8, hw, hw, generator=g).half().float(), t=torch.tensor(1.1),
                enc=torch.randn(4, 1, 1024, generator=g).half().float(), ids=torch.tensor([[6.0, 127.0, 0.02]] * 4))
