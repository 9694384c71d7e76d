"""The domain / flow ViT encoders (timm vit_base_patch16_384 [EXT], PARITY UNPINNED) on the HIP path
against the fp32 oracle restatement, the fused resize + patch-unfold kernel against F.interpolate + unfold, and the wiring
from a clip's frames to the UNet's `domain_features` / `flow_features` inputs."""
import pytest
import torch

DEV = "cuda:0"


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _pair(cfg, seed):
    from lkgd_amd import vit as pv
    from oracle import vit as ov
    o = ov.init_weights_(ov.VisionTransformer(cfg), seed)
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.half().float())
    m = pv.VisionTransformer(pv.ViTConfig(**cfg.__dict__))
    m.load_state_dict(o.state_dict())
    return o, m


def test_vit_parameter_tree_and_checkpoint_loading():
    """timm's parameter names / 86 859 496 parameters; `encoder.`-prefixed checkpoints load as train_svd_lora.py:1419-1433"""
    from lkgd_amd import vit as pv
    from oracle import vit as ov
    with torch.device("meta"):
        m = pv.vit_base_patch16_384()
    assert sum(p.numel() for p in m.parameters()) == 86859496
    names = set(m.state_dict())
    for k in ("cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.11.attn.qkv.bias", "blocks.0.mlp.fc2.weight",
              "blocks.5.norm2.weight", "norm.bias", "head.weight"):
        assert k in names
    assert m.state_dict()["pos_embed"].shape == (1, 577, 768)
    o, t = _pair(ov.TINY_VIT, 3)
    ck = {"encoder." + k: v for k, v in o.state_dict().items()}
    ck["decoder.something"] = torch.zeros(3)
    t2 = pv.VisionTransformer(pv.ViTConfig(**ov.TINY_VIT.__dict__))
    t2.load_encoder_checkpoint(ck)
    assert all(torch.equal(a, b) for a, b in zip(t.state_dict().values(), t2.state_dict().values()))


@pytest.mark.gpu
def test_vit_patchify_vs_interpolate_unfold():
    import torch.nn.functional as F
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(1)
    for (H, W, S, P) in ((50, 70, 64, 16), (576, 1024, 384, 16), (384, 384, 384, 16), (20, 24, 64, 16)):
        x = torch.rand(2, 3, H, W, generator=g)
        ref = F.unfold(F.interpolate(x, size=[S, S], mode="bilinear"), kernel_size=P, stride=P)      # [N, C*P*P, L]
        ref = ref.transpose(1, 2).reshape(-1, 3 * P * P)
        got = ops.vit_patchify(x.to(DEV), S, P)
        assert got.shape == ref.shape and (got.float().cpu() - ref).abs().max() < 1e-3, (H, W)


@pytest.mark.gpu
def test_vit_tiny_vs_oracle():
    from oracle import vit as ov
    o, m = _pair(ov.TINY_VIT, 5)
    m = m.half().to(DEV)
    x = torch.rand(3, 3, 40, 56, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ref = o(torch.nn.functional.interpolate(x, size=[64, 64], mode="bilinear"))
    got = m(x.to(DEV))
    assert got.shape == (3, 40) and _rel(got, ref) < 1e-2, _rel(got, ref)


@pytest.mark.gpu
def test_vit_b16_384_vs_oracle_and_lk_wiring():
    """the real shape (12 blocks, 768 wide, 577 tokens) on 2 clips x 2 frames, then the [B, 1, 1000] features straight into the
    LKGD UNet's fused embedding (train_svd_lora.py:1455-1466 -> models/unet_spatio_temporal_condition.py:536-595)"""
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    from oracle import vit as ov
    o, m = _pair(ov.VIT_B16_384, 7)
    m = m.half().to(DEV)
    frames = torch.rand(2, 2, 3, 96, 128, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        ref = ov.clip_features(o, frames)
    got = m.clip_features(frames.to(DEV))
    r = _rel(got, ref)
    print(f"\nViT-B/16-384 clip features vs oracle: rel L2 {r:.3e}")
    assert got.shape == (2, 1, 1000) and r < 1e-2
    lk = pu.UNetSpatioTemporalConditionModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__)).half().to(DEV)
    pu.init_synthetic_weights_(lk, 9)
    enc = torch.randn(2, 1, 1024, generator=torch.Generator().manual_seed(10)).half().to(DEV)
    fused = lk.fused_embedding(enc, got[:1], got[:1])
    assert fused.shape == (2, 1, 1024) and torch.isfinite(fused.float()).all()
