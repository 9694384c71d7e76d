"""The CLIP image encoder on the HIP path (lkgd_amd/clip.py) against transformers' own fp32 forward of the same weights -
the module the reference calls at pipeline/pipeline_stable_video_diffusion_trans.py:164-203 (`self.image_encoder(image).image_embeds`).
transformers is third-party (not part of /root/reference); it is the checker here, never the thing run by the product path."""
import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")

from lkgd_amd.clip import CLIPVisionConfig, CLIPVisionModelWithProjection

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hf(cfg: CLIPVisionConfig, seed: int):
    torch.manual_seed(seed)
    c = transformers.CLIPVisionConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                                      patch_size=cfg.patch_size, image_size=cfg.image_size, projection_dim=cfg.projection_dim,
                                      hidden_act=cfg.hidden_act, layer_norm_eps=cfg.layer_norm_eps)
    m = transformers.CLIPVisionModelWithProjection(c).float().eval()
    with torch.no_grad():          # non-trivial norms and biases (the default init leaves them at 1 / 0)
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "norm" in n and n.endswith("weight"):
                p.uniform_(0.7, 1.3)
    return m


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_clip_small_matches_transformers(act):
    """two layers, 8 heads of 80 channels, 17 tokens, two images"""
    cfg = CLIPVisionConfig(hidden_size=640, intermediate_size=2560, num_hidden_layers=2, num_attention_heads=8, patch_size=14,
                           image_size=56, projection_dim=512, hidden_act=act)
    ref = _hf(cfg, 3)
    ours = CLIPVisionModelWithProjection(cfg)
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(DEV).half()
    torch.manual_seed(5)
    x = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        want = ref(pixel_values=x)
    got = ours(x.to(DEV))
    assert got.image_embeds.shape == want.image_embeds.shape and got.image_embeds.dtype == torch.float16
    assert _rel(got.last_hidden_state.float().cpu(), want.last_hidden_state) < 1e-2
    assert _rel(got.image_embeds.float().cpu(), want.image_embeds) < 1e-2


def test_clip_vit_h_full_size_matches_transformers():
    """the encoder of SVD's `image_encoder/`: ViT-H/14, 32 layers x 1280, 16 heads of 80, 257 tokens, projection 1024 (random init)"""
    cfg = CLIPVisionConfig()
    ref = _hf(cfg, 11)
    ours = CLIPVisionModelWithProjection(cfg)
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(DEV).half()
    torch.manual_seed(13)
    x = torch.randn(1, 3, 224, 224) * 1.2
    with torch.no_grad():
        want = ref(pixel_values=x).image_embeds
    got = ours(x.to(DEV)).image_embeds.float().cpu()
    assert torch.isfinite(got).all()
    assert _rel(got, want) < 1e-2
    cos = float(torch.nn.functional.cosine_similarity(got.double(), want.double()).min())
    assert cos > 0.9999
    # bitwise deterministic
    again = ours(x.to(DEV)).image_embeds.float().cpu()
    assert torch.equal(again, got)


def test_attn_dense_matches_sdpa():
    from lkgd_amd import ops
    torch.manual_seed(0)
    for (nb, S, heads, hd) in ((2, 257, 16, 80), (1, 17, 8, 80), (3, 50, 4, 128), (2, 64, 5, 64), (1, 1, 2, 8)):
        w = heads * hd
        qkv = torch.randn(nb * S, 3 * w, device=DEV, dtype=torch.float16)
        out = torch.empty(nb * S, w, device=DEV, dtype=torch.float16)
        ops.attn_dense(qkv[:, :w], qkv[:, w:2 * w], qkv[:, 2 * w:], out, nb, S, heads, hd)
        q, k, v = (qkv[:, i * w:(i + 1) * w].float().reshape(nb, S, heads, hd).transpose(1, 2) for i in range(3))
        want = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(nb * S, w)
        err = float((out.float() - want).abs().max())
        assert err < 2e-3, (nb, S, heads, hd, err)
    # shapes the kernel must refuse: K and V of a head do not fit the LDS; head_dim not a multiple of 8
    from lkgd_amd._lib import LkgdHipError
    big = torch.zeros(4096, 3 * 64, device=DEV, dtype=torch.float16)
    with pytest.raises(LkgdHipError):
        ops.attn_dense(big[:, :64], big[:, 64:128], big[:, 128:], torch.empty(4096, 64, device=DEV, dtype=torch.float16), 1, 4096, 1, 64)
