"""The CogVideoX DiT loop with LKGD's latent-knowledge fuse (BASELINE.json configs[4]).
tests/golden/cogvideox.safetensors = the reference's own in-tree `CogVideoXTransformer3DModel.forward`
(CogVideo-main/finetune/models/cogvideox_i2v/cogvideox_transformer_3d.py:473-638, blocks :41-160, LK modules :337-366) executed over
the restated diffusers >= 0.32 pieces of oracle/cogvideox.py (PARITY UNPINNED for those interiors) - make_goldens.py::gen_cogvideox."""
import os

import pytest
import torch
from safetensors.torch import load_file

DEV = "cuda:0"
DIT_SEED = 191


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _inputs(cfg, seed=DIT_SEED + 1, batch=2):
    g = torch.Generator().manual_seed(seed)
    f = (cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1
    return dict(hidden=torch.randn(batch, f, cfg.in_channels, cfg.sample_height, cfg.sample_width, generator=g).half().float(),
                text=torch.randn(batch, cfg.max_text_seq_length, cfg.text_embed_dim, generator=g).half().float(),
                t=torch.tensor([721] * batch), domain=torch.randn(1, 1, 1000, generator=g),
                flow=torch.randn(1, 1, 1000, generator=g))


def _oracle(cfg):
    from oracle import cogvideox as oc
    o = oc.init_weights_(oc.CogVideoXTransformer3DModel(cfg), DIT_SEED)
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.half().float())
    return o


def _hip(o, cfg):
    from lkgd_amd import cogvideox as pc
    m = pc.CogVideoXTransformer3DModel(pc.DiTConfig(**cfg.__dict__))
    missing, unexpected = m.load_state_dict(o.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


@pytest.fixture(scope="module")
def golden(golden_dir):
    return load_file(os.path.join(golden_dir, "cogvideox.safetensors"))


def test_oracle_dit_vs_reference_golden(golden):
    from oracle import cogvideox as oc
    o = _oracle(oc.TINY_DIT)
    ck = float(sum(p.detach().double().abs().sum() for p in o.parameters()))
    assert abs(ck - golden["checksum"].item()) <= 1e-9 * ck
    i = _inputs(oc.TINY_DIT)
    with torch.no_grad():
        y = o(i["hidden"], i["text"], i["t"], i["domain"], i["flow"])[0]
        fused = o.lk_fuse(i["text"], i["domain"], i["flow"])
    assert y.shape == golden["out"].shape == (2, 3, 16, 8, 12)
    assert _rel(y, golden["out"]) < 1e-5 and _rel(fused, golden["fused_text"]) < 1e-5


def test_dit_structure_scheduler_and_names():
    """parameter names / counts of the 2B image-to-video transformer, and the [EXT] DDIM tables' invariants"""
    from lkgd_amd import cogvideox as pc
    from oracle import cogvideox as oc
    with torch.device("meta"):
        o, m = oc.CogVideoXTransformer3DModel(oc.COGVIDEOX_2B_I2V), pc.CogVideoXTransformer3DModel(pc.DiTConfig(in_channels=32))
    so = {k: tuple(v.shape) for k, v in o.state_dict().items() if k != "patch_embed.pos_embedding"}
    sm = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert so == sm
    assert sum(p.numel() for p in m.parameters()) == 1696737100
    assert sum(p.numel() for n, p in m.named_parameters() if "quaternion" not in n) == 1693906752
    for k in ("transformer_blocks.29.attn1.norm_q.weight", "transformer_blocks.0.norm1.linear.bias", "transformer_blocks.7.ff.net.0.proj.weight",
              "transformer_blocks.7.ff.net.2.bias", "patch_embed.text_proj.weight", "patch_embed.proj.weight", "norm_out.linear.weight",
              "norm_final.bias", "proj_out.weight", "time_embedding.linear_2.weight", "quaternion_lora_fuse.r_weight"):
        assert k in sm, k
    for S in (oc.CogVideoXDDIMScheduler, pc.CogVideoXDDIMScheduler):
        s = S()
        s.set_timesteps(50)
        ts = s.timesteps.tolist()
        assert ts[0] == 999 and ts[1] == 979 and ts[-1] == 19 and len(ts) == 50              # trailing spacing
        assert float(s.alphas_cumprod[-1]) == 0.0 and 0.99 < float(s.alphas_cumprod[0]) < 1.0     # zero terminal SNR
        a, b, sa, sb = s.coefficients(999)
        assert sa == 0.0 and sb == 1.0                                                        # x0 = -v at pure noise
        a, b, sa, sb = s.coefficients(19)                                                     # last step lands on x0
        assert abs(a) < 1e-12 and abs(b - 1.0) < 1e-12
    assert abs(pc.dynamic_guidance(6.0, 50, 999) - 7.0) < 1e-9 or pc.dynamic_guidance(6.0, 50, 999) > 1.0
    pe = pc.sincos_pos_embed_3d(128, 6, 4, 3, 1.875, 1.0)
    assert torch.equal(pe, o_pos := torch.from_numpy(oc.get_3d_sincos_pos_embed(128, (6, 4), 3, 1.875, 1.0)).float().flatten(0, 1)) and o_pos.shape == (72, 128)


@pytest.mark.gpu
def test_dit_kernels_vs_torch():
    import torch.nn.functional as F
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (3 * torch.randn(64, 256, generator=g)).half().to(DEV)
    assert (ops.gelu_tanh_(x.clone()).float() - F.gelu(x.float(), approximate="tanh")).abs().max() < 4e-3
    x, res = torch.randn(2 * 11, 64, generator=g).half().to(DEV), torch.randn(2 * 11, 64, generator=g).half().to(DEV)
    gate = torch.randn(4, 64, generator=g).to(DEV)
    idx = torch.tensor([b * 2 + (r >= 3) for b in range(2) for r in range(11)], device=DEV)
    ref = res.float() + gate[idx] * x.float()
    assert (ops.gated_add(x, gate, res, 11, 3).float() - ref).abs().max() < 4e-3
    for C_ in (1920, 2048, 1600):                       # the wide LayerNorm rows of the DiT
        x = torch.randn(37, C_, generator=g).half().to(DEV)
        ga, be = torch.randn(C_, generator=g).to(DEV), torch.randn(C_, generator=g).to(DEV)
        assert (ops.layernorm(x, ga, be, 1e-5).float() - F.layer_norm(x.float(), (C_,), ga, be, 1e-5)).abs().max() < 8e-3
        assert (ops.layernorm(x, None, None, 1e-5).float() - F.layer_norm(x.float(), (C_,), eps=1e-5)).abs().max() < 4e-3
    q = torch.randn(40 * 3, 64, generator=g).half().to(DEV)          # per-head qk norm in place
    ga, be = torch.randn(64, generator=g).to(DEV), torch.randn(64, generator=g).to(DEV)
    ref = F.layer_norm(q.float(), (64,), ga, be, 1e-6)
    ops.layernorm(q, ga, be, 1e-6, out=q)
    assert (q.float() - ref).abs().max() < 8e-3


@pytest.mark.gpu
def test_hip_dit_forward_vs_reference_golden(golden):
    from oracle import cogvideox as oc
    m = _hip(_oracle(oc.TINY_DIT), oc.TINY_DIT).half().to(DEV)
    i = _inputs(oc.TINY_DIT)
    fused = m.fused_text(i["text"].to(DEV), i["domain"].to(DEV), i["flow"].to(DEV))
    assert _rel(fused, golden["fused_text"]) < 2e-3
    out = m(i["hidden"].to(DEV), i["text"].to(DEV), i["t"].to(DEV), i["domain"].to(DEV), i["flow"].to(DEV), return_dict=False)[0]
    r = _rel(out, golden["out"])
    print(f"\nHIP CogVideoX DiT forward vs the reference: rel L2 {r:.3e}")
    assert out.shape == golden["out"].shape and r < 1e-2 and (out.float().cpu() - golden["out"]).abs().max() < 5e-2


@pytest.mark.gpu
def test_hip_dit_loop_vs_oracle():
    """pipeline_cogvideox_image2video.py:829-885: 4 DDIM steps with dynamic CFG, tiny DiT, against the oracle's loop"""
    from lkgd_amd import cogvideox as pc
    from oracle import cogvideox as oc
    cfg = oc.TINY_DIT
    o = _oracle(cfg)
    m = _hip(o, cfg).half().to(DEV)
    g = torch.Generator().manual_seed(5)
    f = 3
    lat = torch.randn(1, f, 16, cfg.sample_height, cfg.sample_width, generator=g)
    img = (0.5 * torch.randn(1, f, 16, cfg.sample_height, cfg.sample_width, generator=g)).half().float()
    pe = torch.randn(2, cfg.max_text_seq_length, cfg.text_embed_dim, generator=g).half().float()
    dom, flow = torch.randn(1, 1, 1000, generator=g), torch.randn(1, 1, 1000, generator=g)
    ref_steps, got_steps = [], []
    ref = oc.denoise(o, oc.CogVideoXDDIMScheduler(), lat.half().float(), img, pe, dom, flow, 4, 6.0, True,
                     callback=lambda i, t, l: ref_steps.append(l.clone()))
    got = pc.denoise(m, pc.CogVideoXDDIMScheduler(), lat.half().to(DEV), img.to(DEV), pe.to(DEV), dom.to(DEV), flow.to(DEV), 4, 6.0,
                     True, callback=lambda i, t, l: got_steps.append(l.clone()))
    for i, (a, b) in enumerate(zip(got_steps, ref_steps)):
        assert _rel(a, b) < 2e-2, (i, _rel(a, b))
    assert _rel(got, ref) < 2e-2 and torch.isfinite(got.float()).all()


@pytest.mark.gpu
def test_hip_dit_real_width_vs_oracle():
    """the 2B model's width (30 heads x 64 = 1920 channels: the wide LayerNorm rows, 1920 / 7680-column GEMMs, 226 text tokens)
    with 2 layers on a small video (5 latent frames of 16 x 24 -> 480 video tokens), against the oracle"""
    from oracle import cogvideox as oc
    cfg = oc.DiTConfig(in_channels=32, num_layers=2, sample_width=24, sample_height=16, sample_frames=17)
    o = _oracle(cfg)
    m = _hip(o, cfg).half().to(DEV)
    i = _inputs(cfg, seed=7)
    with torch.no_grad():
        ref = o(i["hidden"], i["text"], i["t"], i["domain"], i["flow"])[0]
    out = m(i["hidden"].to(DEV), i["text"].to(DEV), i["t"].to(DEV), i["domain"].to(DEV), i["flow"].to(DEV), return_dict=False)[0]
    r = _rel(out, ref)
    print(f"\nreal-width (1920) 2-layer DiT vs oracle: rel L2 {r:.3e}")
    assert out.shape == ref.shape == (2, 5, 16, 16, 24) and r < 1e-2
