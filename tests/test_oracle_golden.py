"""Pins the CPU oracle against fixtures produced by the reference's own code (tests/golden/make_goldens.py)."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import blocks as ob
from oracle import patch_hooks as oph
from oracle import unet as ou
from oracle.loop import denoise
from oracle.scheduler import EulerDiscreteOracle, SchedulerConfig

WSEED = 7  # tests/golden/make_goldens.py


def _checksum(m):
    return float(sum(p.detach().double().abs().sum() for p in m.parameters()))


# ------------------------------------------------------------------------------------------------ scheduler (a2,a3)
@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "scheduler_kat.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_scheduler_tables_and_steps(kat, idx):
    case = kat["cases"][idx]
    s = EulerDiscreteOracle(SchedulerConfig(**kat["config"]))
    s.set_timesteps(case["n"])
    assert torch.equal(s.sigmas, torch.tensor(case["sigmas"]))
    assert torch.equal(s.timesteps, torch.tensor(case["timesteps"]))
    assert float(s.init_noise_sigma) == case["init_noise_sigma"]
    x = torch.tensor(case["x0"]).reshape(1, 2, 4, 3, 3)
    for t, st in zip(s.timesteps, case["steps"]):
        v = torch.tensor(st["v"]).reshape(x.shape)
        assert torch.equal(s.scale_model_input(x, t).flatten(), torch.tensor(st["scaled"]))
        x = s.step(v, t, x)
        assert torch.equal(x.flatten(), torch.tensor(st["prev"]))


def test_scheduler_stochastic_step(golden_dir):
    """s_churn > 0 (scheduling_euler_discrete_karras_fix.py:485-497): the oracle against the reference's own 25 steps,
    fp16 model outputs, caller's generator - bit-exact"""
    with open(os.path.join(golden_dir, "scheduler_churn_kat.json")) as f:
        ck = json.load(f)
    for case in ck["cases"]:
        s = EulerDiscreteOracle(SchedulerConfig(prediction_type=case["prediction_type"]))
        s.set_timesteps(25)
        gn = torch.Generator().manual_seed(case["noise_seed"])
        x = torch.tensor(case["x0"]).reshape(1, 2, 4, 3, 3)
        changed = 0
        for t, st in zip(s.timesteps, case["steps"]):
            v = torch.tensor(st["v"]).reshape(x.shape).half()
            plain = s.step(v, t, x)
            s._step_index -= 1
            x = s.step(v, t, x, generator=gn, **ck["churn"])
            assert torch.equal(x.float().flatten(), torch.tensor(st["prev"]))
            changed += int(not torch.equal(plain, x))
        assert 15 <= changed < 25          # gamma = 0 while sigma > s_tmax, sqrt(2) - 1 afterwards


def test_scheduler_survey_kat(kat):
    # values produced by the reference file itself
    c25 = [c for c in kat["cases"] if c["n"] == 25][0]
    assert c25["sigmas"][0] == 700.0 and c25["sigmas"][-1] == 0.0 and len(c25["sigmas"]) == 26
    assert abs(c25["sigmas"][1] - 545.729248) < 1e-4 and abs(c25["sigmas"][24] - 0.002) < 1e-9
    assert abs(c25["timesteps"][0] - 1.63777) < 1e-4 and abs(c25["timesteps"][24] + 1.553652) < 1e-4
    assert abs(c25["init_noise_sigma"] - 700.000732) < 1e-3
    assert abs(kat["scalar_kat"]["scaled"] - 0.0014285699) < 1e-9
    assert abs(kat["scalar_kat"]["prev"] - 0.6694203615) < 1e-7


# ------------------------------------------------------------------------------------------------ UNet wiring (a4-a6)
@pytest.fixture(scope="module")
def wiring(golden_dir):
    return load_file(os.path.join(golden_dir, "unet_wiring.safetensors"))


def _skip_shapes(cfg, frames, hw):
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


def test_unet_stock_wiring(wiring):
    m = ou.init_weights_(ou.UNetSpatioTemporalConditionControlNetModel(ou.TINY_CONFIG), WSEED)
    assert _checksum(m) == pytest.approx(float(wiring["stock_checksum"]), rel=1e-12)
    g = wiring
    with torch.no_grad():
        y = m(g["in_sample"], g["in_t"], g["in_enc"], added_time_ids=g["in_ids"], return_dict=False)[0]
        torch.testing.assert_close(y, g["stock_out"], rtol=1e-5, atol=1e-5)
        gg = torch.Generator().manual_seed(12)
        shapes, mid_shape = _skip_shapes(ou.TINY_CONFIG, 4, 8)
        down = tuple(0.1 * torch.randn(s, generator=gg) for s in shapes)
        mid = 0.1 * torch.randn(mid_shape, generator=gg)
        y = m(g["in_sample"], g["in_t"], g["in_enc"], down_block_additional_residuals=down,
              mid_block_additional_residual=mid, added_time_ids=g["in_ids"], return_dict=False)[0]
        torch.testing.assert_close(y, g["stock_out_ctrl"], rtol=1e-5, atol=1e-5)
        y = m(g["in_sample"], 0.5, g["in_enc"], added_time_ids=g["in_ids"]).sample
        torch.testing.assert_close(y, g["stock_out_tfloat"], rtol=1e-5, atol=1e-5)


def test_unet_lk_wiring_and_fuse(wiring):
    m = ou.init_weights_(ou.UNetSpatioTemporalConditionModel(ou.TINY_CONFIG), WSEED + 1)
    assert _checksum(m) == pytest.approx(float(wiring["lk_checksum"]), rel=1e-12)
    g = wiring
    with torch.no_grad():
        fused = m.lk_fuse(g["in_enc"], g["in_domain"], g["in_flow"])
        torch.testing.assert_close(fused, g["lk_fused_enc"], rtol=1e-5, atol=1e-5)
        y = m(g["in_sample"], g["in_t"], g["in_enc"], g["in_domain"], g["in_flow"], added_time_ids=g["in_ids"],
              return_dict=False)[0]
        torch.testing.assert_close(y, g["lk_out"], rtol=1e-5, atol=1e-5)


def test_structural_parameter_count():
    # the released SVD unet has 1 524 623 082 parameters; LK extras add 726 796
    with torch.device("meta"):
        a = ou.UNetSpatioTemporalConditionControlNetModel(ou.SVD_CONFIG)
        b = ou.UNetSpatioTemporalConditionModel(ou.SVD_CONFIG)
    na = sum(p.numel() for p in a.parameters())
    assert na == 1_524_623_082
    assert sum(p.numel() for p in b.parameters()) - na == 726_796


# ------------------------------------------------------------------------------------------------ ControlNet (8f rank 1)
def test_controlnet_encoder(golden_dir):
    """models/controlnet_sdv.py: conditioning embedding, encoder, zero convolutions, conditioning_scale, and the
    residuals through the stock UNet - against the reference's own classes"""
    from oracle import controlnet as oc
    g = load_file(os.path.join(golden_dir, "controlnet.safetensors"))
    c = ou.init_weights_(oc.ControlNetSDVModel(ou.TINY_CONFIG), WSEED + 5)
    assert abs(float(sum(p.double().abs().sum() for p in c.parameters())) - float(g["checksum"])) < 1e-6
    with torch.no_grad():
        down, mid = c(g["in_sample"], g["in_t"], g["in_enc"], g["in_ids"], controlnet_cond=g["in_cond"],
                      return_dict=False, conditioning_scale=0.75)
        assert len(down) == 12
        for i, d in enumerate(down):
            torch.testing.assert_close(d, g[f"down_{i}"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(mid, g["mid"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(c.controlnet_cond_embedding(g["in_cond"]), g["cond_embedding"], rtol=1e-5, atol=1e-6)
        down0, mid0 = c(g["in_sample"], g["in_t"], g["in_enc"], g["in_ids"], return_dict=False)
        torch.testing.assert_close(down0[3], g["nocond_down_3"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(mid0, g["nocond_mid"], rtol=1e-4, atol=1e-4)
        u = ou.init_weights_(ou.UNetSpatioTemporalConditionControlNetModel(ou.TINY_CONFIG), WSEED)
        y = u(g["in_sample"], g["in_t"], g["in_enc"], down_block_additional_residuals=down,
              mid_block_additional_residual=mid, added_time_ids=g["in_ids"], return_dict=False)[0]
        torch.testing.assert_close(y, g["unet_out"], rtol=1e-4, atol=1e-4)
    # fresh construction: the zero convolutions make every residual exactly zero
    z = oc.ControlNetSDVModel(ou.TINY_CONFIG)
    with torch.no_grad():
        dz, mz = z(g["in_sample"], g["in_t"], g["in_enc"], g["in_ids"], controlnet_cond=g["in_cond"], return_dict=False)
    assert all(float(d.abs().max()) == 0.0 for d in dz) and float(mz.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ patch hooks (a14)
class _Holder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.spatial = ob.BasicTransformerBlock(128, 2, 64, 1024)
        self.temporal = ob.TemporalBasicTransformerBlock(128, 128, 2, 64, 1024)


def test_patch_joint_hooks(golden_dir):
    g = load_file(os.path.join(golden_dir, "patch_joint.safetensors"))
    torch.manual_seed(21)
    h = ou.init_weights_(_Holder(), 21)
    mask = torch.tensor([False, True, False, True])
    x, enc, tctx = g["in_x"], g["in_enc"], g["in_tctx"]
    with torch.no_grad():
        for name, blk in (("spatial", h.spatial), ("temporal", h.temporal)):
            oph.initialize_joint_layers(blk, "conv")
            # zero-init conv1n => joint branch is the identity
            if name == "spatial":
                y = oph.basic_block_forward(blk, x, enc, mask, enable_joint=True)
                torch.testing.assert_close(y, g["spatial_nojoint"], rtol=1e-5, atol=1e-5)
            blk.conv1n.weight.copy_(g[f"conv1n_{name}"])
            blk.attn1n.load_state_dict({k[len(f"attn1n_{name}."):]: v for k, v in g.items()
                                        if k.startswith(f"attn1n_{name}.")})
        y = oph.basic_block_forward(h.spatial, x, enc, mask, enable_joint=True)
        torch.testing.assert_close(y, g["spatial_joint_noflip"], rtol=1e-5, atol=1e-5)
        y = oph.basic_block_forward(h.spatial, x, enc, mask, enable_joint=True, flip=True, n_frames=3)
        torch.testing.assert_close(y, g["spatial_joint_flip"], rtol=1e-5, atol=1e-5)
        h.spatial.joint_scale = 0.5
        y = oph.basic_block_forward(h.spatial, x, enc, mask, enable_joint=True)
        torch.testing.assert_close(y, g["spatial_joint_scale05"], rtol=1e-5, atol=1e-5)
        y = oph.basic_block_forward(h.spatial, x, enc, mask, enable_joint=False)
        torch.testing.assert_close(y, g["spatial_nojoint"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(h.spatial(x, encoder_hidden_states=enc), g["spatial_nojoint"], rtol=1e-5, atol=1e-5)
        y = oph.temporal_block_forward(h.temporal, x, 3, tctx, mask, enable_joint=True)
        torch.testing.assert_close(y, g["temporal_joint"], rtol=1e-5, atol=1e-5)
        y = oph.temporal_block_forward(h.temporal, x, 3, tctx, mask, enable_joint=False)
        torch.testing.assert_close(y, g["temporal_nojoint"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(h.temporal(x, 3, tctx), g["temporal_nojoint"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("post", ["scale", "conv_fuse"])
def test_patch_joint_post_variants(golden_dir, post):
    """post = 'scale' / 'conv_fuse' of initialize_joint_layers (patch.py:146-158, applied at :484-494, :649-652)"""
    g = load_file(os.path.join(golden_dir, "patch_joint.safetensors"))
    torch.manual_seed(21)
    h = ou.init_weights_(_Holder(), 21)
    mask = torch.tensor([False, True, False, True])
    x, enc, tctx = g["in_x"], g["in_enc"], g["in_tctx"]
    with torch.no_grad():
        for name, blk in (("spatial", h.spatial), ("temporal", h.temporal)):
            oph.initialize_joint_layers(blk, post)
            if post == "scale":
                blk.scale1n.copy_(g[f"{post}.scale1n_{name}"])
            else:
                blk.conv1n.weight.copy_(g[f"{post}.conv1n_{name}"])
            pre = f"{post}.attn1n_{name}."
            blk.attn1n.load_state_dict({k[len(pre):]: v for k, v in g.items() if k.startswith(pre)})
        h.spatial.joint_scale = 0.75
        y = oph.basic_block_forward(h.spatial, x, enc, mask, enable_joint=True)
        torch.testing.assert_close(y, g[f"{post}.spatial_joint"], rtol=1e-5, atol=1e-5)
        y = oph.temporal_block_forward(h.temporal, x, 3, tctx, mask, enable_joint=True)
        torch.testing.assert_close(y, g[f"{post}.temporal_joint"], rtol=1e-5, atol=1e-5)


def test_patch_fsm_hook(golden_dir):
    """a15: track-guided fuse of patch/patch_FSM.py:380-441 against the reference's own ToMeBlock output"""
    g = load_file(os.path.join(golden_dir, "patch_fsm.safetensors"))
    torch.manual_seed(51)
    h = ou.init_weights_(_Holder(), 51)
    track = (g["src_tracks"], g["dst_tracks"], g["vis"])
    res = tuple(int(v) for v in g["track_res"])
    x, enc = g["in_x"], g["in_enc"]
    with torch.no_grad():
        oph.initialize_fsm_layers(h.spatial)
        y = oph.fsm_block_forward(h.spatial, x, enc, track, res)
        torch.testing.assert_close(y, g["fsm_zero_init"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(y, g["fsm_off"], rtol=1e-5, atol=1e-5)
        h.spatial.conv_fuse.weight.copy_(g["conv_fuse_w"])
        h.spatial.conv_fuse.bias.copy_(g["conv_fuse_b"])
        y = oph.fsm_block_forward(h.spatial, x, enc, track, res)
        torch.testing.assert_close(y, g["fsm_on"], rtol=1e-5, atol=1e-5)
        assert (g["fsm_on"] - g["fsm_off"]).abs().max() > 0.5
        y = oph.fsm_block_forward(h.spatial, x, enc, track, res, enable=False)
        torch.testing.assert_close(y, g["fsm_off"], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ loop (a1)
def test_loop_against_reference_pipeline(golden_dir):
    g = load_file(os.path.join(golden_dir, "loop.safetensors"))
    m = ou.init_weights_(ou.UNetSpatioTemporalConditionControlNetModel(ou.TINY_CONFIG), WSEED)
    steps = []
    with torch.no_grad():
        y0 = m(g["unet_in0"], EulerDiscreteOracle().timesteps[0] * 0 + 1.6377699375152588, g["image_embeddings"],
               added_time_ids=g["added_time_ids"], return_dict=False)[0]
        torch.testing.assert_close(y0, g["unet_out0"], rtol=1e-4, atol=1e-4)
        out = denoise(m, EulerDiscreteOracle(), g["latents0"], g["image_latents"], g["image_embeddings"],
                      g["added_time_ids"], 3, callback=lambda i, t, l: steps.append(l.clone()))
    torch.testing.assert_close(torch.stack(steps), g["step_latents"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out, g["final"], rtol=1e-4, atol=1e-4)


def test_rfft_phase_convention_of_real_bins():
    """lkgd_lk_fuse's direct DFT leaves the imaginary part of the DC and Nyquist bins at exactly +0, so atan2 gives +pi for a
    negative real part (csrc/lk_fuse.hip); torch.angle(torch.fft.rfft(x)) - what the reference and the oracle run - agrees on the
    host: never -0 / -pi on those bins"""
    import math
    import torch
    g = torch.Generator().manual_seed(7)
    seen = 0
    for _ in range(50):
        x = torch.randn(4, 256, generator=g)
        X = torch.fft.rfft(x, dim=-1)
        for b in (0, 128):
            neg = X[:, b].real < 0
            seen += int(neg.sum())
            assert not bool(torch.signbit(X[:, b].imag).any())
            assert bool((torch.angle(X[:, b])[neg] == math.pi).all())
    assert seen > 50
