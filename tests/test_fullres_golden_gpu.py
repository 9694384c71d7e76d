"""BASELINE.json configs[1] / configs[2] at the FULL latent resolution (72 x 128, S = 9216 spatial tokens per frame) and the
REAL width, against ONE forward of the reference's own top-level UNets run in fp32 on the CPU of the build container
(tests/golden/make_goldens.py::gen_unet_fullres / gen_unet_fullres_lk; CFG 2 x 2 frames -> N = 4 frame-images):

* unet_fullres.safetensors     <- models/unet_spatio_temporal_condition_controlnet.py:358-508 (stock signature)
* unet_fullres_f14.safetensors <- the same stock forward at the headline geometry itself: CFG 2 x 14 frames (round 4)
* unet_fullres_lk.safetensors  <- models/unet_spatio_temporal_condition.py:448-693 (domain / flow features) without and
                                  with the patch_FSM hook (patch/patch_FSM.py:380-441) active in all 16 spatial blocks

These are the S = 9216 attention / 36 864-row GEMM code paths the benchmark runs (256x320 tiles, row-panel kernel,
16-wave attention workgroups) - the per-kernel tests only see them through row samples.  Weights, inputs, tracks and hook
weights are regenerated from seeds (tests/golden/fullres_cases.py); the stored checksum proves they are the ones the
reference ran with.  Gates: relative L2 <= 1e-2, max-abs <= 5e-2 on O(1) outputs."""
import os

import pytest
import torch
from safetensors.torch import load_file

from golden.fullres_cases import FULLRES_F14_SEED, FULLRES_LK_SEED, fullres_inputs, fullres_tracks, seed_conv_fuse_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hip_model(lk, seed, checksum):
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    ocls = ou.UNetSpatioTemporalConditionModel if lk else ou.UNetSpatioTemporalConditionControlNetModel
    pcls = pu.UNetSpatioTemporalConditionModel if lk else pu.UNetSpatioTemporalConditionControlNetModel
    o = ou.init_weights_(ocls(ou.SVD_CONFIG), seed)
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.half().float())
    ck = float(sum(p.detach().double().abs().sum() for p in o.parameters()))
    assert abs(ck - checksum) <= 1e-9 * ck, "regenerated weights differ from the ones the reference ran with"
    with torch.device("meta"):
        m = pcls(pu.UNetConfig())
    m = m.to_empty(device="cpu")
    m.load_state_dict(o.state_dict(), strict=True)
    del o
    return m.half().to(DEV)


def _gate(got, ref, what):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    mx = (got - ref).abs().max().item()
    print(f"\n{what}: rel L2 {rel:.3e}, max abs {mx:.3e} (ref std {ref.std():.3f})")
    assert rel <= 1e-2 and mx <= 5e-2, f"{what}: rel L2 {rel:.3e}, max abs {mx:.3e}"


def test_stock_unet_full_resolution_vs_reference_golden(golden_dir, c1_oracle_model, c1_hip_model):
    g = load_file(os.path.join(golden_dir, "unet_fullres.safetensors"))
    ck = float(sum(p.detach().double().abs().sum() for p in c1_oracle_model.parameters()))
    assert abs(ck - g["checksum"].item()) <= 1e-9 * ck, "regenerated weights differ from the ones the reference ran with"
    m = c1_hip_model                      # the real-width weights shared by the c1 / 25-step / full-resolution fixtures
    i = fullres_inputs()
    out = m(i["sample"].to(DEV), i["t"].to(DEV), i["enc"].to(DEV), added_time_ids=i["ids"].to(DEV), return_dict=False)[0]
    _gate(out, g["out"], "stock UNet @ 72x128")
    assert abs(out.double().sum().item() - g["out_sum"].item()) <= 2e-3 * g["out_abs_sum"].item()


def test_stock_unet_headline_geometry_vs_reference_golden(golden_dir, c1_oracle_model, c1_hip_model):
    """The headline workload ITSELF (round 4): ONE forward at CFG 2 x 14 frames x 72 x 128 - every kernel at the very shapes
    bench.py times (28 frame-images, S = 9216 spatial tokens, 14-frame temporal attention / Conv3d / GroupNorm at 9216 pixels)
    - against the reference's own `forward` (unet_spatio_temporal_condition_controlnet.py:358-508) run in fp32 on the CPU
    (make_goldens.py::gen_unet_fullres_f14; the fixture stores the result rounded to fp16 and fp64 sums of the fp32 result)."""
    g = load_file(os.path.join(golden_dir, "unet_fullres_f14.safetensors"))
    ck = float(sum(p.detach().double().abs().sum() for p in c1_oracle_model.parameters()))
    assert abs(ck - g["checksum"].item()) <= 1e-9 * ck, "regenerated weights differ from the ones the reference ran with"
    i = fullres_inputs(seed=FULLRES_F14_SEED, frames=14)
    out = c1_hip_model(i["sample"].to(DEV), i["t"].to(DEV), i["enc"].to(DEV), added_time_ids=i["ids"].to(DEV),
                       return_dict=False)[0]
    assert out.shape == (2, 14, 4, 72, 128)
    _gate(out, g["out_f16"].float(), "stock UNet @ 2 x 14 x 72 x 128")
    assert abs(out.double().sum().item() - g["out_sum"].item()) <= 2e-3 * g["out_abs_sum"].item()


def test_lk_unet_with_fsm_hook_full_resolution_vs_reference_golden(golden_dir):
    from lkgd_amd import patch_FSM
    g = load_file(os.path.join(golden_dir, "unet_fullres_lk.safetensors"))
    m = _hip_model(True, FULLRES_LK_SEED, g["checksum"].item())
    i = fullres_inputs(lk=True)
    args = (i["sample"].to(DEV), i["t"].to(DEV), i["enc"].to(DEV), i["domain"].to(DEV), i["flow"].to(DEV))
    out = m(*args, added_time_ids=i["ids"].to(DEV), return_dict=False)[0]
    _gate(out, g["lk_nohook"], "LK UNet @ 72x128")
    track, res = fullres_tracks()
    patch_FSM.apply_patch(m, with_spatial_block=True, with_temporal_block=False)
    patch_FSM.initialize_joint_layers(m)
    seed_conv_fuse_([b for _, b in m.named_modules() if hasattr(b, "conv_fuse")])
    m.invalidate()
    patch_FSM.update_patch(m, track=tuple(t.to(DEV) for t in track), track_res=res)
    patch_FSM.set_joint_attention(m, True)
    out = m(*args, added_time_ids=i["ids"].to(DEV), return_dict=False)[0]
    _gate(out, g["lk_fsm"], "LK UNet + FSM hook @ 72x128")
    delta = (g["lk_fsm"] - g["lk_nohook"]).norm() / g["lk_nohook"].norm()
    assert delta > 2e-2, "the fixture's hook does not change the output enough to test anything"
