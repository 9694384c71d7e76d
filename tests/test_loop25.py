"""The 25-step Euler loop of the headline metric against the reference's own `__call__`
(pipeline_stable_video_diffusion_trans.py:544-640, scheduler utils/scheduling_euler_discrete_karras_fix.py:418-528) run in
fp32 on the CPU of the build container: tests/golden/loop25.safetensors (tiny width) and loop25_c1_realwidth.safetensors
(REAL width, BASELINE.json configs[0] geometry).  Gates are SURVEY.md 8d's: every 5th step and the final latents within
relative L2 <= 5e-2 and cosine >= 0.998; the per-step error curve is printed (and written to gpurun_out/ when that exists)."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from golden.fullres_cases import LOOP25_SEED

REL_GATE, COS_GATE = 5e-2, 0.998


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _cos(a, b):
    a, b = a.float().cpu().flatten().double(), b.float().cpu().flatten().double()
    return float(a @ b / (a.norm() * b.norm()))


def _curve(name, steps, golden):
    cur = [(_rel(s, golden["step_latents"][i]), _cos(s, golden["step_latents"][i])) for i, s in enumerate(steps)]
    print(f"\n{name}: per-step rel L2 " + " ".join(f"{r:.1e}" for r, _ in cur))
    print(f"{name}: per-step 1-cos  " + " ".join(f"{1 - c:.1e}" for _, c in cur))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"loop25_curve_{name}.json"), "w") as f:
            json.dump({"rel_l2": [r for r, _ in cur], "cosine": [c for _, c in cur]}, f)
    return cur


def _gate(name, steps, final, golden):
    assert len(steps) == 25 and golden["step_latents"].shape[0] == 25
    cur = _curve(name, steps, golden)
    for i in list(range(4, 25, 5)) + [24]:
        r, c = cur[i]
        assert r <= REL_GATE and c >= COS_GATE, f"{name} step {i}: rel L2 {r:.3e}, cosine {c:.5f}"
    r, c = _rel(final, golden["final"]), _cos(final, golden["final"])
    assert r <= REL_GATE and c >= COS_GATE, f"{name} final latents: rel L2 {r:.3e}, cosine {c:.5f}"


def _tiny_oracle():
    from oracle import unet as ou
    o = ou.init_weights_(ou.UNetSpatioTemporalConditionControlNetModel(ou.TINY_CONFIG), LOOP25_SEED)
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.half().float())
    return o


def test_oracle_25_step_loop_vs_reference_golden(golden_dir):
    """pins the oracle's loop + scheduler over all 25 sigmas (700 -> 0.002) on the reference's own run"""
    from oracle.loop import denoise
    from oracle.scheduler import EulerDiscreteOracle
    g = load_file(os.path.join(golden_dir, "loop25.safetensors"))
    o = _tiny_oracle()
    ck = float(sum(p.detach().double().abs().sum() for p in o.parameters()))
    assert abs(ck - g["checksum"].item()) <= 1e-9 * ck
    steps = []
    with torch.no_grad():
        out = denoise(o, EulerDiscreteOracle(), g["latents0"], g["image_latents"], g["image_embeddings"],
                      g["added_time_ids"], 25, callback=lambda i, t, l: steps.append(l.clone()))
    for i, s in enumerate(steps):
        assert _rel(s, g["step_latents"][i]) < 1e-3, i
    assert _rel(out, g["final"]) < 1e-3


def _run_hip(model, g, px, steps_out):
    from lkgd_amd.pipeline import StableVideoDiffusionPipeline
    pipe = StableVideoDiffusionPipeline(unet=model)
    return pipe(None, height=px, width=px, num_frames=4, num_inference_steps=25, latents=g["latents0"],
                output_type="latent", image_embeddings=g["image_embeddings"], image_latents=g["image_latents"].half(),
                fps=7, motion_bucket_id=127, noise_aug_strength=0.02,
                callback_on_step_end=lambda p, i, t, kw: (steps_out.append(kw["latents"].clone()), {})[1]).frames


@pytest.mark.gpu
def test_hip_25_step_loop_tiny_vs_reference_golden(golden_dir):
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    g = load_file(os.path.join(golden_dir, "loop25.safetensors"))
    m = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    m.load_state_dict(_tiny_oracle().state_dict())
    steps = []
    final = _run_hip(m.half().to("cuda:0"), g, 64, steps)
    _gate("tiny", steps, final, g)


@pytest.mark.gpu
def test_hip_25_step_loop_real_width_vs_reference_golden(golden_dir, c1_oracle_model, c1_hip_model):
    """the fp16 HIP loop through all 25 steps at the channel widths of the benchmark (1.52 B parameters)"""
    g = load_file(os.path.join(golden_dir, "loop25_c1_realwidth.safetensors"))
    ck = float(sum(p.detach().double().abs().sum() for p in c1_oracle_model.parameters()))
    assert abs(ck - g["checksum"].item()) <= 1e-9 * ck
    steps = []
    final = _run_hip(c1_hip_model, g, 256, steps)
    _gate("realwidth", steps, final, g)
