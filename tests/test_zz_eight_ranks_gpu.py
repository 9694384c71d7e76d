"""The eight-rank layout of BASELINE.json configs[3] against the reference's 14-frame golden, as eight threads of this process
(tests/thread_world.py).  Collected LAST (file name) on purpose: it is the one test whose ranks share a process, so that a surprise
there cannot keep `pytest -x` from running the rest of the suite."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden(golden_dir):
    return load_file(os.path.join(golden_dir, "loop_f14_cfg.safetensors"))


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def test_eight_ranks_cfg_x_4_4_3_3_loop_vs_reference_golden(golden, c1_hip_model, monkeypatch):
    """The layout BASELINE.json configs[3] names - 8 ranks = CFG-parallel x frame slices (4, 4, 3, 3): an UNEVEN split with two
    interior shards per half (halos at both boundaries, all-to-all with uneven rows, padded gathers) - every rank's result
    against the REFERENCE's fp32 run.  Eight GPU processes do not fit this pool's six-process guard, so the ranks run as
    eight THREADS of this process over tests/thread_world.py (a stand-in for the few torch.distributed calls the sharded
    denoiser makes; same stream, barrier-ordered copies).  Every rank replays its recorded launch list from the second Euler
    step on, as a real rank does (the recorder is thread-local since round 5)."""
    import lkgd_amd.dist as ldist
    import lkgd_amd.dist_run as ldist_run
    from lkgd_amd import unet as pu
    from lkgd_amd.pipeline import StableVideoDiffusionPipeline
    from thread_world import ThreadWorld, run_ranks
    world = 8
    dev = torch.device("cuda", 0)
    sd = {k: v.detach() for k, v in c1_hip_model.state_dict().items()}      # device tensors, shared: copied into each rank's model

    def rank_fn(rank):
        with torch.device("meta"):
            m = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig())
        m = m.to(torch.float16).to_empty(device=dev)
        m.load_state_dict(sd, strict=True)
        pipe = StableVideoDiffusionPipeline(unet=m)
        pipe.scheduler.set_timesteps(2)
        runner = ldist_run.DistDenoiser(pipe, world, rank, 14, cfg=True)
        assert runner.plan.splits == (4, 4, 3, 3)
        ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
        lat = (golden["latents0"] * float(pipe.scheduler.init_noise_sigma)).half().to(dev)
        out = runner.denoise(lat, golden["image_latents"].half().to(dev), golden["image_embeddings"].half().to(dev),
                             ids.to(dev), 2, 1.0, 3.0)
        return out.float().cpu()

    tw = ThreadWorld(world)
    monkeypatch.setattr(ldist, "dist", tw)          # lkgd_amd.dist / dist_run say `dist.<call>`: route to this test's world
    monkeypatch.setattr(ldist_run, "dist", tw)
    results = run_ranks(tw, rank_fn)
    for r, out in enumerate(results):
        rel = _rel(out, golden["final"])
        assert rel < 2e-2, f"rank {r} of 8: CFG x (4,4,3,3) sharded 14-frame loop vs the reference: relative L2 {rel:.3e}"
    assert all(torch.equal(results[0], o) for o in results[1:])          # every rank ends with the same gathered latents
