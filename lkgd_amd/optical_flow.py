"""Flow-latent (un)normalisation of the reference's optical-flow pipelines.

Mirrors /root/reference/utils/optical_flow.py:62-77 (``FLOW_LATENT_MEAN`` / ``FLOW_LATENT_STD``,
``optical_flow_latent_normalize`` / ``optical_flow_latent_unnormalize``), used at
/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet_flow.py:641 - the one place where that pipeline differs from
the vanilla loop as the reference runs it (its ControlNet call is commented out, :548-553,:590-607).  One affine map over the
1 MB latent tensor of a clip, once per clip: plain tensor arithmetic in the latents' dtype, as the reference does it.
"""
import torch

FLOW_CLIP_MAX = 50
FLOW_NORM_CLIP_MAX = (2 * FLOW_CLIP_MAX ** 2) ** 0.5
FLOW_LATENT_MEAN = 0.5020191669464111
FLOW_LATENT_STD = 1.2818458080291748


def optical_flow_latent_normalize(tensor: torch.Tensor, scale=1) -> torch.Tensor:
    dt = tensor.dtype
    t = tensor.to(torch.float32) * scale
    return (((t - FLOW_LATENT_MEAN) / FLOW_LATENT_STD) / scale).to(dt)


def optical_flow_latent_unnormalize(tensor: torch.Tensor) -> torch.Tensor:
    return tensor * FLOW_LATENT_STD + FLOW_LATENT_MEAN
