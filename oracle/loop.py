"""Restatement of the reference denoising loop.  ORACLE - test infrastructure only.

Follows /root/reference/pipeline/pipeline_stable_video_diffusion_trans.py: guidance scale :531-536, loop body
:545-592 (CFG duplication :549, ``scale_model_input`` :550, channel concat :553, UNet call :557-563, per-frame CFG
:578-587, Euler step :592), ``prepare_latents`` scaling :330, ``_get_add_time_ids`` :238-252.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .scheduler import EulerDiscreteOracle


def add_time_ids(fps: int, motion_bucket_id: int, noise_aug_strength: float, batch: int, cfg: bool, dtype):
    """``_get_add_time_ids`` :238-252 (the caller passes fps-1, :463)."""
    ids = torch.tensor([[fps, motion_bucket_id, noise_aug_strength]], dtype=dtype).repeat(batch, 1)
    return torch.cat([ids, ids]) if cfg else ids


def denoise(unet: Callable, scheduler: EulerDiscreteOracle, latents: torch.Tensor, image_latents: torch.Tensor,
            image_embeddings: torch.Tensor, added_time_ids: torch.Tensor, num_inference_steps: int,
            min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
            domain_features: Optional[torch.Tensor] = None, flow_features: Optional[torch.Tensor] = None,
            callback: Optional[Callable] = None, controlnet: Optional[Callable] = None,
            controlnet_condition: Optional[torch.Tensor] = None, controlnet_cond_scale: float = 1.0) -> torch.Tensor:
    """``latents``: unit-variance noise [B,F,4,h,w] (scaled by init_noise_sigma here, :330); ``image_latents``
    [2B or B,F,4,h,w] already CFG-concatenated and repeated over frames (:221,:488); ``image_embeddings`` [2B,1,1024]."""
    do_cfg = max_guidance_scale > 1
    num_frames = latents.shape[1]
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    gs = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0).to(latents.dtype)
    gs = gs.repeat(latents.shape[0], 1)[:, :, None, None, None]
    for i, t in enumerate(scheduler.timesteps):
        x = torch.cat([latents] * 2) if do_cfg else latents
        x = scheduler.scale_model_input(x, t)
        x = torch.cat([x, image_latents], dim=2)
        if controlnet is not None:
            # /root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:582-607
            down, mid = controlnet(x, t, encoder_hidden_states=image_embeddings, controlnet_cond=controlnet_condition,
                                   added_time_ids=added_time_ids, conditioning_scale=controlnet_cond_scale,
                                   guess_mode=False, return_dict=False)
            noise_pred = unet(x, t, encoder_hidden_states=image_embeddings, down_block_additional_residuals=down,
                              mid_block_additional_residual=mid, added_time_ids=added_time_ids, return_dict=False)[0]
        elif domain_features is not None:
            noise_pred = unet(x, t, image_embeddings, domain_features, flow_features,
                              added_time_ids=added_time_ids, return_dict=False)[0]
        else:
            noise_pred = unet(x, t, encoder_hidden_states=image_embeddings, added_time_ids=added_time_ids,
                              return_dict=False)[0]
        if do_cfg:
            uncond, cond = noise_pred.chunk(2)
            noise_pred = uncond + gs * (cond - uncond)
        latents = scheduler.step(noise_pred, t, latents)
        if callback is not None:
            callback(i, t, latents)
    return latents
