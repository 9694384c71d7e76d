#!/usr/bin/env python3
"""The reference's inference harness for the vanilla SVD pipeline on the MI355X path - same knobs, same call sequence.

Mirrors /root/reference/run_models/run_inference_svd.py:141-242: the ``args`` dict (``pretrained_model_name_or_path``,
``validation_image``, ``output_dir``, ``height``, ``width``, ``seed``, ``batch_size``) :141-153, ``from_pretrained(...,
torch_dtype=float16)`` + ``enable_model_cpu_offload()`` :166-171, the LoRA file whose tensors are copied into the UNet's
state dict by name (``unet.`` prefix stripped, only names that exist) :183-207, the three seeds :216-219, the pipeline call
with ``[validation_image]`` :227-234 and ``save_gifs_side_by_side`` (utils/util.py:791-859) into
``<output_dir>/validation_images``.  Here the knobs are command-line options with the reference's values as defaults;
the same functions are importable (``run(args)``).

    python -m lkgd_amd.run_inference_svd --pretrained_model_name_or_path /path/to/SVD-XT --validation_image img.jpg

Without a checkpoint directory (the GPU boxes have none) ``--random_init_tiny`` builds a tiny random pipeline directory
first and runs the same code path on it (a plumbing check, not a video).
"""
from __future__ import annotations

import argparse
import datetime
import os
import random
from typing import Dict, List, Optional

import numpy as np
import torch

DEFAULTS = {
    "pretrained_model_name_or_path": "/code/weights/SVD-XT",
    "validation_image": "/code/datasets/test_imgs/900.jpg",
    "output_dir": "./output",
    "height": 512,
    "width": 512,
    "seed": 12345,
    "batch_size": 1,
}


def validate_and_convert_image(image):
    """utils/util.py:861-884"""
    from PIL import Image
    if image is None:
        return None
    if isinstance(image, torch.Tensor):
        if image.ndim == 3 and image.shape[0] in (1, 3):
            if image.shape[0] == 1:
                image = image.repeat(3, 1, 1)
            return Image.fromarray(image.mul(255).clamp(0, 255).byte().permute(1, 2, 0).cpu().numpy())
        return None
    return image if isinstance(image, Image.Image) else None


def save_gifs_side_by_side(videos, output_folder: str, global_step: str = "") -> str:
    """utils/util.py:791-859: one GIF per video (100 ms per frame), then the frames of all GIFs stacked vertically into
    ``combined_frames_<step>_<timestamp>.gif``; the per-video temporaries are removed"""
    from PIL import Image
    os.makedirs(output_folder, exist_ok=True)
    ts = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    paths = []
    for idx, frames in enumerate(videos):
        if isinstance(frames[0], list):
            frames = frames[0]
        pil = [p for p in (validate_and_convert_image(f) for f in frames) if p is not None]
        path = os.path.join(output_folder, f"temp_{idx}_{ts}.gif")
        if pil:
            pil[0].save(path, save_all=True, append_images=pil[1:], loop=0, duration=100)
            paths.append(path)
    gifs = [Image.open(p) for p in paths]
    n = min(g.n_frames for g in gifs)
    out_frames = []
    for i in range(n):
        combined = None
        for g in gifs:
            g.seek(i)
            fr = g.copy().convert("RGB")
            if combined is None:
                combined = fr
            else:
                dst = Image.new("RGB", (max(combined.width, fr.width), combined.height + fr.height))
                dst.paste(combined, (0, 0))
                dst.paste(fr, (0, combined.height))
                combined = dst
        out_frames.append(combined)
    out = os.path.join(output_folder, f"combined_frames_{global_step}_{ts}.gif")
    out_frames[0].save(out, save_all=True, append_images=out_frames[1:], loop=0, duration=100)
    for g in gifs:
        g.close()
    for p in paths:
        os.remove(p)
    return out


def update_unet_from_lora_file(unet, lora_weights_path: str) -> List[str]:
    """run_inference_svd.py:183-207: every tensor of the safetensors file whose name (minus a leading ``unet.``) is a key
    of the UNet's state dict overwrites that parameter; everything else is skipped.  Returns the updated names."""
    from safetensors import safe_open
    sd = unet.state_dict()
    updated = []
    with safe_open(lora_weights_path, framework="pt", device="cpu") as f:
        for key in f.keys():
            name = key.replace("unet.", "", 1)
            if name in sd:
                sd[name].copy_(f.get_tensor(key))
                updated.append(name)
    unet.load_state_dict(sd)
    return updated


def make_tiny_pipeline_dir(path: str, seed: int = 0) -> str:
    """a random-init pipeline directory in the diffusers layout with tiny components (plumbing checks on boxes without
    checkpoints): lkgd_amd UNet / VAE / scheduler + a one-layer CLIP vision tower in transformers' on-disk layout"""
    import json
    from .clip import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModelWithProjection
    from . import unet as pu
    from . import vae as pv
    from .scheduler import EulerDiscreteScheduler
    torch.manual_seed(seed)
    cfg = pu.UNetConfig(sample_size=8, block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2),
                        addition_time_embed_dim=64, projection_class_embeddings_input_dim=192, num_frames=4)
    u = pu.UNetSpatioTemporalConditionControlNetModel(cfg)
    pu.init_synthetic_weights_(u, seed)
    u.half().save_pretrained(os.path.join(path, "unet"))
    v = pv.AutoencoderKLTemporalDecoder(pv.VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=64))
    pu.init_synthetic_weights_(v, seed + 1)
    v.half().save_pretrained(os.path.join(path, "vae"))
    EulerDiscreteScheduler.from_svd_config().save_pretrained(os.path.join(path, "scheduler"))
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                                                          num_attention_heads=2, image_size=224, patch_size=32,
                                                          projection_dim=1024, hidden_act="quick_gelu"))
    pu.init_synthetic_weights_(clip, seed + 2)
    with torch.no_grad():
        clip.vision_model.embeddings.position_embedding.weight.normal_(0, 0.02)
        clip.vision_model.embeddings.class_embedding.normal_(0, 0.02)
    clip.half().save_pretrained(os.path.join(path, "image_encoder"))
    os.makedirs(os.path.join(path, "feature_extractor"), exist_ok=True)
    with open(os.path.join(path, "feature_extractor", "preprocessor_config.json"), "w") as fh:
        json.dump({"image_processor_type": "CLIPImageProcessor", "do_normalize": True, "image_mean": list(CLIPImageProcessor.OPENAI_MEAN),
                   "image_std": list(CLIPImageProcessor.OPENAI_STD), "size": {"shortest_edge": 224},
                   "crop_size": {"height": 224, "width": 224}}, fh, indent=2)
    return path


def run(args: Dict, lora_weights_path: Optional[str] = None, pipeline_kwargs: Optional[Dict] = None):
    """returns (video_frames, path of the combined GIF)"""
    from PIL import Image
    from .pipeline import StableVideoDiffusionPipeline
    validation_image = Image.open(args["validation_image"]).convert("RGB")
    pipeline = StableVideoDiffusionPipeline.from_pretrained(args["pretrained_model_name_or_path"], torch_dtype=torch.float16,
                                                            low_cpu_mem_usage=False, device_map=None)
    pipeline.enable_model_cpu_offload()
    if lora_weights_path:
        names = update_unet_from_lora_file(pipeline.unet, lora_weights_path)
        print(f"updated {len(names)} UNet parameters from {lora_weights_path}")
    val_save_dir = os.path.join(args["output_dir"], "validation_images")
    os.makedirs(val_save_dir, exist_ok=True)
    seed = args["seed"]
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    video_frames = pipeline([validation_image] * int(args.get("batch_size", 1)), width=args["width"], height=args["height"],
                            **(pipeline_kwargs or {})).frames
    return video_frames, save_gifs_side_by_side(video_frames, val_save_dir)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    for k, v in DEFAULTS.items():
        ap.add_argument("--" + k, type=type(v), default=v)
    ap.add_argument("--lora_weights_path", default=None)
    ap.add_argument("--num_frames", type=int, default=None)
    ap.add_argument("--num_inference_steps", type=int, default=25)
    ap.add_argument("--decode_chunk_size", type=int, default=None)
    ap.add_argument("--motion_bucket_id", type=int, default=127)
    ap.add_argument("--random_init_tiny", action="store_true",
                    help="build a tiny random pipeline directory (and a random validation image) under output_dir and run on it")
    a = ap.parse_args(argv)
    args = {k: getattr(a, k) for k in DEFAULTS}
    if a.random_init_tiny:
        from PIL import Image
        os.makedirs(args["output_dir"], exist_ok=True)
        args["pretrained_model_name_or_path"] = make_tiny_pipeline_dir(os.path.join(args["output_dir"], "tiny_svd"))
        args["validation_image"] = os.path.join(args["output_dir"], "validation.png")
        Image.fromarray(np.random.RandomState(0).randint(0, 256, (96, 128, 3), dtype=np.uint8)).save(args["validation_image"])
        args["height"], args["width"] = 64, 64
    kw = {"num_inference_steps": a.num_inference_steps, "motion_bucket_id": a.motion_bucket_id}
    if a.num_frames is not None:
        kw["num_frames"] = a.num_frames
    if a.decode_chunk_size is not None:
        kw["decode_chunk_size"] = a.decode_chunk_size
    frames, gif = run(args, a.lora_weights_path, kw)
    print(f"{len(frames)} video(s) x {len(frames[0])} frames -> {gif}")


if __name__ == "__main__":
    main()
