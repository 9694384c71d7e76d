"""StableVideoDiffusionPipeline with the reference's ``__call__`` signature; the denoising loop runs on the HIP path.

Mirrors /root/reference/pipeline/pipeline_stable_video_diffusion_trans.py: ``__call__`` :352-656 (signature :352-372,
loop :545-640), ``_encode_image`` :157-203, ``_encode_vae_image`` :205-226, ``_get_add_time_ids`` :228-254,
``decode_latents`` :256-283, ``prepare_latents`` :299-331, ``check_inputs`` :285-297.

Scope: the hot path is the loop body.  ``denoise()`` is that loop: per step ONE glue kernel
(CFG duplicate + scale_model_input + channel concat -> channels-last tokens), the UNet forward on tokens, ONE glue
kernel (per-frame CFG + v-prediction + Euler update).  No host<->device sync inside the loop (sigma tables live on the
host).  CLIP / VAE are boundary stages: any modules with the diffusers interface may be passed in and are called as the
reference calls them; they are not re-implemented here (8f rank 2).
The LKGD conditioning (domain / flow ViT logits) is passed as ``domain_features`` / ``flow_features`` (intended wiring:
CogVideo-main/finetune/models/cogvideox_i2v/pipeline_cogvideox_image2video.py:794-799,849-859) and fused ONCE per clip.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Union

import torch

from . import ops
from . import replay as _replay
from . import trace as _trace
from ._lib import LkgdHipError
from .image_ops import resize_with_antialiasing
from .image_processor import VaeImageProcessor, tensor2vid
from .scheduler import EulerDiscreteScheduler


@dataclass
class StableVideoDiffusionPipelineOutput:
    frames: Union[torch.Tensor, list]


def _randn_tensor(shape, generator, device, dtype):
    """diffusers' `randn_tensor` [EXT utils/torch_utils.py]: a CPU generator serves a GPU target by sampling on the CPU (so
    seeds reproduce across devices); a list of generators samples per batch entry"""
    if isinstance(generator, (list, tuple)):
        if len(generator) == 1:
            generator = generator[0]
        else:
            per = (1,) + tuple(shape[1:])
            return torch.cat([_randn_tensor(per, g, device, dtype) for g in generator], dim=0)
    gdev = generator.device if isinstance(generator, torch.Generator) else torch.device(device)
    if gdev.type != torch.device(device).type and gdev.type == "cpu":
        return torch.randn(tuple(shape), generator=generator, device="cpu", dtype=dtype).to(device)
    return torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)


def _append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


class StableVideoDiffusionPipeline:
    model_cpu_offload_seq = "image_encoder->unet->vae"
    _callback_tensor_inputs = ["latents"]

    def __init__(self, vae=None, image_encoder=None, unet=None, scheduler: Optional[EulerDiscreteScheduler] = None,
                 feature_extractor=None, controlnet=None):
        self.vae, self.image_encoder, self.unet = vae, image_encoder, unet
        #: optional lkgd_amd.controlnet.ControlNetSDVModel (reference pipeline_stable_video_diffusion_controlnet.py:156-178)
        self.controlnet = controlnet
        self.scheduler = scheduler if scheduler is not None else EulerDiscreteScheduler.from_svd_config()
        self.feature_extractor = feature_extractor
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)       # reference :150
        self._guidance_scale = None
        self._num_timesteps = 0
        #: replay the per-step UNet forward from a captured HIP graph (one launch instead of ~1000): the forward is
        #: shape-static across the Euler steps, only the timestep scalar and the input tokens change (static buffers)
        self.use_hip_graph = False   # opt-in: eager launches are already hidden behind GPU work on one GPU, and bench.py times
                                     # individual GEMM launches with events, which a graph replay cannot expose
        self._graph = None
        #: replay the step's forward from a recorded launch list (denoise(); LKGD_NO_REPLAY=1 = walk the modules every step)
        self.use_replay = __import__("os").environ.get("LKGD_NO_REPLAY", "0") != "1"
        #: private allocator pools the recorded forwards allocate from (lkgd_amd/replay.py): the plan's working set is the eager
        #: peak of one forward, and the blocks stay cached here between calls; ``release_arena()`` returns them to the driver
        self._arenas = _replay.ArenaSet()

    # ---- loading (DiffusionPipeline.from_pretrained [EXT]; call sites run_models/run_inference_svd.py:166-168,
    #      utils/util.py:536) --------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, torch_dtype=torch.float16, variant: Optional[str] = None,
                        unet=None, vae=None, image_encoder=None, feature_extractor=None, scheduler=None, controlnet=None,
                        unet_class=None, device: Optional[str] = "cuda", **_ignored):
        """a local diffusers pipeline directory (``unet/ vae/ image_encoder/ feature_extractor/ scheduler/``).  The UNet,
        the VAE, the scheduler and the CLIP image encoder (`image_encoder/` -> lkgd_amd.clip, transformers' parameter names;
        `feature_extractor/` -> its mean / std normalisation) are lkgd_amd's own classes: no diffusers, no transformers.  Components passed
        as keywords are used as given (``unet=...`` as utils/util.py:607-616 does).  hub / offload keywords are accepted
        and ignored (``low_cpu_mem_usage``, ``device_map``, ``local_files_only`` ...)."""
        import os
        from .unet import UNetSpatioTemporalConditionControlNetModel
        from .vae import AutoencoderKLTemporalDecoder
        root = pretrained_model_name_or_path
        if not os.path.isdir(root):
            raise OSError(f"{root} is not a local pipeline directory (there is no hub access)")

        def has(sub, f):
            return os.path.exists(os.path.join(root, sub, f))
        if unet is None:
            unet = (unet_class or UNetSpatioTemporalConditionControlNetModel).from_pretrained(
                root, subfolder="unet", torch_dtype=torch_dtype, variant=variant)
        if vae is None and has("vae", "config.json"):
            vae = AutoencoderKLTemporalDecoder.from_pretrained(root, subfolder="vae", torch_dtype=torch_dtype, variant=variant)
        if scheduler is None and has("scheduler", "scheduler_config.json"):
            scheduler = EulerDiscreteScheduler.from_pretrained(root, subfolder="scheduler")
        if image_encoder is None and has("image_encoder", "config.json"):
            from .clip import CLIPVisionModelWithProjection          # transformers' class name and checkpoint layout, HIP forward
            image_encoder = CLIPVisionModelWithProjection.from_pretrained(root, subfolder="image_encoder",
                                                                          torch_dtype=torch_dtype, variant=variant)
        if feature_extractor is None and has("feature_extractor", "preprocessor_config.json"):
            from .clip import CLIPImageProcessor
            feature_extractor = CLIPImageProcessor.from_pretrained(root, subfolder="feature_extractor")
        pipe = cls(vae=vae, image_encoder=image_encoder, unet=unet, scheduler=scheduler,
                   feature_extractor=feature_extractor, controlnet=controlnet)
        return pipe.to(device) if device is not None else pipe

    def to(self, device=None, dtype=None):
        """``DiffusionPipeline.to`` [EXT]: every module component to the device (the HIP models run on cuda only)"""
        for name in ("unet", "vae", "image_encoder", "controlnet"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "to"):
                m = m.to(device=device, dtype=dtype) if dtype is not None else m.to(device)
                setattr(self, name, m)
        return self

    def save_pretrained(self, save_directory: str, **kw):
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        index = {"_class_name": "StableVideoDiffusionPipeline", "_lkgd_amd": True}
        for name in ("unet", "vae", "image_encoder", "feature_extractor", "scheduler"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "save_pretrained"):
                m.save_pretrained(os.path.join(save_directory, name))
                index[name] = [type(m).__module__.split(".")[0], type(m).__name__]
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)

    # ---- reference helpers -------------------------------------------------------------------------------------
    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        if isinstance(self.guidance_scale, (int, float)):
            return self.guidance_scale > 1
        return self.guidance_scale.max() > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def _execution_device(self):
        return self.unet.device

    def enable_model_cpu_offload(self, *a, **k):
        return None   # a memory knob of the reference (run_inference_svd.py:171); nothing to offload at 288 GB

    def check_inputs(self, image, height, width):
        import PIL.Image
        if not isinstance(image, (torch.Tensor, PIL.Image.Image, list)):
            raise ValueError("`image` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or "
                             f"`List[PIL.Image.Image]` but is {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, batch_size, num_videos_per_prompt,
                          do_classifier_free_guidance):
        add_time_ids = [fps, motion_bucket_id, noise_aug_strength]
        passed = self.unet.config.addition_time_embed_dim * len(add_time_ids)
        expected = self.unet.add_embedding.linear_1.in_features
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of "
                             f"{passed} was created. The model has an incorrect config.")
        ids = torch.tensor([add_time_ids], dtype=dtype).repeat(batch_size * num_videos_per_prompt, 1)
        if do_classifier_free_guidance:
            ids = torch.cat([ids, ids])
        return ids

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_frames, num_channels_latents // 2, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}.")
        if latents is None:
            latents = _randn_tensor(shape, generator, device, dtype)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return ops.scale(latents, float(self.scheduler.init_noise_sigma))

    def _encode_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        """boundary stage (CLIP), reference :157-203; the image encoder itself is the caller's module"""
        if self.image_encoder is None:
            raise LkgdHipError("no image_encoder given: pass `image_embeddings=` to __call__ or call denoise()")
        if not isinstance(image, torch.Tensor):
            # reference :166-175: PIL -> [0,1] tensor, normalise, anti-aliased resize to CLIP's 224x224, un-normalise
            image = self.image_processor.numpy_to_pt(self.image_processor.pil_to_numpy(image))
            image = image * 2.0 - 1.0
            image = resize_with_antialiasing(image, (224, 224))
            image = (image + 1.0) / 2.0
        dtype = next(self.image_encoder.parameters()).dtype
        image = self.feature_extractor(images=image, do_normalize=True, do_center_crop=False, do_resize=False,
                                       do_rescale=False, return_tensors="pt").pixel_values
        emb = self.image_encoder(image.to(device=device, dtype=dtype)).image_embeds.unsqueeze(1)
        bs, seq, _ = emb.shape
        emb = emb.repeat(1, num_videos_per_prompt, 1).view(bs * num_videos_per_prompt, seq, -1)
        if do_classifier_free_guidance:
            emb = torch.cat([torch.zeros_like(emb), emb])
        return emb

    def _encode_vae_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        if self.vae is None:
            raise LkgdHipError("no vae given: pass `image_latents=` to __call__ or call denoise()")
        lat = self.vae.encode(image.to(device=device)).latent_dist.mode()
        if do_classifier_free_guidance:
            lat = torch.cat([torch.zeros_like(lat), lat])
        return lat.repeat(num_videos_per_prompt, 1, 1, 1)

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        if self.vae is None:
            raise LkgdHipError("no vae given: use output_type='latent'")
        latents = latents.flatten(0, 1)
        latents = 1 / self.vae.config.scaling_factor * latents
        frames = []
        for i in range(0, latents.shape[0], decode_chunk_size):
            n_in = latents[i:i + decode_chunk_size].shape[0]
            frames.append(self.vae.decode(latents[i:i + decode_chunk_size], num_frames=n_in).sample)
        frames = torch.cat(frames, dim=0)
        frames = frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4)
        return frames.float()

    # ---- the hot path ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, image_latents: torch.Tensor, image_embeddings: torch.Tensor,
                added_time_ids: torch.Tensor, num_inference_steps: int = 25, min_guidance_scale: float = 1.0,
                max_guidance_scale: float = 3.0, domain_features: Optional[torch.Tensor] = None,
                flow_features: Optional[torch.Tensor] = None, callback_on_step_end: Optional[Callable] = None,
                callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                controlnet_condition: Optional[torch.Tensor] = None, controlnet_cond_scale: float = 1.0) -> torch.Tensor:
        """Reference loop :503-640.  ``latents`` [B,F,4,h,w] already scaled by init_noise_sigma (fp16 or fp32, updated
        in place and returned); ``image_latents`` [cfg*B,F,4,h,w] fp16; ``image_embeddings`` [cfg*B,1,1024].
        ``controlnet_condition`` [cfg*B,F,3,8h,8w] (already preprocessed and duplicated for CFG) runs ``self.controlnet``
        before the UNet every step and feeds its residuals in (pipeline_stable_video_diffusion_controlnet.py:582-607)."""
        unet, sch = self.unet, self.scheduler
        dev = unet.device
        B, F, _, H, W = latents.shape
        cfg = 2 if max_guidance_scale > 1 else 1
        if image_latents.shape[0] != cfg * B or image_embeddings.shape[0] != cfg * B:
            raise ValueError("image_latents / image_embeddings must carry cfg*batch entries (uncond first)")
        latents = latents.to(dev).contiguous()
        image_latents = image_latents.to(device=dev, dtype=torch.float16).contiguous()
        sch.set_timesteps(num_inference_steps, device=None)
        self._num_timesteps = len(sch.timesteps_host)
        guidance = torch.linspace(min_guidance_scale, max_guidance_scale, F, dtype=torch.float32)
        self._guidance_scale = _append_dims(guidance.unsqueeze(0).repeat(B, 1), latents.ndim)
        guidance_dev = guidance.to(dev)
        enc = image_embeddings.to(dev)
        if domain_features is not None:
            if not hasattr(unet, "fused_embedding"):
                raise LkgdHipError("domain/flow features need the LKGD UNet (UNetSpatioTemporalConditionModel)")
            enc = unet.fused_embedding(enc, domain_features.to(dev), flow_features.to(dev))   # once per clip
        ids = added_time_ids.to(dev)
        vpred = sch.config.prediction_type == "v_prediction"
        from . import patch as _patch
        _patch.set_joint_attention(unet, enable=True)           # reference :555 (no-op unless the model is patched)
        ctrl = None
        if controlnet_condition is not None:
            if self.controlnet is None:
                raise LkgdHipError("controlnet_condition given but the pipeline has no controlnet")
            if controlnet_condition.shape[0] != cfg * B:
                raise ValueError("controlnet_condition must carry cfg*batch entries")
            ctrl = controlnet_condition.to(device=dev, dtype=torch.float16).contiguous()
        fwd = self._graphed_forward(cfg * B, F, H, W, enc, ids) if (self.use_hip_graph and ctrl is None) else None
        # The step's forward is shape-static over the Euler steps: the first step runs for real and is RECORDED as a flat list of
        # C-ABI launches (lkgd_amd/replay.py), the other 24 replay it between in-place updates of its inputs (token buffer,
        # timestep) - bit-identical, and the Python module walk (64 ms of host time per forward against 90 ms of device time)
        # stops leaving the queue dry at the 18x32 / 9x16 levels: +1.1 % frames/s on one GPU (profiles/r05_bench_pair_replay.txt).
        # LKGD_NO_REPLAY=1 walks the modules every step.
        use_replay = fwd is None and self.use_replay
        if use_replay:
            tok_buf = torch.empty(cfg * B * F * H * W, 8, dtype=torch.float16, device=dev)
            t_dev = torch.zeros(cfg * B, dtype=torch.float32, device=dev)
            enc_r, ids_r = enc.to(torch.float16).contiguous(), ids.to(torch.float32).contiguous()
            if ctrl is not None:
                self.controlnet.prepare()
                self.controlnet._cond_tokens(ctrl, cfg * B, F, H, W)       # once per clip, outside the recorded forward

            def forward_static():
                down = mid = None
                if ctrl is not None:
                    down, mid, _ = self.controlnet.forward_tokens(tok_buf, cfg * B, F, H, W, t_dev, enc_r, ids_r, ctrl,
                                                                  controlnet_cond_scale)
                return unet.forward_tokens(tok_buf, cfg * B, F, H, W, t_dev, enc_r, ids_r, down, mid)[0]
        timers = _trace.StepTimers() if _trace.STEP_TIMERS else None      # LKGD_STEP_TIMERS=1: device ms per Euler step
        self.last_step_timers = timers
        recorded = None
        try:
            for i, t in enumerate(sch.timesteps_host):
                sigma, sigma_next = sch.sigmas_host[i], sch.sigmas_host[i + 1]
                _trace.push(f"euler_step_{i}")                                  # roctx range (LKGD_ROCTX=1), else a no-op
                t_ev = timers.start() if timers is not None else None
                if fwd is not None:
                    ops.prepare_unet_input(latents, image_latents, cfg, sigma, out=fwd.tok)
                    noise_tok = fwd.run(t)
                elif use_replay:
                    ops.prepare_unet_input(latents, image_latents, cfg, sigma, out=tok_buf)
                    t_dev.fill_(float(t))
                    if recorded is not None:
                        noise_tok = recorded.run(ops.GEMM_EVENTS)
                    else:
                        with _replay.record(self._arenas.take(dev, (cfg * B, F, H, W, ctrl is not None, id(unet._pk)))) as recorded:
                            recorded.result = forward_static()
                        noise_tok = recorded.result
                else:
                    tok = ops.prepare_unet_input(latents, image_latents, cfg, sigma)
                    down = mid = None
                    if ctrl is not None:      # residuals stay channels-last token matrices between the two models
                        down, mid, _ = self.controlnet.forward_tokens(tok, cfg * B, F, H, W, t, enc, ids, ctrl,
                                                                      controlnet_cond_scale)
                    noise_tok, _ = unet.forward_tokens(tok, cfg * B, F, H, W, t, enc, ids, down, mid)
                ops.cfg_euler_step(noise_tok, latents, guidance_dev, cfg, sigma, sigma_next, v_prediction=vpred)
                if timers is not None:
                    timers.stop(t_ev)
                _trace.pop()
                if callback_on_step_end is not None:
                    kw = {k: {"latents": latents}[k] for k in callback_on_step_end_tensor_inputs}
                    out = callback_on_step_end(self, i, t, kw)
                    latents = out.pop("latents", latents) if isinstance(out, dict) else latents
        finally:
            # the recorded forward's activations go back to the allocator whatever ended the loop (a callback's exception, a
            # launch error, KeyboardInterrupt): the plan and its recording stand-in reference each other
            if recorded is not None:
                recorded.release()
        sch._step_index = num_inference_steps
        if timers is not None:
            timers.finish()
        return latents

    def release_arena(self) -> None:
        """return the recorded forwards' working-set pool(s) to the driver (they are kept between calls otherwise)"""
        self._arenas.clear()

    def arena_reserved_bytes(self) -> int:
        """device memory the recorded forward's private pool holds (the plan's working set)"""
        return self._arenas.reserved_bytes()

    def _latents_for_decode(self, latents: torch.Tensor) -> torch.Tensor:
        """hook between the loop and the VAE decode (identity here; the flow pipeline un-normalises)"""
        return latents

    def _graphed_forward(self, Bc: int, F: int, H: int, W: int, enc: torch.Tensor, ids: torch.Tensor):
        """HIP-graph replay of ``unet.forward_tokens`` for fixed shapes / conditioning.  Static buffers: input tokens,
        timestep scalar; the captured kernels are exactly the ones the eager path launches (same stream semantics:
        everything in lkgd_amd/csrc is capturable - no allocation or sync inside).  Re-captured when shapes, weights or
        the conditioning tensors change."""
        unet = self.unet
        key = (Bc, F, H, W, enc.data_ptr(), enc._version, ids.data_ptr(), ids._version, id(unet._pk),
               getattr(unet, "_joint_attn_mask", None) is not None)
        g = self._graph
        if g is not None and g.key == key:
            return g
        dev = unet.device

        class _G:
            pass
        g = _G()
        g.key = key
        g.tok = torch.empty(Bc * F * H * W, 8, dtype=torch.float16, device=dev)
        g.t = torch.zeros(1, dtype=torch.float32, device=dev)
        g.tok.zero_()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                    # warm-up on a side stream (allocator pools, lazy caches)
            unet.forward_tokens(g.tok, Bc, F, H, W, g.t, enc, ids)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g.out, _ = unet.forward_tokens(g.tok, Bc, F, H, W, g.t, enc, ids)
        g.graph = graph

        def run(t_value: float, _g=g):
            _g.t.fill_(float(t_value))
            _g.graph.replay()
            return _g.out
        g.run = run
        self._graph = g
        return g

    @torch.no_grad()
    def __call__(self, image, height: int = 576, width: int = 1024, num_frames: Optional[int] = None,
                 num_inference_steps: int = 25, min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
                 fps: int = 7, motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1,
                 generator=None, latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "pil",
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], return_dict: bool = True,
                 # extensions (keyword-only in practice; defaults keep the reference call sites unchanged)
                 image_embeddings: Optional[torch.Tensor] = None, image_latents: Optional[torch.Tensor] = None,
                 domain_features: Optional[torch.Tensor] = None, flow_features: Optional[torch.Tensor] = None,
                 # pipeline_stable_video_diffusion_controlnet.py:356-380: a [F,3,H,W] (or [1,F,3,H,W]) tensor in [0,1]
                 controlnet_condition: Optional[torch.Tensor] = None, controlnet_cond_scale: float = 1.0):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        if image is not None:
            self.check_inputs(image, height, width)
        if isinstance(image, torch.Tensor):
            batch_size = image.shape[0]
        elif isinstance(image, list):
            batch_size = len(image)
        elif image is None:
            batch_size = image_latents.shape[0] // (2 if max_guidance_scale > 1 else 1)
        else:
            batch_size = 1
        device = self._execution_device
        self._guidance_scale = max_guidance_scale
        cfg = self.do_classifier_free_guidance
        if image_embeddings is None:
            image_embeddings = self._encode_image(image, device, num_videos_per_prompt, cfg)
        fps = fps - 1
        if image_latents is None:
            img = self.image_processor.preprocess(image, height=height, width=width).to(device)      # reference :466
            noise = _randn_tensor(img.shape, generator, device, img.dtype)      # drawn for the execution device (:467)
            img = img + noise_aug_strength * noise
            # reference :470-484: an fp16 VAE with `force_upcast` encodes in fp32 and is cast back right after (so the
            # decode at the end runs in fp16 again, :643-645).  The HIP VAE computes fp16 activations with fp32 accumulation
            # whatever the module dtype says (lkgd_amd/vae.py), so casting it there and back would only re-pack its weights
            # twice per call: the module is left alone, and what the upcast protects against - an fp16 overflow in the
            # encoder - is checked instead (deviations)
            vae_dtype = getattr(self.vae, "dtype", None)
            from .vae import AutoencoderKLTemporalDecoder
            hip_vae = isinstance(self.vae, AutoencoderKLTemporalDecoder)
            needs_upcasting = (not hip_vae and vae_dtype == torch.float16 and
                               bool(getattr(self.vae.config, "force_upcast", False)))
            if needs_upcasting:
                self.vae.to(dtype=torch.float32)
            elif vae_dtype is not None and vae_dtype != img.dtype:
                img = img.to(vae_dtype)
            image_latents = self._encode_vae_image(img, device, num_videos_per_prompt, cfg)
            image_latents = image_latents.to(image_embeddings.dtype)                                  # reference :480
            if needs_upcasting:
                self.vae.to(dtype=torch.float16)
            if hip_vae and not bool(torch.isfinite(image_latents).all()):
                raise LkgdHipError("VAE encode produced non-finite latents (fp16 range exceeded in the encoder): the reference "
                                   "encodes in fp32 under force_upcast; scale the input or encode outside and pass image_latents")
        image_latents = image_latents.to(device=device, dtype=torch.float16)
        if image_latents.dim() == 4:      # [cfg*B,4,h,w] -> repeat over frames (:488)
            image_latents = image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        added_time_ids = self._get_add_time_ids(fps, motion_bucket_id, noise_aug_strength, torch.float32, batch_size,
                                                num_videos_per_prompt, cfg).to(device)
        self.scheduler.set_timesteps(num_inference_steps, device=None)
        lat = self.prepare_latents(batch_size * num_videos_per_prompt, num_frames, self.unet.config.in_channels,
                                   height, width, torch.float16, device, generator, latents)
        if controlnet_condition is not None:
            # reference :546-550: VaeImageProcessor.preprocess ([0,1] -> [-1,1]), add the batch axis, duplicate for CFG
            cc = controlnet_condition
            if isinstance(cc, torch.Tensor) and cc.dim() == 5:      # already batched [1,F,3,H,W] in [0,1]
                cc = 2.0 * cc.to(device) - 1.0
            else:
                cc = self.image_processor.preprocess(cc, height=height, width=width).to(device)
            if cc.dim() == 4:
                cc = cc.unsqueeze(0)
            controlnet_condition = torch.cat([cc] * 2) if cfg else cc
        lat = self.denoise(lat, image_latents.contiguous(), image_embeddings, added_time_ids, num_inference_steps,
                           min_guidance_scale, max_guidance_scale, domain_features, flow_features,
                           callback_on_step_end, callback_on_step_end_tensor_inputs, controlnet_condition,
                           controlnet_cond_scale)
        if output_type != "latent":
            frames = self.decode_latents(self._latents_for_decode(lat), num_frames, decode_chunk_size)
            frames = tensor2vid(frames, self.image_processor, output_type=output_type)               # reference :644
        else:
            frames = lat
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)


class StableVideoDiffusionPipelineControlNet(StableVideoDiffusionPipeline):
    """/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py: ``controlnet_condition`` is the SECOND positional
    argument (:352-376); the loop runs the ControlNet-SVD encoder before the UNet every step (:582-607)"""

    def __call__(self, image, controlnet_condition=None, height: int = 576, width: int = 1024, num_frames=None,
                 num_inference_steps: int = 25, min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
                 fps: int = 7, motion_bucket_id: int = 127, noise_aug_strength: float = 0.02, decode_chunk_size=None,
                 num_videos_per_prompt=1, generator=None, latents=None, output_type="pil", callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=["latents"], return_dict: bool = True, controlnet_cond_scale=1.0,
                 batch_size=1, **extensions):
        return super().__call__(image, height, width, num_frames, num_inference_steps, min_guidance_scale,
                                max_guidance_scale, fps, motion_bucket_id, noise_aug_strength, decode_chunk_size,
                                num_videos_per_prompt, generator, latents, output_type, callback_on_step_end,
                                callback_on_step_end_tensor_inputs, return_dict,
                                controlnet_condition=self._condition(controlnet_condition),
                                controlnet_cond_scale=controlnet_cond_scale, **extensions)

    def _condition(self, controlnet_condition):
        return controlnet_condition


class StableVideoDiffusionPipelineControlNetFlow(StableVideoDiffusionPipelineControlNet):
    """/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet_flow.py (BASELINE.json configs[3]) AS THE
    REFERENCE RUNS IT: same signature, but its ControlNet call and the condition's preprocessing are commented out
    (:548-553,:590-607) - ``controlnet_condition`` is accepted and unused, the UNet runs without residuals - and the
    denoised latents are flow latents, un-normalised before the VAE decode (:641, utils/optical_flow.py:62-77)."""

    def _condition(self, controlnet_condition):
        return None

    def release_arena(self) -> None:
        """return the recorded forwards' working-set pool(s) to the driver (they are kept between calls otherwise)"""
        self._arenas.clear()

    def arena_reserved_bytes(self) -> int:
        """device memory the recorded forward's private pool holds (the plan's working set)"""
        return self._arenas.reserved_bytes()

    def _latents_for_decode(self, latents: torch.Tensor) -> torch.Tensor:
        from .optical_flow import optical_flow_latent_unnormalize
        return optical_flow_latent_unnormalize(latents)
