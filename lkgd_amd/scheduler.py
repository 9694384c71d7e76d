"""EulerDiscreteScheduler with the reference's interface, MI355X-side.

Mirrors /root/reference/utils/scheduling_euler_discrete_karras_fix.py: constructor :178-246, ``init_noise_sigma``
:248-255, ``scale_model_input`` :264-288, ``set_timesteps`` :290-350, ``_sigma_to_t`` :352-373,
``_convert_to_karras`` :376-399, ``_init_step_index`` :401-416, ``step`` :418-528, ``add_noise`` :530-553.

The sigma / timestep tables are tiny host computations (numpy, float64 -> float32 exactly as the reference) and stay
on the HOST: the loop then needs no device->host sync at all (the reference syncs in ``_init_step_index`` and indexes
a device tensor every step).  ``.sigmas`` / ``.timesteps`` are still exposed as tensors on the requested device for
callers that read them.  Tensor arithmetic (``scale_model_input``, ``step``) runs in HIP kernels
(include/lkgd_hip.h section 8); inside lkgd_amd.pipeline the same math is fused into the loop-glue kernels.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops

logger = logging.getLogger(__name__)


@dataclass
class EulerDiscreteSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, prediction_type: str = "epsilon",
                 interpolation_type: str = "linear", use_karras_sigmas: Optional[bool] = False,
                 sigma_min: Optional[float] = None, sigma_max: Optional[float] = None,
                 timestep_spacing: str = "linspace", timestep_type: str = "discrete", steps_offset: int = 0,
                 rescale_betas_zero_snr: bool = False):
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, prediction_type=prediction_type,
            interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas, sigma_min=sigma_min,
            sigma_max=sigma_max, timestep_spacing=timestep_spacing, timestep_type=timestep_type,
            steps_offset=steps_offset, rescale_betas_zero_snr=rescale_betas_zero_snr)
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:
            raise NotImplementedError("rescale_betas_zero_snr is not used by SVD")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.use_karras_sigmas = use_karras_sigmas
        self.is_scale_input_called = False
        self.num_inference_steps = None
        self._step_index = None
        self._device = None
        self.set_timesteps_full()

    @classmethod
    def from_svd_config(cls):
        """SVD ``scheduler/scheduler_config.json``"""
        return cls(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                   prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True,
                   sigma_min=0.002, sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous",
                   steps_offset=1)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, **_ignored):
        """``SchedulerMixin.from_pretrained`` [EXT]: ``scheduler_config.json`` of a local pipeline directory"""
        from .loading import load_config
        import inspect
        raw = load_config(pretrained_model_name_or_path, subfolder, "scheduler_config.json")
        known = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in raw.items() if k in known})

    def save_pretrained(self, save_directory: str, **_ignored):
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": "EulerDiscreteScheduler"}
        cfg.update({k: v for k, v in vars(self.config).items()})
        with open(os.path.join(save_directory, "scheduler_config.json"), "w") as fh:
            json.dump(cfg, fh, indent=2, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))

    # ---- host tables ------------------------------------------------------------------------------------------
    def _train_sigmas(self) -> np.ndarray:
        ac = self.alphas_cumprod.numpy()
        return ((1 - ac) / ac) ** 0.5

    def set_timesteps_full(self):
        """state right after construction (reference __init__ :220-241): the full training schedule"""
        n = self.config.num_train_timesteps
        sigmas = self._train_sigmas()[::-1].copy()
        timesteps = np.linspace(0, n - 1, n, dtype=float)[::-1].copy()
        if self.use_karras_sigmas:
            log_sigmas = np.log(sigmas)
            sigmas = self._convert_to_karras(sigmas, n)
            timesteps = np.array([self._sigma_to_t(s, log_sigmas) for s in sigmas])
        self._publish(sigmas, timesteps, None)

    def _publish(self, sigmas: np.ndarray, timesteps: np.ndarray, device):
        s32 = torch.from_numpy(np.asarray(sigmas)).to(torch.float32)
        if self.config.timestep_type == "continuous" and self.config.prediction_type == "v_prediction":
            t32 = torch.Tensor([0.25 * s.log() for s in s32])
        else:
            t32 = torch.from_numpy(np.asarray(timesteps).astype(np.float32))
        s32 = torch.cat([s32, torch.zeros(1)])
        self.sigmas_host: List[float] = [float(x) for x in s32]
        self.timesteps_host: List[float] = [float(x) for x in t32]
        self.sigmas = s32.to(device) if device is not None else s32
        self.timesteps = t32.to(device) if device is not None else t32
        self._step_index = None

    @property
    def init_noise_sigma(self):
        m = max(self.sigmas_host)
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return torch.tensor(m)
        # float32 arithmetic as the reference's tensor expression (max_sigma**2 + 1) ** 0.5
        return (torch.tensor(m, dtype=torch.float32) ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def _convert_to_karras(self, in_sigmas, num_inference_steps):
        smin = self.config.sigma_min if self.config.sigma_min is not None else float(in_sigmas[-1])
        smax = self.config.sigma_max if self.config.sigma_max is not None else float(in_sigmas[0])
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        min_inv_rho, max_inv_rho = smin ** (1 / rho), smax ** (1 / rho)
        return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho

    @staticmethod
    def _sigma_to_t(sigma, log_sigmas):
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = log_sigmas[low_idx], log_sigmas[high_idx]
        w = np.clip((low - log_sigma) / (low - high), 0, 1)
        return ((1 - w) * low_idx + w * high_idx).reshape(sigma.shape)

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        if c.timestep_spacing == "linspace":
            timesteps = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif c.timestep_spacing == "leading":
            step_ratio = c.num_train_timesteps // num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
            timesteps += c.steps_offset
        elif c.timestep_spacing == "trailing":
            step_ratio = c.num_train_timesteps / num_inference_steps
            timesteps = (np.arange(c.num_train_timesteps, 0, -step_ratio)).round().copy().astype(np.float32)
            timesteps -= 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported. Please make sure to choose one of 'linspace', "
                             "'leading' or 'trailing'.")
        sigmas = self._train_sigmas()
        log_sigmas = np.log(sigmas)
        if c.interpolation_type == "linear":
            sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        elif c.interpolation_type == "log_linear":
            sigmas = torch.linspace(np.log(sigmas[-1]), np.log(sigmas[0]), num_inference_steps + 1).exp().numpy()
        else:
            raise ValueError(f"{c.interpolation_type} is not implemented. Please specify interpolation_type to "
                             "either 'linear' or 'log_linear'")
        if self.use_karras_sigmas:
            sigmas = self._convert_to_karras(sigmas, num_inference_steps)
            timesteps = np.array([self._sigma_to_t(s, log_sigmas) for s in sigmas])
        self._device = device
        self._publish(sigmas, timesteps, device)

    def _init_step_index(self, timestep):
        t = float(timestep)   # a device tensor syncs here once, as in the reference (:405-416)
        cand = [i for i, v in enumerate(self.timesteps_host) if v == t]
        if not cand:
            raise ValueError(f"timestep {t} is not one of scheduler.timesteps")
        self._step_index = cand[1] if len(cand) > 1 else cand[0]

    # ---- tensor ops -------------------------------------------------------------------------------------------
    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas_host[self._step_index]
        self.is_scale_input_called = True
        return ops.scale(sample, 1.0 / ((sigma ** 2 + 1) ** 0.5))

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, s_churn: float = 0.0,
             s_tmin: float = 0.0, s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None,
             return_dict: bool = True):
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `EulerDiscreteScheduler.step()` is not supported. Make sure to pass"
                             " one of the `scheduler.timesteps` as a timestep.")
        if not self.is_scale_input_called:
            logger.warning("The `scale_model_input` function should be called before `step` to ensure correct "
                           "denoising. See `StableDiffusionPipeline` for a usage example.")
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma, sigma_next = self.sigmas_host[self._step_index], self.sigmas_host[self._step_index + 1]
        pt = self.config.prediction_type
        if pt not in ("epsilon", "v_prediction"):
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, or `v_prediction`")
        gamma = min(s_churn / (len(self.sigmas_host) - 1), 2 ** 0.5 - 1) if s_tmin <= sigma <= s_tmax else 0.0
        noise = None
        if gamma > 0 or generator is not None:
            # noise as randn_tensor draws it (model_output's dtype; a generator of another device draws on that device and the
            # tensor is moved).  The reference draws at EVERY step, used or not: with a caller's generator that is observable
            # (its state after the step), so it is drawn here too; the global RNG is left alone
            gdev = generator.device if generator is not None else model_output.device
            noise = torch.randn(model_output.shape, generator=generator, device=gdev, dtype=model_output.dtype)
        if gamma > 0:
            # stochastic step (reference :485-497), scalars in fp32 as the reference's 0-dim tensors
            noise = noise.to(model_output.device)
            sig = torch.tensor(sigma, dtype=torch.float32)
            sig_hat = sig * (gamma + 1)
            churn = float((sig_hat ** 2 - sig ** 2) ** 0.5)
            prev = ops.euler_step(model_output, sample, sigma, sigma_next, v_prediction=(pt == "v_prediction"),
                                  noise=noise, sigma_hat=float(sig_hat), s_noise=float(s_noise), churn=churn)
        else:
            prev = ops.euler_step(model_output, sample, sigma, sigma_next, v_prediction=(pt == "v_prediction"))
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return EulerDiscreteSchedulerOutput(prev_sample=prev)

    def add_noise(self, original_samples, noise, timesteps):
        idx = []
        for t in timesteps:
            c = [i for i, v in enumerate(self.timesteps_host) if v == float(t)]
            idx.append(c[0])
        sig = torch.tensor([self.sigmas_host[i] for i in idx], device=original_samples.device,
                           dtype=original_samples.dtype)
        while sig.dim() < original_samples.dim():
            sig = sig.unsqueeze(-1)
        return original_samples + noise * sig   # training-side helper, not on the sampling path

    def __len__(self):
        return self.config.num_train_timesteps
