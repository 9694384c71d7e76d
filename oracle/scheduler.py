"""Restatement of the reference Euler-discrete scheduler.  ORACLE - test infrastructure only.

Follows /root/reference/utils/scheduling_euler_discrete_karras_fix.py:
``__init__`` :178-246, ``init_noise_sigma`` :248-255, ``scale_model_input`` :264-288, ``set_timesteps`` :290-350,
``_sigma_to_t`` :352-373, ``_convert_to_karras`` :376-399, ``_init_step_index`` :401-416, ``step`` :418-528.

Pinned: tests/golden/scheduler_kat.json was produced by running that reference file itself (name-only diffusers
stubs, tests/golden/make_goldens.py).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class SchedulerConfig:
    """SVD ``scheduler/scheduler_config.json`` values."""
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"
    prediction_type: str = "v_prediction"
    interpolation_type: str = "linear"
    use_karras_sigmas: bool = True
    sigma_min: float = 0.002
    sigma_max: float = 700.0
    timestep_spacing: str = "leading"
    timestep_type: str = "continuous"
    steps_offset: int = 1


class EulerDiscreteOracle:
    order = 1

    def __init__(self, cfg: SchedulerConfig = SchedulerConfig()):
        self.config = cfg
        if cfg.beta_schedule == "linear":
            betas = torch.linspace(cfg.beta_start, cfg.beta_end, cfg.num_train_timesteps, dtype=torch.float32)
        elif cfg.beta_schedule == "scaled_linear":
            betas = torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, cfg.num_train_timesteps,
                                   dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(cfg.beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.sigmas = None
        self.timesteps = None
        self._step_index = None
        self.set_timesteps(cfg.num_train_timesteps)

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return m
        return (m ** 2 + 1) ** 0.5

    def _convert_to_karras(self, in_sigmas, n):
        smin = self.config.sigma_min if self.config.sigma_min is not None else float(in_sigmas[-1])
        smax = self.config.sigma_max if self.config.sigma_max is not None else float(in_sigmas[0])
        rho = 7.0
        ramp = np.linspace(0, 1, n)
        return (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho

    @staticmethod
    def _sigma_to_t(sigma, log_sigmas):
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = log_sigmas[low_idx], log_sigmas[high_idx]
        w = np.clip((low - log_sigma) / (low - high), 0, 1)
        return ((1 - w) * low_idx + w * high_idx).reshape(sigma.shape)

    def set_timesteps(self, n: int, device=None):
        c = self.config
        self.num_inference_steps = n
        if c.timestep_spacing == "linspace":
            timesteps = np.linspace(0, c.num_train_timesteps - 1, n, dtype=np.float32)[::-1].copy()
        elif c.timestep_spacing == "leading":
            ratio = c.num_train_timesteps // n
            timesteps = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.float32) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = c.num_train_timesteps / n
            timesteps = np.arange(c.num_train_timesteps, 0, -ratio).round().copy().astype(np.float32) - 1
        else:
            raise ValueError(c.timestep_spacing)
        ac = self.alphas_cumprod.numpy()
        sigmas = ((1 - ac) / ac) ** 0.5
        log_sigmas = np.log(sigmas)
        if c.interpolation_type == "linear":
            sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        else:
            raise ValueError(c.interpolation_type)
        if c.use_karras_sigmas:
            sigmas = self._convert_to_karras(sigmas, n)
            timesteps = np.array([self._sigma_to_t(s, log_sigmas) for s in sigmas])
        sigmas = torch.from_numpy(sigmas).to(torch.float32)
        if c.timestep_type == "continuous" and c.prediction_type == "v_prediction":
            self.timesteps = torch.Tensor([0.25 * s.log() for s in sigmas])
        else:
            self.timesteps = torch.from_numpy(timesteps.astype(np.float32))
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self._step_index = None

    def _init_step_index(self, t):
        cand = (self.timesteps == t).nonzero()
        self._step_index = (cand[1] if len(cand) > 1 else cand[0]).item()

    def scale_model_input(self, sample, t):
        if self._step_index is None:
            self._init_step_index(t)
        sigma = self.sigmas[self._step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, t, sample, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, generator=None):
        """scheduling_euler_discrete_karras_fix.py:418-528; s_churn > 0 is the stochastic form (:485-497)"""
        if self._step_index is None:
            self._init_step_index(t)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        gamma = min(s_churn / (len(self.sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigma <= s_tmax else 0.0
        sigma_hat = sigma * (gamma + 1)
        if gamma > 0 or generator is not None:      # the reference draws at every step; a caller's generator must advance alike
            noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
        if gamma > 0:
            sample = sample + (noise * s_noise) * (sigma_hat ** 2 - sigma ** 2) ** 0.5
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = sample - sigma_hat * model_output
        elif pt == "v_prediction":
            x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        else:
            raise ValueError(pt)
        derivative = (sample - x0) / sigma_hat
        dt = self.sigmas[self._step_index + 1] - sigma_hat
        prev = (sample + derivative * dt).to(model_output.dtype)
        self._step_index += 1
        return prev
