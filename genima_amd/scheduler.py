"""Schedulers of the Genima hot path (host-side tables; the per-element math runs in gn_scale_pad / gn_euler_step /
gn_add_noise on the device).

Mirrors the surface the reference touches on diffusers 0.29.0 schedulers: ``EulerDiscreteScheduler`` inside
``self.pipe(...)`` (controller/agent/sd_controlnet_agent.py:67-76) and ``DDPMScheduler.add_noise`` / ``.config`` in the
trainer (diffusion/train_controlnet_genima.py:1012-1040, 1350-1399).  The beta/sigma tables are built with float32 torch
CPU ops exactly as diffusers builds them (SURVEY.md Appendix B pins sigma_max = 14.614647 and the trailing timesteps).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from .configs import SD_TURBO_SCHEDULER


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)


def _alphas_cumprod(cfg) -> torch.Tensor:
    n = cfg["num_train_timesteps"]
    if cfg.get("beta_schedule", "scaled_linear") == "scaled_linear":
        betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
    elif cfg["beta_schedule"] == "linear":
        betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
    else:
        raise NotImplementedError(cfg["beta_schedule"])
    return torch.cumprod(1.0 - betas, dim=0)


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, **cfg):
        full = dict(SD_TURBO_SCHEDULER)
        full.update(cfg)
        self.config = _Config(**full)
        self.alphas_cumprod = _alphas_cumprod(full)
        ac = self.alphas_cumprod
        self._train_sigmas = (((1 - ac) / ac) ** 0.5).numpy()
        self.timesteps: Optional[torch.Tensor] = None
        self.sigmas: Optional[torch.Tensor] = None
        self.num_inference_steps = None
        # diffusers 0.29.0 draws randn_tensor(model_output.shape, generator=...) in every step() even when gamma == 0
        # (SURVEY.md Appendix D.5 [VERIFY]); the pipeline mirrors the draw so an episode's shared generator advances alike.
        self.draws_step_noise = True

    @classmethod
    def from_config(cls, cfg, **kw):
        d = dict(vars(cfg)) if isinstance(cfg, SimpleNamespace) else dict(cfg)
        d.update(kw)
        d.pop("_class_name", None)
        d.pop("_diffusers_version", None)
        return cls(**d)

    @property
    def init_noise_sigma(self) -> float:
        smax = float(self.sigmas.max()) if self.sigmas is not None else float(self._train_sigmas.max())
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return smax
        return float((smax ** 2 + 1) ** 0.5)

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        sp = self.config.timestep_spacing
        if sp == "trailing":
            ts = (np.round(np.arange(n, 0, -n / num_inference_steps)) - 1).astype(np.float32)
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy().astype(np.float32)
            ts += self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        else:
            raise ValueError(sp)
        sig = np.interp(ts, np.arange(0, n), self._train_sigmas)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(ts)
        self.sigmas = torch.from_numpy(sig)
        return self

    def input_scale(self, i: int) -> float:
        s = float(self.sigmas[i])
        return float(1.0 / (s * s + 1.0) ** 0.5)


class EulerAncestralDiscreteScheduler(EulerDiscreteScheduler):
    """sdxl-turbo's default sampler: same sigma table / trailing timesteps / input scaling as
    EulerDiscrete, but every step lands on sigma_down and re-injects fresh unit noise scaled by sigma_up:
        sigma_up   = sqrt(sigma_to^2 * (sigma_from^2 - sigma_to^2) / sigma_from^2),   sigma_down = sqrt(sigma_to^2 - sigma_up^2)
        x <- x + eps * (sigma_down - sigma_from) + noise * sigma_up            (diffusers EulerAncestralDiscreteScheduler.step)."""
    ancestral = True

    @property
    def init_noise_sigma(self) -> float:
        smax = float(self.sigmas.max()) if self.sigmas is not None else float(self._train_sigmas.max())
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return smax
        return float((smax ** 2 + 1) ** 0.5)

    def ancestral_sigmas(self, i: int):
        s_from, s_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
        down = (s_to ** 2 - up ** 2) ** 0.5
        return float(down), float(up)


class DDPMScheduler:
    """Training-side noise scheduler: ``add_noise`` coefficients + config (diffusion/train_controlnet_genima.py:1350-1399)."""

    def __init__(self, **cfg):
        full = dict(SD_TURBO_SCHEDULER, _class_name="DDPMScheduler")
        full.update(cfg)
        self.config = _Config(**full)
        self.alphas_cumprod = _alphas_cumprod(full)

    def add_noise_coeffs(self, timesteps: torch.Tensor):
        ac = self.alphas_cumprod[timesteps.to("cpu", torch.long)]
        return (ac ** 0.5).to(torch.float32), ((1 - ac) ** 0.5).to(torch.float32)
