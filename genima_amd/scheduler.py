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
        if full.get("prediction_type", "epsilon") != "epsilon":  # the device step is x + eps * (sigma_next - sigma) only
            raise NotImplementedError(f"{type(self).__name__}: prediction_type={full['prediction_type']!r} is not built "
                                      "(SD-Turbo / SDXL-Turbo predict epsilon; SURVEY.md Appendix B)")
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

    # ---- sampling side: ``log_validation`` swaps the TRAINING scheduler class into the pipeline and runs 4 steps
    # (diffusion/train_controlnet_genima.py:545-553, 631-638).  Epsilon prediction without sample clipping, so a step is linear in
    # (x, eps, noise): x <- A x + B eps + C z; the pipeline lowers it to the scale / euler-step / add-noise kernels.
    sampler = "linear"
    ancestral = False
    draws_step_noise = True
    init_noise_sigma = 1.0

    @classmethod
    def from_config(cls, cfg, **kw):
        d = dict(vars(cfg)) if isinstance(cfg, SimpleNamespace) else dict(cfg)
        d.update(kw)
        d.pop("_class_name", None)
        d.pop("_diffusers_version", None)
        return cls(**d)

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        sp = self.config.get("timestep_spacing", "leading")
        if sp == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)) - 1
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy() + self.config.get("steps_offset", 0)
        elif sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy()
        else:
            raise ValueError(sp)
        if self.config.get("clip_sample", False) or self.config.get("thresholding", False):
            raise NotImplementedError("clip_sample / thresholding make the DDPM / DDIM step non-linear; SD-family configs set them to false")
        if self.config.get("prediction_type", "epsilon") != "epsilon":
            raise NotImplementedError("only epsilon prediction is on the Genima path")
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(ts.astype(np.int64))
        return self

    def input_scale(self, i: int) -> float:
        return 1.0

    def _alphas(self, i: int):
        t = int(self.timesteps[i])
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self._final_alpha_cumprod()
        return t, a_t, a_prev

    def _final_alpha_cumprod(self) -> float:
        return 1.0

    def step_coeffs(self, i: int):
        """diffusers DDPMScheduler.step, variance_type "fixed_small": -> (A, B, C)."""
        t, a_t, a_prev = self._alphas(i)
        b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1.0 - cur_a
        c_x0 = a_prev ** 0.5 * cur_b / b_t
        c_xt = cur_a ** 0.5 * b_prev / b_t
        var = max(b_prev / b_t * cur_b, 1e-20)
        return c_x0 / a_t ** 0.5 + c_xt, -c_x0 * b_t ** 0.5 / a_t ** 0.5, (var ** 0.5 if t > 0 else 0.0)


class DDIMScheduler(DDPMScheduler):
    """diffusers DDIMScheduler.step with eta = 0 (deterministic): x <- sqrt(a_prev) x0 + sqrt(1 - a_prev) eps."""
    draws_step_noise = False

    def __init__(self, **cfg):
        super().__init__(**cfg)
        self.config._class_name = "DDIMScheduler"

    def _final_alpha_cumprod(self) -> float:
        return 1.0 if self.config.get("set_alpha_to_one", True) else float(self.alphas_cumprod[0])

    def step_coeffs(self, i: int):
        t, a_t, a_prev = self._alphas(i)
        return (a_prev / a_t) ** 0.5, (1.0 - a_prev) ** 0.5 - (a_prev * (1.0 - a_t) / a_t) ** 0.5, 0.0
