"""CPU oracle (numpy): EulerDiscreteScheduler (trailing) and DDPM add_noise as the reference uses them.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates diffusers==0.29.0
``schedulers/scheduling_euler_discrete.py`` / ``scheduling_ddpm.py`` (absent third-party dependency,
poetry.lock:595-596) as reached from ``self.pipe(...)`` (controller/agent/sd_controlnet_agent.py:67-76)
and ``noise_scheduler.add_noise`` (diffusion/train_controlnet_genima.py:1359).  Pinned by the
golden tables of SURVEY.md Appendix B (tests/golden/scheduler_tables.json): timesteps bit-exact,
sigmas to float32 6 dp.
"""
from __future__ import annotations

import numpy as np


def betas_scaled_linear(cfg) -> np.ndarray:
    """fp32 torch ops exactly as diffusers builds the schedule (numpy's linspace/cumprod round differently
    in the 7th digit; SURVEY Appendix B pins the torch result: sigma_max = 14.614647)."""
    import torch

    n = cfg["num_train_timesteps"]
    b = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
    return b.numpy()


def alphas_cumprod(cfg) -> np.ndarray:
    import torch

    return torch.cumprod(1.0 - torch.from_numpy(betas_scaled_linear(cfg)), dim=0).numpy()


def train_sigmas(cfg) -> np.ndarray:
    import torch

    ac = torch.from_numpy(alphas_cumprod(cfg))
    return (((1 - ac) / ac) ** 0.5).numpy()


def euler_set_timesteps(cfg, num_inference_steps: int):
    """-> (timesteps int64 [N], sigmas float32 [N+1], init_noise_sigma)."""
    n = cfg["num_train_timesteps"]
    spacing = cfg.get("timestep_spacing", "trailing")
    if spacing == "trailing":
        step_ratio = n / num_inference_steps
        ts = np.round(np.arange(n, 0, -step_ratio)) - 1
    elif spacing == "leading":
        step_ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy() + cfg.get("steps_offset", 0)
    elif spacing == "linspace":
        ts = np.linspace(0, n - 1, num_inference_steps)[::-1].copy()
    else:
        raise ValueError(spacing)
    ts = ts.astype(np.float32)
    sig = np.interp(ts, np.arange(0, n), train_sigmas(cfg))
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    smax = float(sig.max())
    init = smax if spacing in ("linspace", "trailing") else float((smax * smax + 1.0) ** 0.5)
    return ts.astype(np.int64), sig, init


def euler_scale_model_input(x: np.ndarray, sigma: float) -> np.ndarray:
    return (x / np.float32((np.float32(sigma) ** 2 + 1) ** 0.5)).astype(x.dtype)


def euler_step(eps: np.ndarray, sigma: float, sigma_next: float, x: np.ndarray) -> np.ndarray:
    """epsilon prediction, gamma = 0: x <- x + eps * (sigma_next - sigma), fp32 then cast back."""
    xf = x.astype(np.float32)
    ef = eps.astype(np.float32)
    x0 = xf - np.float32(sigma) * ef
    d = (xf - x0) / np.float32(sigma)
    return (xf + d * np.float32(np.float32(sigma_next) - np.float32(sigma))).astype(x.dtype)


def ddpm_add_noise_coeffs(cfg, t: np.ndarray):
    ac = alphas_cumprod(cfg)
    import torch

    act = torch.from_numpy(ac)[torch.as_tensor(np.asarray(t), dtype=torch.long)]
    return (act ** 0.5).numpy(), ((1 - act) ** 0.5).numpy()


def ddpm_add_noise(cfg, x0: np.ndarray, noise: np.ndarray, t: np.ndarray) -> np.ndarray:
    a, b = ddpm_add_noise_coeffs(cfg, t)
    shp = (-1,) + (1,) * (x0.ndim - 1)
    return a.reshape(shp) * x0 + b.reshape(shp) * noise


# ---- samplers log_validation swaps in (diffusion/train_controlnet_genima.py:545-553): the published step() of diffusers' DDPMScheduler
# (variance_type "fixed_small", clip_sample False) and DDIMScheduler (eta 0), written out un-folded (x0 prediction first).
def trailing_timesteps(cfg, num_inference_steps: int) -> np.ndarray:
    n = cfg["num_train_timesteps"]
    return (np.round(np.arange(n, 0, -n / num_inference_steps)) - 1).astype(np.int64)


def _prev_alphas(cfg, t: int, num_inference_steps: int, final_alpha: float):
    ac = alphas_cumprod(cfg).astype(np.float64)
    prev_t = t - cfg["num_train_timesteps"] // num_inference_steps
    return float(ac[t]), float(ac[prev_t] if prev_t >= 0 else final_alpha)


def ddpm_step(cfg, eps, t: int, x, noise, num_inference_steps: int):
    a_t, a_prev = _prev_alphas(cfg, t, num_inference_steps, 1.0)
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    x0 = (x - b_t ** 0.5 * eps) / a_t ** 0.5
    mean = (a_prev ** 0.5 * cur_b / b_t) * x0 + (cur_a ** 0.5 * b_prev / b_t) * x
    var = max(b_prev / b_t * cur_b, 1e-20)
    return mean + (var ** 0.5) * noise if t > 0 else mean


def ddim_step(cfg, eps, t: int, x, num_inference_steps: int, set_alpha_to_one: bool = True):
    final = 1.0 if set_alpha_to_one else float(alphas_cumprod(cfg)[0])
    a_t, a_prev = _prev_alphas(cfg, t, num_inference_steps, final)
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
