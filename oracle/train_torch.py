"""CPU oracle for the ControlNet fine-tune step: torch autograd over the fp32 restatement in oracle/sd_torch.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for the same reason as sd_torch.py: the arithmetic lives in
diffusers==0.29.0 / torch / accelerate, none of which the reference vendors, and the reference has no training tests or golden
vectors.  What is restated here is the reference's own step body (diffusion/train_controlnet_genima.py):

    noisy_latents = noise_scheduler.add_noise(latents, noise, timesteps)                       :1359
    down, mid = controlnet(noisy_latents, t, encoder_hidden_states, controlnet_cond)           :1368-1374
    model_pred = unet(noisy_latents, t, ctx, down_block_additional_residuals, mid_...)         :1377-1388
    loss = F.mse_loss(model_pred.float(), noise.float(), reduction="mean")                     :1391-1400
    accelerator.backward(loss); clip_grad_norm_(params, 1.0); optimizer.step(); zero_grad      :1402-1408
    optimizer = torch.optim.AdamW(params, lr, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8) :1178-1185
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.nn.functional as F

from . import sd_torch as O

Tensor = torch.Tensor


def train_forward_backward(unet_sd: Dict[str, Tensor], cn_sd: Dict[str, Tensor], unet_cfg, cn_cfg, latents: Tensor, noise: Tensor,
                           t: Tensor, sqrt_ac: Tensor, sqrt_1mac: Tensor, ctx: Tensor, cond: Tensor,
                           q: Callable = O._id, added=None, prediction_type: str = "epsilon") -> Tuple[Tensor, Dict[str, Tensor], Tensor]:
    """NCHW fp32 inputs.  -> (loss, {name: d loss / d controlnet parameter}, model_pred).
    ``added`` = (text_embeds, time_ids): the SDXL step (diffusion/train_controlnet_sdxl_genima.py:1448-1471) passes them to both nets."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in cn_sd.items()}
    noisy = q(sqrt_ac.view(-1, 1, 1, 1) * latents + sqrt_1mac.view(-1, 1, 1, 1) * noise)
    down, mid = O.controlnet_forward(params, cn_cfg, noisy, t, ctx, cond, q=q, added=added)
    pred = O.unet_forward(unet_sd, unet_cfg, noisy, t, ctx, down, mid, q=q, added=added)
    target = noise
    if prediction_type == "v_prediction":  # noise_scheduler.get_velocity (diffusion/train_controlnet_genima.py:1393-1394)
        target = q(sqrt_ac.view(-1, 1, 1, 1) * noise - sqrt_1mac.view(-1, 1, 1, 1) * latents)
    loss = F.mse_loss(pred.float(), target.float(), reduction="mean")
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    return loss.detach(), grads, pred.detach()


def adamw_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], lr: float, max_grad_norm: float = 1.0, betas=(0.9, 0.999),
               weight_decay: float = 1e-2, eps: float = 1e-8) -> Tuple[Dict[str, Tensor], float]:
    """One clip_grad_norm_ + torch.optim.AdamW step from zero moments.  -> (new params, pre-clip global norm)."""
    ps = [torch.nn.Parameter(v.detach().clone()) for v in params.values()]
    for p, k in zip(ps, params):
        p.grad = grads[k].detach().clone()
    norm = float(torch.nn.utils.clip_grad_norm_(ps, max_grad_norm))
    torch.optim.AdamW(ps, lr=lr, betas=betas, weight_decay=weight_decay, eps=eps).step()
    return {k: p.detach() for k, p in zip(params, ps)}, norm
