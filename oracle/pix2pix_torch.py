"""CPU oracle for the InstructPix2Pix family: the pipeline loop and the fine-tune step as torch (autograd) over oracle/sd_torch.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED like sd_torch.py (diffusers 0.29.0 is absent from the reference
tree and the image; the reference has no tests for this path).  Restated here:

  * diffusers ``StableDiffusionInstructPix2PixPipeline.__call__`` as ``SDPix2PixAgent.infer`` calls it
    (controller/agent/sd_pix2pix_agent.py:51-60): prompt rows [text, negative, negative]; image latents = ``vae.encode(2 x - 1)
    .latent_dist.mode()`` (unscaled), [lat, lat, 0] under guidance; per step ``unet(cat([scale_model_input(latents)] * 3, image latents))``,
    sigma-space detour ``x0 = x - sigma * eps`` -> ``uncond + g (text - image) + ig (image - uncond)`` -> ``eps = (x0 - x) / (-sigma)``,
    Euler step; decode ``latents / scaling_factor``;
  * the reference's own step body, diffusion/train_instruct_pix2pix_genima.py:1165-1255: posterior sample x scaling_factor, DDPM
    add_noise, ``original_image_embeds = vae.encode(original).latent_dist.mode()``, conditioning dropout (:1204-1233),
    ``cat([noisy, embeds], dim=1)``, epsilon target, MSE; EMA step of diffusers ``EMAModel`` (:1271-1272).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import scheduler as OS
from . import sd_torch as O

Tensor = torch.Tensor


def pipeline(unet_sd, unet_cfg, vae_sd, vae_cfg, text_sd, text_cfg, sched_cfg, ids: Tensor, neg_ids: Optional[Tensor], image01: Tensor,
             latents: Tensor, steps: int, guidance_scale: float, image_guidance_scale: float = 1.5, q: Callable = O._id) -> Tuple[Tensor, Tensor]:
    """image01: NCHW in [0, 1]; latents: unit-variance NCHW draws.  -> (final latents, decoded image in [-1, 1])."""
    cfg = guidance_scale > 1.0 and image_guidance_scale >= 1.0
    ts, sig, init = OS.euler_set_timesteps(sched_cfg, steps)
    ctx = O.clip_text_forward(text_sd, text_cfg, ids, q)
    mean, _ = O.vae_encode_moments(vae_sd, vae_cfg, q(2.0 * image01 - 1.0), q)
    img_lat = q(mean)
    if cfg:
        nctx = O.clip_text_forward(text_sd, text_cfg, neg_ids, q)
        ctx = torch.cat([ctx, nctx, nctx])
        img_lat = torch.cat([img_lat, img_lat, torch.zeros_like(img_lat)])
    x = q(latents * init)
    B = latents.shape[0]
    for i in range(steps):
        sigma = float(sig[i])
        xin = torch.cat([x] * 3) if cfg else x
        scaled = q(xin / float((sigma ** 2 + 1) ** 0.5))
        t = torch.full((xin.shape[0],), float(ts[i]))
        eps = O.unet_forward(unet_sd, unet_cfg, torch.cat([scaled, img_lat], dim=1), t, ctx, q=q)
        if cfg:
            x0 = xin - sigma * eps                        # "karras style" detour of the diffusers pipeline
            x0_t, x0_i, x0_u = x0.chunk(3)
            x0 = x0_u + guidance_scale * (x0_t - x0_i) + image_guidance_scale * (x0_i - x0_u)
            eps = (x0 - x) / (-sigma)
        x = q(torch.from_numpy(OS.euler_step(eps.numpy(), sigma, float(sig[i + 1]), x.numpy())))
    assert x.shape[0] == B
    return x, O.vae_decode(vae_sd, vae_cfg, q(x / vae_cfg.get("scaling_factor", 0.18215)), q)


def conditioning_dropout(ctx: Tensor, null_ctx: Tensor, image_embeds: Tensor, random_p: Tensor, p: float) -> Tuple[Tensor, Tensor]:
    """diffusion/train_instruct_pix2pix_genima.py:1204-1233."""
    B = ctx.shape[0]
    prompt_mask = (random_p < 2 * p).reshape(B, 1, 1)
    ctx = torch.where(prompt_mask, null_ctx, ctx)
    dt = image_embeds.dtype
    image_mask = 1 - ((random_p >= p).to(dt) * (random_p < 3 * p).to(dt))
    return ctx, image_mask.reshape(B, 1, 1, 1) * image_embeds


def train_forward_backward(unet_sd: Dict[str, Tensor], unet_cfg, latents: Tensor, noise: Tensor, t: Tensor, sqrt_ac: Tensor,
                           sqrt_1mac: Tensor, ctx: Tensor, image_embeds: Tensor, q: Callable = O._id):
    """-> (loss, {name: d loss / d unet parameter}, model_pred): :1194, :1236-1255."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in unet_sd.items()}
    noisy = q(sqrt_ac.view(-1, 1, 1, 1) * latents + sqrt_1mac.view(-1, 1, 1, 1) * noise)
    pred = O.unet_forward(params, unet_cfg, torch.cat([noisy, image_embeds], dim=1), t, ctx, q=q)
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    return loss.detach(), grads, pred.detach()


def ema_decay(optimization_step: int, decay: float = 0.9999, min_decay: float = 0.0, update_after_step: int = 0) -> float:
    """diffusers EMAModel.get_decay without warm-up (the reference constructs EMAModel with its defaults, :821-824)."""
    step = max(0, optimization_step - update_after_step - 1)
    if step <= 0:
        return 0.0
    return max(min((1 + step) / (10 + step), decay), min_decay)


def ema_step(shadow: Tensor, param: Tensor, optimization_step: int) -> Tensor:
    """EMAModel.step for one tensor: ``s_param.sub_(one_minus_decay * (s_param - param))`` with the decay of the incremented step count."""
    one_minus = 1 - ema_decay(optimization_step)
    return shadow - one_minus * (shadow - param)
