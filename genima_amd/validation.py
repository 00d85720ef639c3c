"""``log_validation`` of the ControlNet trainer (diffusion/train_controlnet_genima.py:517-718) on the HIP path.

The reference builds a fresh ``StableDiffusionControlNetPipeline`` around the live modules (:532-542), swaps in the TRAINING
scheduler class built from the pipeline scheduler's config (:545-553), runs 4 steps at guidance 0 (:631-638) and scores the sample
against the ground-truth render (:642-650).  Picking the task / episode / frame from the dataset tree and the wandb / tensorboard
upload (:575-625, 663-716) are control plane and stay with the caller; this module is the device part and the error images.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .host import ControlNetModel
from .pipeline import StableDiffusionControlNetPipeline
from .scheduler import DDIMScheduler, DDPMScheduler, EulerDiscreteScheduler

_SCHEDULERS = {"ddpm": DDPMScheduler, "euler_discrete": EulerDiscreteScheduler, "ddim": DDIMScheduler}


def validation_pipeline(vae, text_encoder, tokenizer, unet, controlnet, train_scheduler: str = "ddpm", device="cuda"):
    """:532-553 -- the pipeline around the live modules with the training scheduler class swapped in.  ``controlnet`` may be a
    ``ControlNetModel`` or a ``training.ControlNetTrainer`` (its current fp32 master weights are exported)."""
    if train_scheduler not in _SCHEDULERS:
        raise ValueError(f"Scheduler {train_scheduler} not supported")
    if hasattr(controlnet, "controlnet_state_dict"):
        controlnet = ControlNetModel(dict(controlnet.cn_cfg), controlnet.controlnet_state_dict())
    base = EulerDiscreteScheduler()
    pipe = StableDiffusionControlNetPipeline(vae, text_encoder, tokenizer, unet, controlnet, base)
    pipe.scheduler = _SCHEDULERS[train_scheduler].from_config(pipe.scheduler.config)
    pipe.to(device)
    pipe.set_progress_bar_config(disable=True)
    return pipe


def normalized_error(image: np.ndarray, gt_image: np.ndarray):
    """:642-650 verbatim semantics: uint8 difference (wraps like numpy's), its mean square, and the difference / sqrt(mse) * 255."""
    difference_image = np.array(image) - np.array(gt_image)
    mse = np.mean(np.square(difference_image))
    norm_mse = difference_image / np.sqrt(mse) if mse > 0 else difference_image
    return norm_mse * 255, mse


def log_validation(pipe, validation_image, gt_image, validation_prompt: str, seed: Optional[int] = None, num_samples: int = 1) -> List[Dict]:
    """One validation item: 4 inference steps, guidance 0, seeded generator on the pipeline's device (:560-563, 631-660)."""
    generator = None if seed is None else torch.Generator(device=pipe.device).manual_seed(seed)
    images, errors, mse = [], [], 0.0
    for _ in range(num_samples):
        image = pipe(prompt=validation_prompt, image=validation_image, num_inference_steps=4, generator=generator, guidance_scale=0.0).images[0]
        images.append(image)
        err, mse = normalized_error(np.asarray(image), np.asarray(gt_image))
        errors.append(err)
    return [{"validation_image": validation_image, "gt_image": gt_image, "images": images, "errors": errors,
             "validation_prompt": validation_prompt, "mse": mse}]
