"""Drop-in diffusion-agent plugins: same class names, constructor and ``infer`` contract as the reference's
controller/agent/{diffusion_agent,sd_controlnet_agent}.py, so ``diffusion_agent._target_: 'agent.SDControlNetAgent'``
(controller/cfgs/eval_genima.yaml:27-28) can resolve to this module unchanged.
"""
from __future__ import annotations

import os
import re

import torch

from . import configs
from .host import ControlNetModel
from .pipeline import StableDiffusionControlNetPipeline


def _natural_key(s):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


class _ResizeCenterCrop:
    """transforms.Compose([Resize(r, BILINEAR), CenterCrop(r)]) on PIL images (controller/agent/diffusion_agent.py:44-62)."""

    def __init__(self, r):
        self.r = r

    def __call__(self, im):
        from PIL import Image

        w, h = im.size
        if min(w, h) != self.r:
            if w <= h:
                nw, nh = self.r, int(self.r * h / w)
            else:
                nw, nh = int(self.r * w / h), self.r
            im = im.resize((nw, nh), Image.BILINEAR)
            w, h = im.size
        l, t = int(round((w - self.r) / 2.0)), int(round((h - self.r) / 2.0))
        return im.crop((l, t, l + self.r, t + self.r))


class DiffusionAgent:
    def __init__(self, eval_cfg):
        self.eval_cfg = eval_cfg
        self.pipe = None
        self.load_checkpoint()
        self.set_optimizations()
        self.common_setup()

    def load_checkpoint(self):
        raise NotImplementedError()

    def set_optimizations(self):
        cfg = self.eval_cfg
        if getattr(cfg, "vae_slicing", False):
            self.pipe.enable_vae_slicing()
        if getattr(cfg, "upcast_vae", False):
            self.pipe.upcast_vae()
        if getattr(cfg, "fused_projections", False):
            self.pipe.fuse_qkv_projections(vae=False)
        if getattr(cfg, "enable_xformers_memory_efficient_attention", False):
            self.pipe.enable_xformers_memory_efficient_attention()
        self.pipe.set_progress_bar_config(disable=(not getattr(cfg, "show_diffusion_progress", False)))
        if getattr(cfg, "torch_compile", False):
            self.pipe.enable_hip_graph(True)  # torch.compile(mode="reduce-overhead") == CUDA graphs -> one hipGraph per call
        self.pipe.to(cfg.device)

    def common_setup(self):
        r = self.eval_cfg.image_resolution
        self.transform_to_resolution = _ResizeCenterCrop(r)
        self.transform_to_half_resolution = _ResizeCenterCrop(r // 2)

    def infer(self, *args, **kwargs):
        raise NotImplementedError()


class SDControlNetAgent(DiffusionAgent):
    """SD-Turbo + ControlNet agent (controller/agent/sd_controlnet_agent.py:12-76)."""

    def load_checkpoint(self):
        cfg = self.eval_cfg
        ckpt = cfg.diffusion_ckpt
        controlnet = None
        if ckpt and os.path.isdir(ckpt):
            dirs = sorted([d for d in os.listdir(ckpt) if "checkpoint" in d], key=_natural_key)
            cn_dir = os.path.join(ckpt, dirs[-1], "controlnet") if dirs else ckpt
            if os.path.exists(os.path.join(cn_dir, "config.json")):
                controlnet = ControlNetModel.from_pretrained(cn_dir)
        if cfg.sd_ckpt and os.path.isdir(str(cfg.sd_ckpt)):
            self.pipe = StableDiffusionControlNetPipeline.from_pretrained(cfg.sd_ckpt, controlnet=controlnet,
                                                                          safety_checker=getattr(cfg, "safety_checker", None))
        elif str(cfg.sd_ckpt).startswith("synthetic:"):  # e.g. "synthetic:sd-turbo" / "synthetic:tiny" (no checkpoints offline)
            self.pipe = StableDiffusionControlNetPipeline.from_synthetic(configs.family(str(cfg.sd_ckpt).split(":", 1)[1]))
            if controlnet is not None:
                self.pipe.controlnet = controlnet
        else:
            raise FileNotFoundError(f"sd_ckpt {cfg.sd_ckpt!r} is not a local diffusers directory (no network access); "
                                    "use a local path or 'synthetic:<family>'")

    def infer(self, *args, **kwargs):
        return self.pipe(prompt=kwargs["prompts"], image=kwargs["images"], negative_prompt=kwargs.get("negative_prompts"),
                         num_inference_steps=kwargs["num_inference_steps"], guidance_scale=kwargs["guidance_scale"],
                         generator=kwargs.get("generator"))


class SDXLControlNetAgent(SDControlNetAgent):
    """SDXL-Turbo + ControlNet agent (controller/agent/sdxl_controlnet_agent.py:11-76): same checkpoint resolution and ``infer``
    contract; the pipeline class carries the SDXL deltas.  ``autoencoder: taesdxl`` (AutoencoderTiny) is not built."""

    def load_checkpoint(self):
        from .pipeline import StableDiffusionXLControlNetPipeline

        cfg = self.eval_cfg
        if "taesdxl" in str(getattr(cfg, "autoencoder", "")):
            raise NotImplementedError("AutoencoderTiny (taesdxl) is not built on the HIP path; use the SDXL AutoencoderKL")
        ckpt = cfg.diffusion_ckpt
        controlnet = None
        if ckpt and os.path.isdir(ckpt):
            dirs = sorted([d for d in os.listdir(ckpt) if "checkpoint" in d], key=_natural_key)
            cn_dir = os.path.join(ckpt, dirs[-1], "controlnet") if dirs else ckpt
            if os.path.exists(os.path.join(cn_dir, "config.json")):
                controlnet = ControlNetModel.from_pretrained(cn_dir)
        if cfg.sd_ckpt and os.path.isdir(str(cfg.sd_ckpt)):
            self.pipe = StableDiffusionXLControlNetPipeline.from_pretrained(cfg.sd_ckpt, controlnet=controlnet)
        elif str(cfg.sd_ckpt).startswith("synthetic:"):
            self.pipe = StableDiffusionXLControlNetPipeline.from_synthetic(configs.family(str(cfg.sd_ckpt).split(":", 1)[1]))
            if controlnet is not None:
                self.pipe.controlnet = controlnet
        else:
            raise FileNotFoundError(f"sd_ckpt {cfg.sd_ckpt!r} is not a local diffusers directory (no network access); "
                                    "use a local path or 'synthetic:<family>'")
