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


def _resolve_controlnet(eval_cfg):
    """``<diffusion_ckpt>/checkpoint-<max>/controlnet`` by natural sort, else ``<diffusion_ckpt>`` itself
    (controller/agent/sd_controlnet_agent.py:21-35).  Like the reference, a path that holds no ControlNet is an error -- never a
    silently zero-initialised network.  Only the explicit synthetic setup (``sd_ckpt: 'synthetic:<family>'`` with an empty or
    ``'synthetic:'`` ``diffusion_ckpt``) keeps the family's seeded synthetic ControlNet."""
    ckpt = eval_cfg.diffusion_ckpt
    if str(eval_cfg.sd_ckpt).startswith("synthetic:") and (not ckpt or str(ckpt).startswith("synthetic:")):
        return None
    if not ckpt or not os.path.isdir(str(ckpt)):
        raise FileNotFoundError(f"diffusion_ckpt {ckpt!r} is not a directory (expected <dir>/checkpoint-N/controlnet or a ControlNet directory)")
    dirs = sorted([d for d in os.listdir(ckpt) if "checkpoint" in d], key=_natural_key)
    cn_dir = os.path.join(ckpt, dirs[-1], "controlnet") if dirs else ckpt
    if not os.path.exists(os.path.join(cn_dir, "config.json")):
        raise FileNotFoundError(f"no ControlNet checkpoint under {cn_dir} (config.json + diffusion_pytorch_model.safetensors)")
    return ControlNetModel.from_pretrained(cn_dir)


def _load_tiny_autoencoder(name: str):
    from .host import AutoencoderTiny

    if os.path.isdir(name):
        return AutoencoderTiny.from_pretrained(name)
    if name.startswith("synthetic:"):
        return AutoencoderTiny.from_config(configs.TAESD, seed=7)
    raise FileNotFoundError(f"autoencoder {name!r} is not a local AutoencoderTiny directory (no network access)")


class SDControlNetAgent(DiffusionAgent):
    """SD-Turbo + ControlNet agent (controller/agent/sd_controlnet_agent.py:12-76)."""

    pipeline_cls = StableDiffusionControlNetPipeline
    tiny_tag = "taesd"

    def load_checkpoint(self):
        cfg = self.eval_cfg
        controlnet = _resolve_controlnet(cfg)
        if cfg.sd_ckpt and os.path.isdir(str(cfg.sd_ckpt)):
            self.pipe = self.pipeline_cls.from_pretrained(cfg.sd_ckpt, controlnet=controlnet, variant="fp16",
                                                          safety_checker=getattr(cfg, "safety_checker", None))
        elif str(cfg.sd_ckpt).startswith("synthetic:"):  # e.g. "synthetic:sd-turbo" / "synthetic:tiny" (no checkpoints offline)
            self.pipe = self.pipeline_cls.from_synthetic(configs.family(str(cfg.sd_ckpt).split(":", 1)[1]))
            if controlnet is not None:
                self.pipe.controlnet = controlnet
        else:
            raise FileNotFoundError(f"sd_ckpt {cfg.sd_ckpt!r} is not a local diffusers directory (no network access); "
                                    "use a local path or 'synthetic:<family>'")
        autoencoder = str(getattr(cfg, "autoencoder", "") or "")
        if self.tiny_tag in autoencoder:  # "taesd" also matches "taesdxl", exactly as the reference's substring test does
            self.pipe.vae = _load_tiny_autoencoder(autoencoder)

    def infer(self, *args, **kwargs):
        return self.pipe(prompt=kwargs["prompts"], image=kwargs["images"], negative_prompt=kwargs.get("negative_prompts"),
                         num_inference_steps=kwargs["num_inference_steps"], guidance_scale=kwargs["guidance_scale"],
                         generator=kwargs.get("generator"))


class SDXLControlNetAgent(SDControlNetAgent):
    """SDXL-Turbo + ControlNet agent (controller/agent/sdxl_controlnet_agent.py:11-76): same checkpoint resolution and ``infer``
    contract; the pipeline class carries the SDXL deltas; ``autoencoder: taesdxl`` swaps in the AutoencoderTiny decoder (:44-49)."""

    from .pipeline import StableDiffusionXLControlNetPipeline as pipeline_cls  # noqa: E402

    tiny_tag = "taesdxl"


class SDPix2PixAgent(DiffusionAgent):
    """InstructPix2Pix agent (controller/agent/sd_pix2pix_agent.py:11-60): the fine-tuned 8-channel UNet from
    ``<diffusion_ckpt>/checkpoint-<max>/unet`` (natural sort; else ``<diffusion_ckpt>/unet``) inside the base checkpoint's pipeline."""

    def load_checkpoint(self):
        from .host import UNet2DConditionModel
        from .pix2pix import StableDiffusionInstructPix2PixPipeline

        cfg = self.eval_cfg
        ckpt = cfg.diffusion_ckpt
        synthetic = str(cfg.sd_ckpt).startswith("synthetic:")
        unet = None
        if not (synthetic and (not ckpt or str(ckpt).startswith("synthetic:"))):
            if not ckpt or not os.path.isdir(str(ckpt)):
                raise FileNotFoundError(f"diffusion_ckpt {ckpt!r} is not a directory (expected <dir>/checkpoint-N/unet or <dir>/unet)")
            dirs = sorted([d for d in os.listdir(ckpt) if "checkpoint" in d], key=_natural_key)
            root = os.path.join(ckpt, dirs[-1]) if dirs else ckpt
            if not os.path.exists(os.path.join(root, "unet", "config.json")):
                raise FileNotFoundError(f"no InstructPix2Pix UNet under {root}/unet (config.json + diffusion_pytorch_model.safetensors)")
            unet = UNet2DConditionModel.from_pretrained(root, "unet")
        if cfg.sd_ckpt and os.path.isdir(str(cfg.sd_ckpt)):
            self.pipe = StableDiffusionInstructPix2PixPipeline.from_pretrained(cfg.sd_ckpt, unet=unet, variant="fp16")
        elif synthetic:
            self.pipe = StableDiffusionInstructPix2PixPipeline.from_synthetic(configs.family(str(cfg.sd_ckpt).split(":", 1)[1]))
            if unet is not None:
                self.pipe.unet = unet
        else:
            raise FileNotFoundError(f"sd_ckpt {cfg.sd_ckpt!r} is not a local diffusers directory (no network access); "
                                    "use a local path or 'synthetic:<family>'")

    def infer(self, *args, **kwargs):
        return self.pipe(prompt=kwargs["prompts"], image=kwargs["images"], negative_prompt=kwargs.get("negative_prompts"),
                         num_inference_steps=kwargs["num_inference_steps"], guidance_scale=kwargs["guidance_scale"],
                         generator=kwargs.get("generator"))
