"""Host-side mirrors of the module surfaces the reference touches on diffusers / transformers
(SURVEY.md section 8b "Module surface used by the trainer" and "Pipeline call surface").

``UNet2DConditionModel`` / ``ControlNetModel`` / ``AutoencoderKL`` / ``CLIPTextModel`` keep the reference's call
signatures (NCHW fp16 tensors in, diffusers-style outputs) and checkpoint layout (``from_pretrained`` / ``save_pretrained``
on ``config.json`` + safetensors with diffusers key names), but every FLOP runs in libgenima_hip.so through an ``Engine``;
there is no torch compute fallback.  Weights are an fp32 master state dict (what the trainer's optimizer owns) plus an
f16 packed device copy (``packing.pack_state_dict``), re-packed explicitly by ``.to(device)`` / ``load_state_dict``.
"""
from __future__ import annotations

import copy
import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch

from . import configs, graphs, packing, schema, weights
from .engine import Engine
from ._lib import GenimaHipError


class FrozenConfig(dict):
    """dict with attribute access (diffusers ``FrozenDict`` behaviour the reference relies on: ``vae.config.scaling_factor``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def nchw_to_nhwc(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    """Layout interop at the diffusers-shaped boundary (NCHW in, NHWC inside)."""
    y = x.permute(0, 2, 3, 1)
    if cpad is not None and cpad > y.shape[-1]:
        z = torch.zeros(tuple(y.shape[:-1]) + (cpad,), dtype=y.dtype, device=y.device)
        z[..., : y.shape[-1]] = y
        return z
    return y.contiguous()


def nhwc_to_nchw(x: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    if c is not None:
        x = x[..., :c]
    return x.permute(0, 3, 1, 2).contiguous()


class HipModule:
    schema_fn = None
    default_config = None
    weight_name = "diffusion_pytorch_model.safetensors"

    def __init__(self, config: dict, state_dict=None, seed: int = 0, gen_device="cpu"):
        self.config = FrozenConfig(copy.deepcopy(dict(config)))
        self._schema = type(self).schema_fn(self.config)
        if state_dict is None:
            state_dict = weights.synth_state_dict(self._schema, seed, device=gen_device)
        self._check(state_dict)
        self._sd = OrderedDict((k, state_dict[k].detach().to(torch.float32)) for k in self._schema)
        self.device = torch.device("cpu")
        self.dtype = torch.float16
        self.W = None
        self.training = False
        self._engine: Optional[Engine] = None
        self._pack_gen = 0  # bumped by every re-pack: recorded programs bake the packed tensors' addresses in (pipeline cache key)

    # ---- checkpoint / config surface --------------------------------------------------------------------------------
    def _check(self, sd):
        missing = [k for k in self._schema if k not in sd]
        if missing:
            raise KeyError(f"{type(self).__name__}: state dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in self._schema.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{type(self).__name__}: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")

    @classmethod
    def from_config(cls, config, seed: int = 0, gen_device="cpu"):
        """Random-init module with the seeded synthetic weights of weights.synth_state_dict (drawn on ``gen_device``)."""
        return cls(config, None, seed, gen_device)

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, torch_dtype=None, variant=None, **kw):
        """``variant`` selects the weight file as diffusers does: None = the full-precision file, "fp16" = ``*.fp16.safetensors``
        (the agents pass variant="fp16", the trainer None: controller/agent/sd_controlnet_agent.py:36-42,
        diffusion/train_controlnet_genima.py:1042-1064)."""
        cfg, sd = weights.load_diffusers_dir(path, subfolder, variant=variant)
        return cls(cfg, sd)

    def save_pretrained(self, path, **kw):
        weights.save_diffusers_dir(path, dict(self.config), self._sd, torch.float32, type(self).weight_name)

    def register_to_config(self, **kw):
        self.config.update(kw)

    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, sd, strict=True):
        self._check(sd)
        self._sd = OrderedDict((k, sd[k].detach().to(torch.float32).cpu()) for k in self._schema)
        if self.W is not None:
            self._pack()
        return SimpleNamespace(missing_keys=[], unexpected_keys=[k for k in sd if k not in self._schema])

    def parameters(self):
        return list(self._sd.values())

    def named_parameters(self):
        return list(self._sd.items())

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def requires_grad_(self, flag=True):
        return self

    # no-op speed knobs of the reference's call sites (native NHWC / flash attention are always on)
    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None

    def enable_gradient_checkpointing(self):
        return None

    def enable_slicing(self):
        return None

    def fuse_qkv_projections(self):
        return None

    # ---- device ---------------------------------------------------------------------------------------------------
    def _pack(self):
        self.W = packing.pack_state_dict(self._sd, self.device)
        self._pack_gen += 1

    def to(self, device=None, dtype=None, memory_format=None, **kw):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if device is not None:
            dev = torch.device(device)
            if dev.type == "cuda":
                if dev.index is None:
                    dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
                if not torch.cuda.is_available():
                    raise GenimaHipError("no ROCm device visible; the Genima HIP path has no CPU fallback")
                if self.W is None or dev != self.device:
                    self.device = dev
                    self._pack()
                    self._engine = None
            else:
                self.device, self.W, self._engine = dev, None, None
        return self

    def engine(self) -> Engine:
        if self.W is None:
            raise GenimaHipError(f"{type(self).__name__} is not on a ROCm device: call .to('cuda') first (no CPU fallback)")
        if self._engine is None:
            self._engine = Engine(self.device)
        return self._engine


def _t_dev(timestep, B, device) -> torch.Tensor:
    t = torch.as_tensor(timestep, dtype=torch.float32)
    if t.dim() == 0:
        t = t[None].expand(B)
    return t.to(device).contiguous()


def _added(added_cond_kwargs, device):
    """SDXL ``added_cond_kwargs={"text_embeds": [B, 1280], "time_ids": [B, 6]}`` (diffusion/train_controlnet_sdxl_genima.py:1448-1471)
    -> the (text_embeds f16, time_ids f32) device pair the lowering takes."""
    if not added_cond_kwargs:
        return None
    return (added_cond_kwargs["text_embeds"].to(device, torch.float16).contiguous(),
            added_cond_kwargs["time_ids"].to(device, torch.float32).contiguous())


class UNet2DConditionModel(HipModule):
    schema_fn = staticmethod(schema.unet_schema)

    def __call__(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, return_dict=True, added_cond_kwargs=None, **kw):
        """Call surface of diffusion/train_controlnet_genima.py:1377-1388 (NCHW tensors); ``added_cond_kwargs`` for SDXL."""
        E = self.engine()
        B = sample.shape[0]
        x8 = nchw_to_nhwc(sample.to(self.device, torch.float16), 8)
        ctx = encoder_hidden_states.to(self.device, torch.float16).contiguous()
        kv = graphs.emit_cross_kv(E, self.W, ctx, "unet")
        down = None
        if down_block_additional_residuals is not None:
            down = [nchw_to_nhwc(r.to(self.device, torch.float16)) for r in down_block_additional_residuals]
        mid = None if mid_block_additional_residual is None else nchw_to_nhwc(mid_block_additional_residual.to(self.device, torch.float16))
        eps = graphs.emit_unet(E, self.W, self.config, x8, _t_dev(timestep, B, self.device), kv, down, mid,
                               added=_added(added_cond_kwargs, self.device))
        out = nhwc_to_nchw(eps, self.config["out_channels"])
        return SimpleNamespace(sample=out) if return_dict else (out,)


class ControlNetModel(HipModule):
    schema_fn = staticmethod(schema.controlnet_schema)

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, conditioning_embedding_out_channels=(16, 32, 96, 256),
                  conditioning_channels: int = 3, load_weights_from_unet: bool = True, seed: int = 0):
        """``ControlNetModel.from_unet`` (diffusion/train_controlnet_genima.py:1071): encoder weights copied from the UNet,
        zero-initialised ``controlnet_*`` convs and cond-embedding ``conv_out``, fresh cond-embedding convs."""
        cfg = {k: v for k, v in unet.config.items() if k not in ("out_channels", "up_block_types")}
        cfg.update(_class_name="ControlNetModel", conditioning_channels=conditioning_channels,
                   conditioning_embedding_out_channels=list(conditioning_embedding_out_channels), global_pool_conditions=False)
        sch = schema.controlnet_schema(cfg)
        sd = weights.synth_state_dict(sch, seed)
        usd = unet.state_dict()
        for k in sch:
            if load_weights_from_unet and k in usd:
                sd[k] = usd[k].clone()
            elif k.startswith("controlnet_down_blocks") or k.startswith("controlnet_mid_block") \
                    or k.startswith("controlnet_cond_embedding.conv_out"):
                sd[k] = torch.zeros_like(sd[k])
        return cls(cfg, sd)

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, guess_mode=False,
                 return_dict=True, added_cond_kwargs=None, **kw):
        """Call surface of diffusion/train_controlnet_genima.py:1368-1374: -> (list of 12 (SDXL: 9) down residuals, mid residual), NCHW."""
        E = self.engine()
        B = sample.shape[0]
        x8 = nchw_to_nhwc(sample.to(self.device, torch.float16), 8)
        cond8 = nchw_to_nhwc(controlnet_cond.to(self.device, torch.float16), 8)
        ctx = encoder_hidden_states.to(self.device, torch.float16).contiguous()
        kv = graphs.emit_cross_kv(E, self.W, ctx, "cn")
        cemb = graphs.emit_controlnet_cond(E, self.W, self.config, cond8)
        outs, mid = graphs.emit_controlnet(E, self.W, self.config, x8, _t_dev(timestep, B, self.device), kv, cemb, conditioning_scale,
                                           added=_added(added_cond_kwargs, self.device))
        outs = [nhwc_to_nchw(o) for o in outs]
        mid = nhwc_to_nchw(mid)
        if return_dict:
            return SimpleNamespace(down_block_res_samples=outs, mid_block_res_sample=mid)
        return outs, mid


class DiagonalGaussianDistribution:
    def __init__(self, moments_nchw: torch.Tensor):
        self.mean, self.logvar = moments_nchw.float().chunk(2, dim=1)
        self.logvar = self.logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(HipModule):
    schema_fn = staticmethod(schema.vae_schema)
    stream_scale = 1.0  # != 1: the residual stream is carried scaled (enable_stream_scaling)

    def enable_stream_scaling(self, s: float = 1.0 / 64.0):
        """The f16 answer to diffusers' ``force_upcast`` / ``pipe.upcast_vae()`` (the stock SDXL VAE's residual stream exceeds f16's range,
        so diffusers decodes in fp32): keep f16 storage but carry the stream multiplied by ``s`` -- an exact re-parametrisation of the
        weights (packing.scale_vae_stream) with the stream GroupNorms' eps scaled by s^2.  Outputs are unchanged up to f16 rounding."""
        self.stream_scale = float(s)
        if self.W is not None:
            self._pack()
        return self

    def _pack(self):
        if self.stream_scale == 1.0:
            return super()._pack()
        self.W = packing.pack_state_dict(packing.scale_vae_stream(self._sd, self.stream_scale), self.device)
        self.W["__meta__"]["vae_stream_scale"] = self.stream_scale
        self._pack_gen += 1

    def decode(self, z, return_dict=True, **kw):
        E = self.engine()
        z8 = nchw_to_nhwc(z.to(self.device, torch.float16), 8)
        img = graphs.emit_vae_decode(E, self.W, self.config, z8)
        out = nhwc_to_nchw(img, self.config["out_channels"])
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def encode(self, x, return_dict=True):
        E = self.engine()
        x8 = nchw_to_nhwc(x.to(self.device, torch.float16), 8)
        m = graphs.emit_vae_encode_moments(E, self.W, self.config, x8)
        dist = DiagonalGaussianDistribution(nhwc_to_nchw(m, 2 * self.config["latent_channels"]))
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)


class AutoencoderTiny(HipModule):
    """diffusers ``AutoencoderTiny`` (TAESD): the fast decoder the agents swap in when ``autoencoder`` names a taesd checkpoint
    (controller/agent/sd_controlnet_agent.py:45-49, sdxl_controlnet_agent.py:44-49).  Decode only: the trainer encodes with the
    AutoencoderKL."""
    schema_fn = staticmethod(schema.taesd_schema)

    def _pack(self):
        super()._pack()
        last = max(int(k.split(".")[2]) for k in self._sd if k.startswith("decoder.layers.") and k.endswith(".bias"))
        b = self._sd[f"decoder.layers.{last}.bias"]
        self.W["decoder.out_bias_shifted"] = packing.pack_vec(b - 0.5, self.W[f"decoder.layers.{last}.weight"].shape[0]).to(self.device)

    def decode(self, z, return_dict=True, **kw):
        E = self.engine()
        z8 = nchw_to_nhwc(z.to(self.device, torch.float16), 8)
        img = graphs.emit_taesd_decode(E, self.W, self.config, z8)
        out = nhwc_to_nchw(img, self.config["out_channels"])
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def encode(self, x, return_dict=True):
        raise NotImplementedError("AutoencoderTiny.encode is not on the Genima hot path (the trainer encodes with AutoencoderKL, "
                                  "diffusion/train_controlnet_genima.py:1329-1332)")


class CLIPTextModel(HipModule):
    schema_fn = staticmethod(schema.clip_text_schema)
    weight_name = "model.safetensors"

    def __call__(self, input_ids, attention_mask=None, output_hidden_states=False, **kw):
        """``text_encoder(ids)[0]`` (train_controlnet_genima.py:1362); with ``output_hidden_states=True`` the SDXL call of
        train_controlnet_sdxl_genima.py:879-893: ``out[0]`` = last hidden state (or ``text_embeds`` for the projection tower),
        ``out[-1][-2]`` = penultimate hidden state (only the last two entries of the hidden-state list are materialised)."""
        E = self.engine()
        ids = input_ids.to(self.device, torch.int32).contiguous()
        if not output_hidden_states:
            return (graphs.emit_clip_text(E, self.W, self.config, ids),)
        hidden = []
        last = graphs.emit_clip_text(E, self.W, self.config, ids, hidden)
        first = last
        if "text_projection.weight" in self.W:
            first = E.linear(E.gather_rows(last, E.argmax_rows(ids)), self.W["text_projection.weight"])
        return (first, [None] * (len(hidden) - 1) + hidden[-2:])


class CLIPTextModelWithProjection(CLIPTextModel):
    """SDXL's second tower (OpenCLIP bigG): same encoder + ``text_projection`` on the EOT row."""
