"""InstructPix2Pix family on libgenima_hip.so (reference controller/agent/sd_pix2pix_agent.py,
diffusion/train_instruct_pix2pix_genima.py): the alternative Genima base without a ControlNet -- an SD UNet whose ``conv_in`` takes
8 channels (noisy latents | VAE latents of the observation), fine-tuned as a whole.

  * ``expand_conv_in``                           the 4 -> 8 channel ``conv_in`` surgery (train_instruct_pix2pix_genima.py:800-818)
  * ``StableDiffusionInstructPix2PixPipeline``   diffusers' call surface as ``SDPix2PixAgent.infer`` uses it (sd_pix2pix_agent.py:51-60):
                                                 CLIP text -> VAE encode(image).mode() -> N x (cat -> UNet -> [3-way guidance] -> step) -> VAE decode
  * ``InstructPix2PixTrainer``                   the step body :1165-1273 -- VAE encode (sample / mode), DDPM noise, conditioning dropout,
                                                 trainable 8-channel UNet forward + backward on the tape, clip + AdamW, EMA (``--use_ema``)

Everything numeric runs in the HIP library (same kernels as the ControlNet path + ``gn_scale_cat_pad`` / ``gn_ema_flat``); there is no
CPU fallback.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import graphs, schema
from . import train_ops as T
from ._lib import GenimaHipError
from .engine import Engine
from .host import AutoencoderKL, CLIPTextModel, UNet2DConditionModel
from .pipeline import PipelineOutput, StableDiffusionControlNetPipeline, _load_tokenizer, randn_latents
from .scheduler import EulerDiscreteScheduler
from .training import ControlNetTrainer, Graph, TrainParams, pad_context, t_unet_full

F16, F32 = torch.float16, torch.float32


def expand_conv_in(unet_sd: Dict[str, torch.Tensor], in_channels: int = 8) -> "OrderedDict[str, torch.Tensor]":
    """``conv_in`` [C, 4, 3, 3] -> [C, in_channels, 3, 3]: the pretrained weights in the first input channels, zeros in the added ones
    (diffusion/train_instruct_pix2pix_genima.py:800-818); the bias is ``nn.Conv2d``'s fresh init there -- kept as it is here, callers
    that want the reference's draw overwrite it."""
    out = OrderedDict(unet_sd)
    w = unet_sd["conv_in.weight"]
    if w.shape[1] == in_channels:
        return out
    assert w.shape[1] < in_channels, (w.shape, in_channels)
    nw = torch.zeros((w.shape[0], in_channels) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
    nw[:, : w.shape[1]] = w
    out["conv_in.weight"] = nw
    return out


class StableDiffusionInstructPix2PixPipeline(StableDiffusionControlNetPipeline):
    """``StableDiffusionInstructPix2PixPipeline.__call__`` of diffusers 0.29 as the reference's agent calls it (prompt, image,
    negative_prompt, num_inference_steps, guidance_scale, generator; ``image_guidance_scale`` defaults to 1.5):

        do_cfg        = guidance_scale > 1 and image_guidance_scale >= 1
        prompt rows   = [prompt, negative, negative]                     (cfg)
        image latents = vae.encode(2 * image - 1).latent_dist.mode()     NOT multiplied by scaling_factor; [lat, lat, 0] under cfg
        each step     : eps = unet(cat([scale_model_input(latents)] * 3 | image latents, dim=1), t, prompt rows)
                        eps = uncond + g * (text - image) + ig * (image - uncond)
        images        = vae.decode(latents / scaling_factor)

    diffusers routes the guidance combination of sigma-space schedulers through the predicted original sample and back; the three weights
    (1 - ig, g, ig - g) sum to one, so that detour is the identity on eps and the combination is applied to eps directly."""

    def __init__(self, vae: AutoencoderKL, text_encoder: CLIPTextModel, tokenizer, unet: UNet2DConditionModel, scheduler,
                 safety_checker=None, feature_extractor=None, requires_safety_checker=False):
        super().__init__(vae, text_encoder, tokenizer, unet, None, scheduler, safety_checker)
        lat = vae.config["latent_channels"]
        if unet.config["in_channels"] != 2 * lat:
            raise ValueError(f"InstructPix2Pix needs unet.in_channels == latent_channels + image latent channels = {2 * lat}, "
                             f"got {unet.config['in_channels']}")

    @classmethod
    def from_pretrained(cls, path, unet=None, safety_checker=None, torch_dtype=None, variant=None, allow_hash_tokenizer: bool = False, **kw):
        import json

        if unet is None:
            unet = UNet2DConditionModel.from_pretrained(path, "unet", variant=variant)
        vae = AutoencoderKL.from_pretrained(path, "vae", variant=variant)
        text = CLIPTextModel.from_pretrained(path, "text_encoder", variant=variant)
        with open(os.path.join(path, "scheduler", "scheduler_config.json")) as f:
            sched = EulerDiscreteScheduler.from_config(json.load(f))
        tok = _load_tokenizer(path, "tokenizer", allow_hash_tokenizer, text.config["vocab_size"])
        return cls(vae, text, tok, unet, sched, safety_checker)

    @classmethod
    def from_synthetic(cls, family: dict, seed: int = 0, gen_device="cpu"):
        unet = UNet2DConditionModel.from_config(family["unet"], seed + 1, gen_device)
        vae = AutoencoderKL.from_config(family["vae"], seed + 3, gen_device)
        text = CLIPTextModel.from_config(family["text"], seed + 4, gen_device)
        return cls(vae, text, None, unet, EulerDiscreteScheduler.from_config(family["scheduler"]))

    def _modules(self):
        return (self.vae, self.text_encoder, self.unet)

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            for m in self._modules():
                m.to(device)
            self.device = self.unet.device
            self._progs.clear()
        return self

    def _build(self, B: int, H: int, W: int, steps: int, guidance=None):
        """guidance = (guidance_scale, image_guidance_scale) or None."""
        dev = self.device
        E = Engine(dev, record=True)
        s = self.vae_scale_factor
        h, w = H // s, W // s
        L = self.tokenizer.model_max_length if hasattr(self.tokenizer, "model_max_length") else 77
        Cl = self.vae.config["latent_channels"]
        io = SimpleNamespace()
        Bn = 3 * B if guidance else B
        io.ids = E.buf("in_ids", (Bn, L), dtype=torch.int32, zero=True)
        io.image_u8 = E.buf("in_image", (B, H, W, 3), dtype=torch.uint8, zero=True)
        io.noise = E.buf("in_noise", (B, h, w, Cl), zero=True)
        io.latents = E.buf("latents", (B, h, w, Cl), zero=True)
        sch = self.scheduler
        sch.set_timesteps(steps)
        io.timesteps = [int(t) for t in sch.timesteps.tolist()]
        if getattr(sch, "ancestral", False) or getattr(sch, "sampler", "euler") != "euler":
            raise NotImplementedError("the InstructPix2Pix pipeline is built for the Euler sampler of the SD-Turbo base (sd_pix2pix_agent.py:36-41)")
        E.scale_pad(io.noise, sch.init_noise_sigma, Cl, out=io.latents)
        img8 = E.image_u8_to_f16(io.image_u8, 8, 2.0, -1.0, name="img8")  # VaeImageProcessor.preprocess: [0, 1] -> [-1, 1]
        mom = graphs.emit_vae_encode_moments(E, self.vae.W, self.vae.config, img8)  # latent_dist.mode() = the mean = channels [0, Cl)
        ctx = graphs.emit_clip_text(E, self.text_encoder.W, self.text_encoder.config, io.ids)
        kv = graphs.emit_cross_kv(E, self.unet.W, ctx, "unet")

        def scalar(v):
            t = torch.full((B,), float(v), dtype=torch.float32, device=dev)
            E._keepalive(t)
            return t
        io.first_step_op = E.num_ops
        for i in range(steps):
            sigma, sigma_next = float(sch.sigmas[i]), float(sch.sigmas[i + 1])
            t_dev = torch.full((Bn,), float(sch.timesteps[i]), dtype=torch.float32, device=dev)
            E._keepalive(t_dev)
            x8 = E.buf("x8", (Bn, h, w, 8))
            E.scale_cat_pad(io.latents, Cl, mom, Cl, 8, sch.input_scale(i), 1.0, out=x8[:B])        # text row:  latents | image latents
            if guidance:
                E.scale_cat_pad(io.latents, Cl, mom, Cl, 8, sch.input_scale(i), 1.0, out=x8[B:2 * B])  # image row: same input, negative prompt
                E.scale_cat_pad(io.latents, Cl, mom, Cl, 8, sch.input_scale(i), 0.0, out=x8[2 * B:])   # uncond row: zero image latents
            eps = graphs.emit_unet(E, self.unet.W, self.unet.config, x8, t_dev, kv)
            if guidance:  # uncond + g (text - image) + ig (image - uncond) = (1 - ig) uncond + g text + (ig - g) image
                g, ig = guidance
                part = E.add_noise(eps[2 * B:], eps[:B], scalar(1.0 - ig), scalar(g), name="eps_cfg_a")
                eps = E.add_noise(part, eps[B:2 * B], scalar(1.0), scalar(ig - g), name="eps_cfg")
            E.euler_step(io.latents, eps, sigma, sigma_next)
            if i == 0:
                io.ops_per_step = E.num_ops - io.first_step_op
        io.first_vae_op = E.num_ops
        z8 = E.scale_pad(io.latents, 1.0 / self.vae.config["scaling_factor"], 8, name="z8")
        img = graphs.emit_vae_decode(E, self.vae.W, self.vae.config, z8)
        io.out_u8 = E.image_f16_to_u8(img, name="out_u8")
        io.engine = E
        from .engine import save_tune_table

        save_tune_table()
        if self.use_graph:
            side = torch.cuda.Stream(device=dev)
            E.use_stream(side)
            with torch.cuda.stream(side):
                E.run()
                side.synchronize()
                E.capture()
            io.stream = side
        return io

    def __call__(self, prompt=None, image=None, negative_prompt=None, num_inference_steps: int = 100, guidance_scale: float = 7.5,
                 image_guidance_scale: float = 1.5, generator=None, latents: Optional[torch.Tensor] = None, output_type: str = "pil",
                 prompt_ids=None, return_dict: bool = True, **kw):
        guidance = (float(guidance_scale), float(image_guidance_scale)) if (guidance_scale > 1.0 and image_guidance_scale >= 1.0) else None
        if prompt_ids is None:
            prompt_ids = self.encode_ids(prompt)
        B = prompt_ids.shape[0]
        img_u8 = self._images_to_u8(image, B)
        assert img_u8.shape[0] == B, "one input image per prompt"
        H, W = int(img_u8.shape[1]), int(img_u8.shape[2])
        if guidance:  # prompt_embeds = cat([prompt, negative, negative])
            neg = "" if negative_prompt is None else negative_prompt
            neg_ids = self.encode_ids([neg] * B if isinstance(neg, str) else list(neg))
            assert neg_ids.shape == prompt_ids.shape, "one negative prompt per prompt"
            prompt_ids = torch.cat([prompt_ids, neg_ids, neg_ids], dim=0)
        io = self.program(B, H, W, num_inference_steps, guidance)
        E: Engine = io.engine
        s = self.vae_scale_factor
        C = self.vae.config["latent_channels"]
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        if latents is None:
            latents = randn_latents((B, C, H // s, W // s), generator, self.device)
        else:
            latents = latents.to(self.device, torch.float16)
        if sch.draws_step_noise and generator is not None:
            for _ in range(num_inference_steps):  # diffusers 0.29's per-step (unused) randn draw of the Euler scheduler
                randn_latents((B, C, H // s, W // s), generator, self.device)
        stream = getattr(io, "stream", None)
        cur = torch.cuda.current_stream(self.device)
        io.ids.copy_(prompt_ids.to(torch.int32))
        io.image_u8.copy_(img_u8)
        io.noise.copy_(latents.permute(0, 2, 3, 1))
        if stream is not None:
            stream.wait_stream(cur)
            E.launch()
            cur.wait_stream(stream)
        else:
            E.use_stream(cur)
            E.run()
        out = io.out_u8
        if output_type == "latent":
            images = io.latents.permute(0, 3, 1, 2).clone()
        elif output_type in ("pt", "np_u8"):
            images = out.clone()
        else:
            arr = out.cpu().numpy()
            if output_type == "np":
                images = arr
            else:
                from PIL import Image

                images = [Image.fromarray(a) for a in arr]
        return PipelineOutput(images) if return_dict else (images, None)


def ema_decay_at(optimization_step: int, decay: float = 0.9999, min_decay: float = 0.0, update_after_step: int = 0,
                 use_ema_warmup: bool = False, inv_gamma: float = 1.0, power: float = 2.0 / 3.0) -> float:
    """diffusers ``EMAModel.get_decay`` (training_utils.py, 0.29): 0 until ``update_after_step``, then (1 + n) / (10 + n) (or the warm-up
    curve), clamped to [min_decay, decay]."""
    step = max(0, optimization_step - update_after_step - 1)
    if step <= 0:
        return 0.0
    cur = 1.0 - (1.0 + step / inv_gamma) ** -power if use_ema_warmup else (1.0 + step) / (10.0 + step)
    return max(min(cur, decay), min_decay)


class InstructPix2PixTrainer(ControlNetTrainer):
    """The optimisation step of diffusion/train_instruct_pix2pix_genima.py on one GPU: ``unet`` (8-channel ``conv_in``) is the trainable
    network -- it lives in ``self.cn`` (flat fp32 master / gradient / Adam moments + f16 working copy), so global-norm clipping, AdamW, loss
    scaling, accumulation, the data-parallel gradient exchange and the checkpoint layout are the base class's; the checkpoint subfolder is
    ``unet`` (+ ``unet_ema`` with ``use_ema``, :846-856)."""

    trainable_subfolder = "unet"

    def __init__(self, E: Engine, unet_cfg, unet_sd, *, use_ema: bool = False, ema_decay: float = 0.9999,
                 conditioning_dropout_prob: Optional[float] = None, **kw):
        if unet_cfg["in_channels"] != 8:
            raise ValueError("InstructPix2Pix trains a UNet with in_channels = 8 (expand_conv_in)")
        super().__init__(E, unet_cfg, unet_cfg, OrderedDict(), unet_sd, **kw)
        self.use_ema, self.ema_decay_max, self.ema_steps = bool(use_ema), float(ema_decay), 0
        self.ema = self.cn.master.clone() if use_ema else None  # EMAModel(unet.parameters()): the shadow starts as a copy (:821-824)
        self.cdp = conditioning_dropout_prob
        self.null_ids, self._null_ctx = None, None  # tokenize_captions([""]) of the reference (:1213-1215): set_null_prompt(ids)
        self.vae_W = None

    def set_null_prompt(self, input_ids: torch.Tensor):
        """Token ids [1, 77] of the empty prompt, as the run's tokenizer produces them (conditioning dropout's ``null_conditioning``)."""
        self.null_ids, self._null_ctx = input_ids.reshape(1, -1).to(torch.int32), None

    def _trainable_schema(self):
        return schema.unet_schema(self.cn_cfg)

    # ---- forward + backward
    def forward_backward(self, latents8, noise8, t_dev, sqrt_ac, sqrt_1mac, ctx, image_latents8, c_valid: int = 4, added=None):
        """latents8 / noise8 f16 [B, h, w, 8] (4 valid channels), ctx f16 [B, L, D], image_latents8: f16 [B, h, w, >= 4] whose first 4
        channels are the (masked) ``original_image_embeds``: ``concatenated_noisy_latents = cat([noisy, image_embeds], dim=1)`` (:1236-1239)."""
        E = self.E
        g = Graph(E)
        noisy = E.add_noise(latents8, noise8, sqrt_ac, sqrt_1mac)
        x8 = E.scale_cat_pad(noisy, c_valid, image_latents8, c_valid, 8)
        ctx_pad, L = pad_context(ctx), ctx.shape[1]
        pred = t_unet_full(g, self.cn, self.cn_cfg, x8, t_dev, ctx_pad, L, added)
        loss, dpred = T.mse_loss(E, pred.t, noise8, c_valid, grad_scale=self.loss_scale / self.grad_accum)
        pred.cell[0] = dpred
        buckets = self.allreduce if hasattr(self.allreduce, "begin") else None
        if buckets is not None and self._will_sync():
            buckets.begin(self.cn.grad, self.cn.layout, g.first_use, len(g.tape))
            g.on_entry_done = buckets.entry_done
            g.fire_indices = getattr(buckets, "fire_indices", None)
        self._side_wgrad(g)
        g.backward()
        self.last["pred"] = pred.t
        return loss

    def step(self, *args, **kw):
        loss = super().step(*args, **kw)
        if self.use_ema and self.sync_gradients:  # ema_unet.step(unet.parameters()) after every synced step (:1269-1272)
            self.ema_steps += 1
            T.ema_flat(self.E, self.ema, self.cn.master, 1.0 - ema_decay_at(self.ema_steps, self.ema_decay_max))
        return loss

    # ---- checkpoints
    def ema_state_dict(self):
        from .packing import unpack_state_dict

        packed = OrderedDict()
        for name, (o, shape) in self.cn.layout.items():
            n = 1
            for d in shape:
                n *= d
            packed[name] = self.ema[o:o + n].view(shape)
        return unpack_state_dict(packed, self._trainable_schema(), self.cn.temb_slices)

    def save_state(self, output_dir: str, global_step: int) -> str:
        from safetensors.torch import save_file

        from . import weights

        d = super().save_state(output_dir, global_step)
        if self.use_ema:
            weights.save_diffusers_dir(os.path.join(d, "unet_ema"), dict(self.cn_cfg), self.ema_state_dict(), torch.float32)
            save_file({"ema_steps": torch.tensor([self.ema_steps], dtype=torch.int64)}, os.path.join(d, "ema_flat.safetensors"))
        return d

    def load_state(self, checkpoint_dir: str) -> int:
        from safetensors.torch import load_file

        from . import weights

        step = super().load_state(checkpoint_dir)
        if self.use_ema:
            _, sd = weights.load_diffusers_dir(os.path.join(checkpoint_dir, "unet_ema"))
            sd = OrderedDict((k, sd[k]) for k in self._trainable_schema())
            self.ema.copy_(TrainParams(self.E, sd).master)
            self.ema_steps = int(load_file(os.path.join(checkpoint_dir, "ema_flat.safetensors"))["ema_steps"][0])
        return step

    def copy_ema_to_unet(self):
        """``ema_unet.copy_to(unet.parameters())`` before the final save (:1347-1348)."""
        self.cn.master.copy_(self.ema)
        self.cn.sync_half()

    # ---- the whole step body from a collated batch
    def train_step(self, batch) -> torch.Tensor:
        """batch: ``original_pixel_values`` / ``edited_pixel_values`` (NCHW in [-1, 1], or NHWC f16 8-channel) and ``input_ids`` [b, 77]
        -- the collate_fn output of diffusion/train_instruct_pix2pix_genima.py:1018-1037.  RNG draws in the reference's order: posterior
        sample, noise, timesteps, then the dropout's ``random_p``."""
        E, dev = self.E, self.E.device
        if self.vae_W is None:
            raise GenimaHipError("attach_frozen(...) first")
        args = self._on_front_stream(lambda: self._front_p2p(batch), batch)  # (ControlNetTrainer: the front of the step on its own stream)
        loss = self.step(*args)
        self._steps_seen += 1
        return loss

    def _front_p2p(self, batch):
        E, dev = self.E, self.E.device
        edited8 = self._nhwc8(batch["edited_pixel_values"])
        orig8 = self._nhwc8(batch["original_pixel_values"])
        ids = batch["input_ids"].to(dev, torch.int32).contiguous()
        B = edited8.shape[0]
        Cl = self.vae_cfg["latent_channels"]
        mom = graphs.emit_vae_encode_moments(E, self.vae_W, self.vae_cfg, edited8)
        shape = tuple(mom.shape[:-1]) + (Cl,)
        lat8 = T.latent_sample(E, mom, torch.randn(shape, generator=self._gen_dev, device=dev, dtype=F32).to(F16), Cl,
                               self.vae_cfg.get("scaling_factor", 0.18215))
        noise8 = E.scale_pad(torch.randn(shape, generator=self._gen_dev, device=dev, dtype=F32).to(F16), 1.0, 8)
        t = torch.randint(0, int(self.noise_scheduler.config.num_train_timesteps), (B,), generator=self._gen_cpu)
        sa, s1 = self.noise_scheduler.add_noise_coeffs(t)
        ctx = graphs.emit_clip_text(E, self.text_W, self.text_cfg, ids)
        img_mom = graphs.emit_vae_encode_moments(E, self.vae_W, self.vae_cfg, orig8)  # latent_dist.mode(): the mean, unscaled (:1197-1200)
        if self.cdp is not None:
            random_p = torch.rand(B, generator=self._gen_cpu)
            ctx, img_mom = self.apply_conditioning_dropout(ctx, img_mom, random_p)
        return lat8, noise8, t.to(dev, F32), sa.to(dev), s1.to(dev), ctx, img_mom

    def apply_conditioning_dropout(self, ctx, img_mom, random_p: torch.Tensor):
        """:1204-1233 -- prompt rows with random_p < 2p become the encoding of ``""``; image latents are zeroed where p <= random_p < 3p
        is FALSE ... i.e. kept unless random_p is in [p, 3p) -- ``image_mask = 1 - (random_p >= p) * (random_p < 3p)``."""
        E, dev, p = self.E, self.E.device, float(self.cdp)
        B = ctx.shape[0]
        if self._null_ctx is None:
            if self.null_ids is None:
                raise GenimaHipError("conditioning dropout needs the empty prompt's token ids: call set_null_prompt(tokenizer('').input_ids)")
            self._null_ctx = graphs.emit_clip_text(E, self.text_W, self.text_cfg, self.null_ids.to(dev).contiguous()).clone()
        prompt_mask = (random_p < 2 * p).to(F32)
        image_mask = 1.0 - ((random_p >= p).to(F32) * (random_p < 3 * p).to(F32))
        null = self._null_ctx.expand(B, -1, -1).contiguous()
        # torch.where(mask, null, ctx) with a {0, 1} mask = (1 - m) * ctx + m * null, exact in f16 for finite inputs
        ctx = E.add_noise(ctx, null, (1.0 - prompt_mask).to(dev), prompt_mask.to(dev))
        img_mom = E.add_noise(img_mom, img_mom, image_mask.to(dev), torch.zeros(B, dtype=F32, device=dev))
        return ctx, img_mom
