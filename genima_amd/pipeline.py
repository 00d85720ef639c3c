"""``StableDiffusionControlNetPipeline``-shaped denoise pipeline on libgenima_hip.so.

Keeps the call surface the reference uses (controller/agent/sd_controlnet_agent.py:67-76,
diffusion/train_controlnet_genima.py:632-638; semantics in SURVEY.md Appendix D):
    pipe(prompt=, image=, negative_prompt=, num_inference_steps=, guidance_scale=, generator=)  ->  out.images / out[0]
but the whole call -- CLIP text encode, ControlNet cond-embedding, N x (ControlNet + UNet + Euler step), VAE decode, uint8
post-process -- is lowered ONCE per (batch, size, steps) shape to a recorded ``gn_program`` and replayed from C++ on one HIP
stream (optionally as a captured hipGraph), so nothing returns to Python inside the loop.  Work hoisted out of the step loop
because it is constant across steps: the cross-attention K/V projections of the prompt (23 layers) and the ControlNet
conditioning embedding.
"""
from __future__ import annotations

import os
import zlib
from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import graphs
from .engine import Engine
from .host import AutoencoderKL, AutoencoderTiny, CLIPTextModel, ControlNetModel, UNet2DConditionModel
from .scheduler import EulerDiscreteScheduler
from ._lib import GenimaHipError


def _load_tokenizer(path: str, subfolder: str, allow_hash: bool, vocab_size: int):
    """The checkpoint's CLIP BPE tokenizer (genima_amd/tokenizer.py).  A pipeline with REAL weights must not fall back to hashed
    ids -- the prompt embedding would be meaningless -- so a missing ``vocab.json`` / ``merges.txt`` raises unless the caller opts in
    (``allow_hash_tokenizer=True``: synthetic-weight benchmarks and tests)."""
    from .tokenizer import CLIPTokenizer

    try:
        return CLIPTokenizer.from_pretrained(path, subfolder)
    except FileNotFoundError:
        if allow_hash:
            return HashTokenizer(vocab_size)
        raise FileNotFoundError(f"{os.path.join(path, subfolder)} holds no CLIP BPE model (vocab.json + merges.txt): a checkpoint with real "
                                "weights needs its real tokenizer; pass allow_hash_tokenizer=True only for synthetic weights") from None


class HashTokenizer:
    """Stand-in tokenizer for SYNTHETIC weights only: the CLIP BPE vocab/merges are not available offline,
    so words are hashed to ids in [1000, vocab-3); BOS/EOS/pad ids and the pad-to-77 shape follow the SD-2.x tokenizer.
    ``from_pretrained`` loads the checkpoint's real BPE model (genima_amd/tokenizer.py) and refuses to fall back to this class."""

    model_max_length = 77

    def __init__(self, vocab_size: int = 49408, pad_token_id: int = 0):
        self.vocab_size, self.pad_token_id = vocab_size, pad_token_id
        self.bos_token_id, self.eos_token_id = vocab_size - 2, vocab_size - 1

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        L = max_length or self.model_max_length
        ids = np.full((len(texts), L), self.pad_token_id, dtype=np.int64)
        lo, hi = min(1000, self.vocab_size // 4), self.vocab_size - 3
        for i, t in enumerate(texts):
            toks = [self.bos_token_id] + [lo + zlib.crc32(w.encode()) % (hi - lo) for w in t.lower().split()][: L - 2]
            toks.append(self.eos_token_id)
            ids[i, : len(toks)] = toks
        return SimpleNamespace(input_ids=torch.from_numpy(ids))


class PipelineOutput:
    def __init__(self, images, nsfw_content_detected=None):
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected

    def __getitem__(self, i):
        return (self.images, self.nsfw_content_detected)[i]


def randn_latents(shape, generator, device, dtype=torch.float16) -> torch.Tensor:
    """diffusers ``randn_tensor`` semantics: with a list of generators sample i is drawn as
    (1, C, H, W) from generator[i] (the reference passes the SAME generator B times -> sequential draws from one stream)."""
    def draw(shp, g):
        gdev = g.device if g is not None else torch.device(device)
        return torch.randn(shp, generator=g, device=gdev, dtype=dtype).to(device)

    if isinstance(generator, (list, tuple)):
        if len(generator) == 1:
            generator = generator[0]
        else:
            assert len(generator) == shape[0], "one generator per sample"
            return torch.cat([draw((1,) + tuple(shape[1:]), g) for g in generator], dim=0)
    return draw(tuple(shape), generator)


class StableDiffusionControlNetPipeline:
    def __init__(self, vae: AutoencoderKL, text_encoder: CLIPTextModel, tokenizer, unet: UNet2DConditionModel,
                 controlnet: ControlNetModel, scheduler: EulerDiscreteScheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker=False):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer or HashTokenizer(text_encoder.config["vocab_size"])
        self.unet, self.controlnet, self.scheduler = unet, controlnet, scheduler
        self.safety_checker = safety_checker
        self.vae_scale_factor = 2 ** (len(vae.config["block_out_channels"]) - 1)
        self.device = torch.device("cpu")
        self._progs = {}
        self.use_graph = False
        self.two_streams = os.environ.get("GN_TWO_STREAMS", "1") != "0"  # ControlNet || UNet encoder inside each denoise step
        # the ControlNet's zero convs behind the join, the UNet's skip as their residual operand (no add launch), dealt over both streams at small batch
        self.zero_convs_fused = os.environ.get("GN_ZERO_CONV_FUSED", "1") != "0"
        self._progress = True

    # ---- construction ----------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, controlnet=None, safety_checker=None, torch_dtype=None, variant=None,
                        allow_hash_tokenizer: bool = False, **kw):
        """Reads the diffusers pipeline directory layout (``unet/``, ``vae/``, ``text_encoder/``, ``tokenizer/``, ``scheduler/``);
        ``variant`` selects ``*.fp16.safetensors`` as in the reference's call (controller/agent/sd_controlnet_agent.py:36-42)."""
        import json

        unet = UNet2DConditionModel.from_pretrained(path, "unet", variant=variant)
        vae = AutoencoderKL.from_pretrained(path, "vae", variant=variant)
        text = CLIPTextModel.from_pretrained(path, "text_encoder", variant=variant)
        with open(os.path.join(path, "scheduler", "scheduler_config.json")) as f:
            sched = EulerDiscreteScheduler.from_config(json.load(f))
        tok = _load_tokenizer(path, "tokenizer", allow_hash_tokenizer, text.config["vocab_size"])
        if controlnet is None:
            controlnet = ControlNetModel.from_unet(unet)
        return cls(vae, text, tok, unet, controlnet, sched, safety_checker)

    @classmethod
    def from_synthetic(cls, family: dict, seed: int = 0, gen_device="cpu"):
        """Random-init pipeline of a config family (no checkpoints exist offline; SURVEY.md section 8d synthetic weights)."""
        unet = UNet2DConditionModel.from_config(family["unet"], seed + 1, gen_device)
        cn = ControlNetModel.from_config(family["controlnet"], seed + 2, gen_device)
        vae = AutoencoderKL.from_config(family["vae"], seed + 3, gen_device)
        text = CLIPTextModel.from_config(family["text"], seed + 4, gen_device)
        return cls(vae, text, None, unet, cn, EulerDiscreteScheduler.from_config(family["scheduler"]))

    # ---- knobs the agent calls (controller/agent/diffusion_agent.py:21-42) -------------------------------------------------
    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            for m in (self.vae, self.text_encoder, self.unet, self.controlnet):
                m.to(device)
            self.device = self.unet.device
            self._progs.clear()
        return self

    def set_progress_bar_config(self, disable=False, **kw):
        self._progress = not disable

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None  # flash attention on MFMA is the only attention path

    def fuse_qkv_projections(self, vae=False, **k):
        return None  # q|k are always fused at pack time

    def upcast_vae(self):
        """diffusers casts the VAE to fp32 here.  The HIP VAE stores activations in f16 and accumulates in f32: the SD-2.x VAE and
        the SDXL fp16-fix VAE are in range in that format; for a VAE whose residual stream is not (``force_upcast``) this switches the
        stream scaling on (host.AutoencoderKL.enable_stream_scaling), which is what the SDXL pipeline's from_pretrained does by itself."""
        if hasattr(self.vae, "enable_stream_scaling") and getattr(self.vae, "stream_scale", 1.0) == 1.0:
            self.vae.enable_stream_scaling()
            self._progs.clear() if hasattr(self, "_progs") else None
        return None

    def enable_vae_slicing(self):
        return None

    def enable_hip_graph(self, flag: bool = True):
        """Replay each call as one captured hipGraph (the HIP equivalent of the reference's torch.compile reduce-overhead)."""
        self.use_graph = flag
        self._progs.clear()

    # ---- program construction -----------------------------------------------------------------------------------------------
    def _emit_prompt(self, E: Engine, io, B: int, L: int, H: int, W: int):
        """-> (cross-attention context [B, L, D], added conditions or None).  SD-2.x: the single tower's last hidden state."""
        return graphs.emit_clip_text(E, self.text_encoder.W, self.text_encoder.config, io.ids), None

    def _fill_prompt_inputs(self, io, kw):
        """Per-call inputs beyond ids / image / noise (SDXL: the second tower's ids)."""

    def _build(self, B: int, H: int, W: int, steps: int, guidance: Optional[float] = None):
        """guidance (> 1): classifier-free guidance as diffusers' pipeline runs it -- the ControlNet and the UNet see the batch twice
        (rows [0, B): the negative prompts, rows [B, 2B): the prompts; same latents, same control images) and the step uses
        eps_uncond + guidance * (eps_text - eps_uncond).  The reference's configs run guidance_scale = 0.0 (no doubling)."""
        dev = self.device
        E = Engine(dev, record=True)
        s = self.vae_scale_factor
        h, w = H // s, W // s
        L = self.tokenizer.model_max_length if hasattr(self.tokenizer, "model_max_length") else 77
        io = SimpleNamespace()
        Bn = 2 * B if guidance else B  # rows the networks see
        io.ids = E.buf("in_ids", (Bn, L), dtype=torch.int32, zero=True)
        io.image_u8 = E.buf("in_image", (Bn, H, W, 3), dtype=torch.uint8, zero=True)
        Cl = self.unet.config["in_channels"]
        io.noise = E.buf("in_noise", (B, h, w, Cl), zero=True)       # unit-variance draws (randn_tensor), NHWC
        io.latents = E.buf("latents", (B, h, w, Cl), zero=True)
        sch = self.scheduler
        sch.set_timesteps(steps)
        io.timesteps = [int(t) for t in sch.timesteps.tolist()]
        E.scale_pad(io.noise, sch.init_noise_sigma, Cl, out=io.latents)  # latents = randn * init_noise_sigma

        if self.two_streams:  # the conditioning-image embedding (512x512 convs) is independent of the text tower + K/V hoists
            E.fork()
        cond8 = E.image_u8_to_f16(io.image_u8, 8, 1.0, 0.0, name="cond8")  # VaeImageProcessor(do_normalize=False)
        cemb = graphs.emit_controlnet_cond(E, self.controlnet.W, self.controlnet.config, cond8)
        if self.two_streams:
            E.main()
        ctx, added = self._emit_prompt(E, io, Bn, L, H, W)
        kv_side = self.two_streams and os.environ.get("GN_KV_SIDE", "1") != "0"
        if kv_side:
            # the K / V hoists of the two networks (14 + 32 small M = B x 77 Linears) only share the prompt states: the ControlNet's run on
            # the side stream (behind the conditioning embedding) beside the UNet's
            E.join()
            E.fork()
        kv_cn = graphs.emit_cross_kv(E, self.controlnet.W, ctx, "cn")
        if kv_side:
            E.main()
        kv_un = graphs.emit_cross_kv(E, self.unet.W, ctx, "unet")
        if self.two_streams:
            E.join()
        ancestral = getattr(sch, "ancestral", False)
        linear = getattr(sch, "sampler", "euler") == "linear"  # DDPM / DDIM (log_validation's train-scheduler swap): x <- A x + B eps + C z
        coeffs = [sch.step_coeffs(i) for i in range(steps)] if linear else None
        if ancestral or (linear and any(c[2] > 0.0 for c in coeffs)):  # fresh unit noise per step, scaled on the device
            io.step_noise = E.buf("in_step_noise", (steps, B, h, w, Cl), zero=True)
        ones = torch.ones(B, dtype=torch.float32, device=dev)
        zeros = torch.zeros(B, dtype=torch.float32, device=dev)
        E._keepalive(ones, zeros)

        def scalar(v):
            t = torch.full((B,), float(v), dtype=torch.float32, device=dev)
            E._keepalive(t)
            return t
        # the timestep MLP + every ResNet's time_emb_proj depend on the step's timestep only: all steps' time shifts as ONE pass per network
        # (rows [i * Bn, (i + 1) * Bn) belong to step i) instead of four small launches per network and step
        sh_cn = sh_un = None
        if added is None and getattr(E, "hoist_time_shifts", True):
            t_all = torch.cat([torch.full((Bn,), float(sch.timesteps[i]), dtype=torch.float32, device=dev) for i in range(steps)])
            E._keepalive(t_all)
            with E.scope("cn_t"):
                sh_cn = graphs.emit_time_shifts(E, self.controlnet.W, self.controlnet.config, t_all)
            with E.scope("un_t"):
                sh_un = graphs.emit_time_shifts(E, self.unet.W, self.unet.config, t_all)
        io.first_step_op = E.num_ops
        for i in range(steps):
            sigma, sigma_next = (0.0, 0.0) if linear else (float(sch.sigmas[i]), float(sch.sigmas[i + 1]))
            t_dev = torch.full((Bn,), float(sch.timesteps[i]), dtype=torch.float32, device=dev)
            E._keepalive(t_dev)
            s_cn = None if sh_cn is None else sh_cn[i * Bn:(i + 1) * Bn]
            s_un = None if sh_un is None else sh_un[i * Bn:(i + 1) * Bn]
            if guidance:  # latent_model_input = torch.cat([latents] * 2)
                x8 = E.buf("x8", (Bn, h, w, 8))
                E.scale_pad(io.latents, sch.input_scale(i), 8, out=x8[:B])
                E.scale_pad(io.latents, sch.input_scale(i), 8, out=x8[B:])
            else:
                x8 = E.scale_pad(io.latents, sch.input_scale(i), 8, name="x8")
            # the ControlNet and the UNet encoder + mid block both read only x8: the ControlNet runs on the program's side stream and is
            # joined where the UNet consumes its residuals (fills the CUs the small-M deep-level kernels leave idle)
            if self.two_streams:
                E.fork()
            down, mid = graphs.emit_controlnet(E, self.controlnet.W, self.controlnet.config, x8, t_dev, kv_cn, cemb, 1.0, added=added, shifts=s_cn,
                                               defer_zero_convs=self.zero_convs_fused)
            if self.two_streams:
                E.main()
            eps = graphs.emit_unet(E, self.unet.W, self.unet.config, x8, t_dev, kv_un, down, mid, added=added,
                                   before_residuals=E.join if self.two_streams else None, shifts=s_un)
            if guidance:  # noise_pred = uncond + guidance_scale * (text - uncond)
                eps = E.add_noise(eps[:B], eps[B:], scalar(1.0 - guidance), scalar(guidance), name="eps_cfg")
            if ancestral:
                sigma_down, sigma_up = sch.ancestral_sigmas(i)
                E.euler_step(io.latents, eps, sigma, sigma_down)
                if sigma_up > 0.0:
                    E.add_noise(io.latents, io.step_noise[i], ones, scalar(sigma_up), out=io.latents)
            elif linear:
                A, Bc, Cn = coeffs[i]
                E.add_noise(io.latents, io.latents, scalar(A), zeros, out=io.latents)   # x <- A x
                E.euler_step(io.latents, eps, 1.0, 1.0 + Bc)                             # x <- x + B eps
                if Cn > 0.0:
                    E.add_noise(io.latents, io.step_noise[i], ones, scalar(Cn), out=io.latents)
            else:
                E.euler_step(io.latents, eps, sigma, sigma_next)
            if i == 0:
                io.ops_per_step = E.num_ops - io.first_step_op
        io.first_vae_op = E.num_ops
        z8 = E.scale_pad(io.latents, 1.0 / self.vae.config["scaling_factor"], 8, name="z8")
        if isinstance(self.vae, AutoencoderTiny):  # autoencoder: taesd (controller/agent/sd_controlnet_agent.py:45-49)
            img = graphs.emit_taesd_decode(E, self.vae.W, self.vae.config, z8)
        else:
            img = graphs.emit_vae_decode(E, self.vae.W, self.vae.config, z8)
        io.out_u8 = E.image_f16_to_u8(img, name="out_u8")
        io.engine = E
        from .engine import save_tune_table

        save_tune_table()
        if self.use_graph:
            side = torch.cuda.Stream(device=dev)
            E.use_stream(side)
            with torch.cuda.stream(side):
                E.run()  # warm-up outside capture (lazy module loads)
                side.synchronize()
                E.capture()
            io.stream = side
        return io

    def _modules(self):
        return (self.vae, self.text_encoder, self.unet, self.controlnet)

    def program(self, B, H, W, steps, guidance: Optional[float] = None):
        # a recorded program bakes in the scheduler's tables and the packed weights' addresses: swapping pipe.scheduler /
        # pipe.controlnet / pipe.vae or re-packing a module (load_state_dict) must not replay the stale program
        sch = self.scheduler
        key = (B, H, W, steps, id(sch), type(sch).__name__, getattr(sch.config, "timestep_spacing", None),
               tuple((id(m), m._pack_gen) for m in self._modules()), guidance)
        if key not in self._progs:
            self._progs = {k: v for k, v in self._progs.items() if k[4:8] == key[4:8]}  # drop programs of replaced modules
            if self.unet.W is None:
                raise GenimaHipError("pipeline is not on a ROCm device: call pipe.to('cuda') first (no CPU fallback)")
            self._progs[key] = self._build(B, H, W, steps, guidance)
        return self._progs[key]

    # ---- the call ------------------------------------------------------------------------------------------------------
    def encode_ids(self, prompt) -> torch.Tensor:
        tok = self.tokenizer(prompt, padding="max_length", max_length=getattr(self.tokenizer, "model_max_length", 77),
                             truncation=True, return_tensors="pt")
        return tok.input_ids

    @staticmethod
    def _images_to_u8(image, B) -> torch.Tensor:
        if isinstance(image, torch.Tensor):
            t = image
            if t.dtype != torch.uint8:  # float NCHW in [0,1] like VaeImageProcessor accepts
                t = (t.permute(0, 2, 3, 1) * 255.0).round().clamp(0, 255).to(torch.uint8)
            return t
        imgs = image if isinstance(image, (list, tuple)) else [image]
        arr = np.stack([np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8) for im in imgs])
        if arr.shape[0] == 1 and B > 1:
            arr = np.repeat(arr, B, axis=0)
        return torch.from_numpy(arr)

    def __call__(self, prompt=None, image=None, negative_prompt=None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 generator=None, latents: Optional[torch.Tensor] = None, output_type: str = "pil", prompt_ids=None,
                 height=None, width=None, return_dict: bool = True, **kw):
        guidance = float(guidance_scale) if guidance_scale > 1.0 else None  # diffusers: do_classifier_free_guidance = guidance_scale > 1
        if guidance and type(self)._emit_prompt is not StableDiffusionControlNetPipeline._emit_prompt:
            raise NotImplementedError("classifier-free guidance (guidance_scale > 1) is built for the SD-2.x pipeline only; the SDXL-Turbo "
                                      "agent of the reference runs guidance_scale=0.0 (SURVEY.md Appendix D.1)")
        if prompt_ids is None:
            prompt_ids = self.encode_ids(prompt)
        B = prompt_ids.shape[0]
        img_u8 = self._images_to_u8(image, B)
        assert img_u8.shape[0] == B, "one control image per prompt"
        H, W = int(img_u8.shape[1]), int(img_u8.shape[2])
        if guidance:  # rows [0, B): negative prompts ("" by default, as diffusers), rows [B, 2B): the prompts
            neg = "" if negative_prompt is None else negative_prompt
            neg_ids = self.encode_ids([neg] * B if isinstance(neg, str) else list(neg))
            assert neg_ids.shape == prompt_ids.shape, "one negative prompt per prompt"
            prompt_ids = torch.cat([neg_ids, prompt_ids], dim=0)
            img_u8 = torch.cat([img_u8, img_u8], dim=0)
        io = self.program(B, H, W, num_inference_steps, guidance)
        E: Engine = io.engine
        s = self.vae_scale_factor
        C = self.unet.config["in_channels"]
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        if latents is None:
            latents = randn_latents((B, C, H // s, W // s), generator, self.device)
        else:
            latents = latents.to(self.device, torch.float16)  # unit-variance, scaled by init_noise_sigma on the device
        step_noise = getattr(io, "step_noise", None)
        if step_noise is not None:  # ancestral sampler: the per-step draws are used (same order as diffusers: latents, then one per step)
            for i in range(num_inference_steps):
                step_noise[i].copy_(randn_latents((B, C, H // s, W // s), generator, self.device).permute(0, 2, 3, 1))
        elif sch.draws_step_noise and generator is not None:
            for _ in range(num_inference_steps):  # mirror diffusers 0.29.0's per-step (unused) randn draw
                randn_latents((B, C, H // s, W // s), generator, self.device)
        stream = getattr(io, "stream", None)
        cur = torch.cuda.current_stream(self.device)
        self._fill_prompt_inputs(io, kw)
        io.ids.copy_(prompt_ids.to(torch.int32), non_blocking=False)
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
            arr = out.cpu().numpy()  # D->H 768 KB / sample + the call's only sync
            if output_type == "np":
                images = arr
            else:
                from PIL import Image

                images = [Image.fromarray(a) for a in arr]
        return PipelineOutput(images) if return_dict else (images, None)


class StableDiffusionXLControlNetPipeline(StableDiffusionControlNetPipeline):
    """SDXL(-Turbo) + ControlNet (controller/agent/sdxl_controlnet_agent.py:36-42, 67-76): two CLIP towers whose penultimate hidden
    states form the 2048-wide context, ``text_embeds`` + ``time_ids`` added conditions on both networks, EulerAncestral sampling.
    Same recorded-program execution as the SD-2.x pipeline; call surface ``pipe(prompt=, image=, negative_prompt=,
    num_inference_steps=, guidance_scale=, generator=)`` plus ``prompt_2`` / ``prompt_ids_2``."""

    def __init__(self, vae, text_encoder, text_encoder_2, tokenizer, tokenizer_2, unet, controlnet, scheduler, **kw):
        super().__init__(vae, text_encoder, tokenizer, unet, controlnet, scheduler, **kw)
        self.text_encoder_2 = text_encoder_2
        self.tokenizer_2 = tokenizer_2 or HashTokenizer(text_encoder_2.config["vocab_size"])

    @classmethod
    def from_pretrained(cls, path, controlnet=None, safety_checker=None, torch_dtype=None, variant=None,
                        allow_hash_tokenizer: bool = False, allow_fp16_vae: bool = False, **kw):
        """diffusers SDXL pipeline directory: ``unet/ vae/ text_encoder/ text_encoder_2/ tokenizer/ tokenizer_2/ scheduler/``."""
        import json

        from .host import CLIPTextModelWithProjection
        from .scheduler import EulerAncestralDiscreteScheduler

        unet = UNet2DConditionModel.from_pretrained(path, "unet", variant=variant)
        vae = AutoencoderKL.from_pretrained(path, "vae", variant=variant)
        # diffusers' SDXL pipeline decodes in fp32 when vae.config.force_upcast is set (the stock SDXL VAE overflows f16 and yields
        # NaN / black images); the HIP VAE keeps f16 activations, so such a VAE is refused here instead of producing garbage
        # the HIP VAE keeps f16 activations: such a VAE runs with its residual stream carried at 1/64 (host.AutoencoderKL.enable_stream_scaling,
        # an exact re-parametrisation of the weights), not silently in plain f16; allow_fp16_vae=True skips that for a VAE known to be in range
        if vae.config.get("force_upcast", True) and not allow_fp16_vae:
            vae.enable_stream_scaling()
        text = CLIPTextModel.from_pretrained(path, "text_encoder", variant=variant)
        text2 = CLIPTextModelWithProjection.from_pretrained(path, "text_encoder_2", variant=variant)
        with open(os.path.join(path, "scheduler", "scheduler_config.json")) as f:
            cfg = json.load(f)
        sched_cls = EulerAncestralDiscreteScheduler if "Ancestral" in cfg.get("_class_name", "EulerAncestral") else EulerDiscreteScheduler
        tok = _load_tokenizer(path, "tokenizer", allow_hash_tokenizer, text.config["vocab_size"])
        tok2 = _load_tokenizer(path, "tokenizer_2", allow_hash_tokenizer, text2.config["vocab_size"])
        if controlnet is None:
            controlnet = ControlNetModel.from_unet(unet)
        return cls(vae, text, text2, tok, tok2, unet, controlnet, sched_cls.from_config(cfg))

    @classmethod
    def from_synthetic(cls, family: dict, seed: int = 0, gen_device="cpu"):
        from .host import CLIPTextModelWithProjection
        from .scheduler import EulerAncestralDiscreteScheduler

        unet = UNet2DConditionModel.from_config(family["unet"], seed + 1, gen_device)
        cn = ControlNetModel.from_config(family["controlnet"], seed + 2, gen_device)
        vae = AutoencoderKL.from_config(family["vae"], seed + 3, gen_device)
        text = CLIPTextModel.from_config(family["text"], seed + 4, gen_device)
        text2 = CLIPTextModelWithProjection.from_config(family["text_2"], seed + 5, gen_device)
        return cls(vae, text, text2, None, None, unet, cn, EulerAncestralDiscreteScheduler.from_config(family["scheduler"]))

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            self.text_encoder_2.to(device)
        return super().to(device, *a, **k)

    def _modules(self):
        return super()._modules() + (self.text_encoder_2,)

    def _emit_prompt(self, E: Engine, io, B: int, L: int, H: int, W: int):
        io.ids2 = E.buf("in_ids2", (B, L), dtype=torch.int32, zero=True)
        with E.scope("te1"):
            pen_l, _ = graphs.emit_clip_text_sdxl(E, self.text_encoder.W, self.text_encoder.config, io.ids)
        with E.scope("te2"):
            pen_g, pooled = graphs.emit_clip_text_sdxl(E, self.text_encoder_2.W, self.text_encoder_2.config, io.ids2)
        dl, dg = pen_l.shape[2], pen_g.shape[2]
        ctx = E.buf("ctx_cat", (B, L, dl + dg))
        E.copy4d(pen_l, ctx, (1, 1, B, L), (0, 0, L * dl, dl), (0, 0, L * (dl + dg), dl + dg), dl)
        E.copy4d(pen_g, ctx[:, :, dl:], (1, 1, B, L), (0, 0, L * dg, dg), (0, 0, L * (dl + dg), dl + dg), dg)
        # _get_add_time_ids(original_size=(H, W), crops_coords_top_left=(0, 0), target_size=(H, W))
        time_ids = torch.tensor([[float(H), float(W), 0.0, 0.0, float(H), float(W)]] * B, dtype=torch.float32, device=self.device)
        E._keepalive(time_ids)
        return ctx, (pooled, time_ids)

    def _fill_prompt_inputs(self, io, kw):
        ids2 = kw.get("prompt_ids_2")
        if ids2 is None:
            p2 = kw.get("prompt_2")
            ids2 = self._ids_1 if p2 is None else self.tokenizer_2(p2, padding="max_length", max_length=77, truncation=True,
                                                                   return_tensors="pt").input_ids
        io.ids2.copy_(ids2.to(torch.int32))

    def __call__(self, prompt=None, image=None, prompt_ids=None, **kw):
        if prompt_ids is None:
            prompt_ids = self.encode_ids(prompt)
        self._ids_1 = prompt_ids  # the reference passes one prompt: both tokenizers see the same text
        if kw.get("prompt_ids_2") is None and kw.get("prompt_2") is None and prompt is not None:
            kw["prompt_2"] = prompt
        return super().__call__(prompt=prompt, image=image, prompt_ids=prompt_ids, **kw)
