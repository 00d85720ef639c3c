"""Architecture configs for the Genima hot path (diffusers-style ``config.json`` dictionaries).

The reference loads these architectures from checkpoints:
  * ``stabilityai/sd-turbo`` UNet/VAE/text-encoder  (controller/cfgs/eval_genima.yaml:4,
    diffusion/train_controlnet_genima.py:1042-1064)
  * ``ControlNetModel.from_unet(unet)``             (diffusion/train_controlnet_genima.py:1066-1071)
Key names follow the published diffusers/transformers ``config.json`` schema so a real checkpoint's
config can be dropped in unchanged.  ``*_tiny`` configs keep the topology (4 levels, 2 layers per
block, cross-attention at levels 0-2, head_dim 64, GroupNorm(32)) at reduced width for parity tests.
"""
from __future__ import annotations

import copy

# ----------------------------------------------------------------------------- UNet (SD-2.1 family)
SD_TURBO_UNET = {
    "_class_name": "UNet2DConditionModel",
    "in_channels": 4,
    "out_channels": 4,
    "sample_size": 64,
    "block_out_channels": [320, 640, 1280, 1280],
    "layers_per_block": 2,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "attention_head_dim": [5, 10, 20, 20],  # == number of heads in the SD-2.x family (head dim 64)
    "cross_attention_dim": 1024,
    "use_linear_projection": True,
    "norm_num_groups": 32,
    "norm_eps": 1e-5,
    "act_fn": "silu",
    "flip_sin_to_cos": True,
    "freq_shift": 0,
    "transformer_layers_per_block": 1,
    "resnet_time_scale_shift": "default",
}

TINY_UNET = dict(
    SD_TURBO_UNET,
    block_out_channels=[64, 128, 256, 256],
    attention_head_dim=[1, 2, 4, 4],
    cross_attention_dim=128,
    sample_size=16,
)

# ----------------------------------------------------------------------------- ControlNet
SD_TURBO_CONTROLNET = dict(
    {k: v for k, v in SD_TURBO_UNET.items() if k not in ("out_channels", "up_block_types")},
    _class_name="ControlNetModel",
    conditioning_channels=3,
    conditioning_embedding_out_channels=[16, 32, 96, 256],
    global_pool_conditions=False,
)
TINY_CONTROLNET = dict(
    {k: v for k, v in TINY_UNET.items() if k not in ("out_channels", "up_block_types")},
    _class_name="ControlNetModel",
    conditioning_channels=3,
    conditioning_embedding_out_channels=[16, 32, 96, 256],
    global_pool_conditions=False,
)

# ----------------------------------------------------------------------------- SDXL-Turbo family (BASELINE.json configs[4]; SURVEY Appendix A.5;
# diffusion/train_controlnet_sdxl_genima.py:107, controller/agent/sdxl_controlnet_agent.py:36-42)
SDXL_TURBO_UNET = dict(
    SD_TURBO_UNET,
    sample_size=64,  # 512x512 tiled observations -> 64x64 latents
    block_out_channels=[320, 640, 1280],
    down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
    attention_head_dim=[5, 10, 20],
    transformer_layers_per_block=[1, 2, 10],
    cross_attention_dim=2048,
    addition_embed_type="text_time",
    addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816,  # 6 time ids x 256 + pooled text 1280
)
SDXL_TURBO_CONTROLNET = dict(
    {k: v for k, v in SDXL_TURBO_UNET.items() if k not in ("out_channels", "up_block_types")},
    _class_name="ControlNetModel",
    conditioning_channels=3,
    conditioning_embedding_out_channels=[16, 32, 96, 256],
    global_pool_conditions=False,
)
TINY_XL_UNET = dict(
    SDXL_TURBO_UNET,
    block_out_channels=[64, 128, 256],
    attention_head_dim=[1, 2, 4],
    transformer_layers_per_block=[1, 2, 3],
    cross_attention_dim=192,
    addition_time_embed_dim=32,
    projection_class_embeddings_input_dim=6 * 32 + 128,
    sample_size=32,
)
TINY_XL_CONTROLNET = dict(
    {k: v for k, v in TINY_XL_UNET.items() if k not in ("out_channels", "up_block_types")},
    _class_name="ControlNetModel",
    conditioning_channels=3,
    conditioning_embedding_out_channels=[16, 32, 96, 256],
    global_pool_conditions=False,
)
# ----------------------------------------------------------------------------- AutoencoderKL
SD_TURBO_VAE = {
    "_class_name": "AutoencoderKL",
    "in_channels": 3,
    "out_channels": 3,
    "latent_channels": 4,
    "block_out_channels": [128, 256, 512, 512],
    "layers_per_block": 2,
    "norm_num_groups": 32,
    "act_fn": "silu",
    "scaling_factor": 0.18215,
    "sample_size": 768,
}
TINY_VAE = dict(SD_TURBO_VAE, block_out_channels=[32, 64, 64, 64], sample_size=128)
SDXL_VAE = dict(SD_TURBO_VAE, scaling_factor=0.13025, sample_size=1024)  # madebyollin/sdxl-vae-fp16-fix: same architecture

TAESD = {  # madebyollin/taesd (SD-1.x / 2.x latents) and taesdxl: same architecture (diffusers AutoencoderTiny defaults)
    "_class_name": "AutoencoderTiny",
    "in_channels": 3,
    "out_channels": 3,
    "latent_channels": 4,
    "encoder_block_out_channels": [64, 64, 64, 64],
    "decoder_block_out_channels": [64, 64, 64, 64],
    "block_out_channels": [64, 64, 64, 64],
    "num_encoder_blocks": [1, 3, 3, 3],
    "num_decoder_blocks": [3, 3, 3, 1],
    "act_fn": "relu",
    "upsample_fn": "nearest",
    "upsampling_scaling_factor": 2,
    "latent_magnitude": 3,
    "latent_shift": 0.5,
    "force_upcast": False,
    "scaling_factor": 1.0,
    "shift_factor": 0.0,
}

# ----------------------------------------------------------------------------- CLIP text towers
SD_TURBO_TEXT = {  # OpenCLIP ViT-H/14 text tower truncated to 23 layers
    "_class_name": "CLIPTextModel",
    "vocab_size": 49408,
    "hidden_size": 1024,
    "intermediate_size": 4096,
    "num_hidden_layers": 23,
    "num_attention_heads": 16,
    "max_position_embeddings": 77,
    "hidden_act": "gelu",
    "layer_norm_eps": 1e-5,
    "projection_dim": 0,
}
TINY_TEXT = dict(SD_TURBO_TEXT, vocab_size=1024, hidden_size=128, intermediate_size=512,
                 num_hidden_layers=2, num_attention_heads=2)

# SDXL's two towers (diffusion/train_controlnet_sdxl_genima.py:1027-1071): the penultimate hidden state of each is concatenated to
# the 2048-wide cross-attention context, the pooled + projected bigG output is the `text_embeds` added condition (:879-893)
SDXL_TEXT_L = dict(SD_TURBO_TEXT, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   hidden_act="quick_gelu", projection_dim=0)
SDXL_TEXT_G = dict(SD_TURBO_TEXT, _class_name="CLIPTextModelWithProjection", hidden_size=1280, intermediate_size=5120,
                   num_hidden_layers=32, num_attention_heads=20, hidden_act="gelu", projection_dim=1280)
TINY_XL_TEXT_L = dict(SDXL_TEXT_L, vocab_size=1024, hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1)
TINY_XL_TEXT_G = dict(SDXL_TEXT_G, vocab_size=1024, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                      projection_dim=128)

ACT_CLIP_TEXT = {  # openai CLIP ViT-B/32 text tower (controller/method/genima_act.py:314-346)
    "_class_name": "CLIPTextModel",
    "vocab_size": 49408,
    "hidden_size": 512,
    "intermediate_size": 2048,
    "num_hidden_layers": 12,
    "num_attention_heads": 8,
    "max_position_embeddings": 77,
    "hidden_act": "quick_gelu",
    "layer_norm_eps": 1e-5,
    "projection_dim": 512,
}
TINY_ACT_CLIP_TEXT = dict(ACT_CLIP_TEXT, vocab_size=1024, hidden_size=128, intermediate_size=512,
                          num_hidden_layers=2, num_attention_heads=2, projection_dim=64)

# ----------------------------------------------------------------------------- scheduler (SD-Turbo scheduler_config.json)
SD_TURBO_SCHEDULER = {
    "_class_name": "EulerDiscreteScheduler",
    "num_train_timesteps": 1000,
    "beta_start": 0.00085,
    "beta_end": 0.012,
    "beta_schedule": "scaled_linear",
    "prediction_type": "epsilon",
    "timestep_spacing": "trailing",
    "steps_offset": 1,
    "use_karras_sigmas": False,
    "interpolation_type": "linear",
    "timestep_type": "discrete",
}

# ----------------------------------------------------------------------------- ACT controller
ACT_POLICY = {  # controller/cfgs/method/genima_act.yaml:13-39
    "hidden_dim": 256,
    "enc_layers": 4,
    "dec_layers": 6,
    "dim_feedforward": 2048,
    "nheads": 8,
    "num_queries": 20,
    "state_dim": 8,
    "action_dim": 8,
    "latent_dim": 32,
    "num_views": 4,
    "image_size": 256,
    "backbone": "resnet18",
    "use_lang_cond": True,
    "lang_dim": 512,
    "pre_norm": False,
}
TINY_ACT_POLICY = dict(ACT_POLICY, hidden_dim=64, enc_layers=1, dec_layers=2, dim_feedforward=128, nheads=2,
                       image_size=64, lang_dim=64)


def family(name: str) -> dict:
    """Return the dict of component configs for a model family ("sd-turbo" or "tiny")."""
    if name == "sd-turbo":
        fam = dict(unet=SD_TURBO_UNET, controlnet=SD_TURBO_CONTROLNET, vae=SD_TURBO_VAE, text=SD_TURBO_TEXT,
                   scheduler=SD_TURBO_SCHEDULER, act=ACT_POLICY, act_text=ACT_CLIP_TEXT)
    elif name == "tiny":
        fam = dict(unet=TINY_UNET, controlnet=TINY_CONTROLNET, vae=TINY_VAE, text=TINY_TEXT,
                   scheduler=SD_TURBO_SCHEDULER, act=TINY_ACT_POLICY, act_text=TINY_ACT_CLIP_TEXT)
    elif name == "sdxl-turbo":  # two text towers: text = CLIP-L, text_2 = OpenCLIP bigG with projection
        fam = dict(unet=SDXL_TURBO_UNET, controlnet=SDXL_TURBO_CONTROLNET, vae=SDXL_VAE, text=SDXL_TEXT_L, text_2=SDXL_TEXT_G,
                   scheduler=SD_TURBO_SCHEDULER, act=ACT_POLICY, act_text=ACT_CLIP_TEXT)
    elif name == "tiny-xl":
        fam = dict(unet=TINY_XL_UNET, controlnet=TINY_XL_CONTROLNET, vae=TINY_VAE, text=TINY_XL_TEXT_L, text_2=TINY_XL_TEXT_G,
                   scheduler=SD_TURBO_SCHEDULER, act=TINY_ACT_POLICY, act_text=TINY_ACT_CLIP_TEXT)
    elif name in ("sd-turbo-pix2pix", "tiny-pix2pix"):
        # InstructPix2Pix base (diffusion/train_instruct_pix2pix_genima.py:800-818): the same networks, the UNet's conv_in widened to
        # latent_channels + image-latent channels = 8; no ControlNet
        fam = family(name[: -len("-pix2pix")])
        fam["unet"] = dict(fam["unet"], in_channels=8)
        del fam["controlnet"]
        return fam
    else:
        raise KeyError(f"unknown model family {name!r}")
    return copy.deepcopy(fam)
