"""Checkpoint / plugin loading rules that need no GPU: weight-file variant selection and sharded safetensors (diffusers semantics;
reference call sites controller/agent/sd_controlnet_agent.py:32-42 ``variant="fp16"`` and diffusion/train_controlnet_genima.py:1042-1064
``variant=None``), the agent's refusal to run without a ControlNet (sd_controlnet_agent.py:21-35), the tokenizer requirement for real
weights, SDXL ``force_upcast``, the AutoencoderTiny schema, scheduler ``prediction_type``."""
import json
import os
import types

import pytest
import torch
from safetensors.torch import save_file

from genima_amd import configs, schema, weights


def _cfg_dir(d, cfg):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)


def test_variant_selects_the_weight_file(tmp_path):
    d = str(tmp_path / "unet")
    _cfg_dir(d, {"x": 1})
    full = {"a.weight": torch.full((4,), 1.0009765625 + 2 ** -20)}  # not representable in f16
    save_file(full, os.path.join(d, "diffusion_pytorch_model.safetensors"))
    save_file({k: v.half() for k, v in full.items()}, os.path.join(d, "diffusion_pytorch_model.fp16.safetensors"))
    _, sd = weights.load_diffusers_dir(str(tmp_path), "unet")                      # trainer: variant=None -> fp32 master weights
    assert torch.equal(sd["a.weight"], full["a.weight"])
    _, sd16 = weights.load_diffusers_dir(str(tmp_path), "unet", variant="fp16")    # agents: variant="fp16"
    assert torch.equal(sd16["a.weight"], full["a.weight"].half().float()) and not torch.equal(sd16["a.weight"], sd["a.weight"])
    os.remove(os.path.join(d, "diffusion_pytorch_model.safetensors"))
    _, only16 = weights.load_diffusers_dir(str(tmp_path), "unet")                  # only the fp16 file exists: use it
    assert torch.equal(only16["a.weight"], sd16["a.weight"])


def test_sharded_safetensors(tmp_path):
    d = str(tmp_path / "unet")
    _cfg_dir(d, {})
    a, b = {"a": torch.arange(4.0)}, {"b": torch.arange(6.0).view(2, 3)}
    save_file(a, os.path.join(d, "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file(b, os.path.join(d, "diffusion_pytorch_model-00002-of-00002.safetensors"))
    with open(os.path.join(d, "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": {"a": "diffusion_pytorch_model-00001-of-00002.safetensors",
                                                  "b": "diffusion_pytorch_model-00002-of-00002.safetensors"}}, f)
    _, sd = weights.load_diffusers_dir(d)
    assert torch.equal(sd["a"], a["a"]) and torch.equal(sd["b"], b["b"])
    with pytest.raises(FileNotFoundError):
        _cfg_dir(str(tmp_path / "empty"), {})
        weights.load_diffusers_dir(str(tmp_path / "empty"))


def _eval_cfg(**kw):
    base = dict(diffusion_ckpt="", sd_ckpt="synthetic:tiny", device="cuda", image_resolution=512, show_diffusion_progress=False,
                autoencoder="")
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_agent_refuses_to_run_without_a_controlnet(tmp_path):
    """The reference lists ``diffusion_ckpt`` and ``from_pretrained`` s what it finds -- a bad path raises.  A zero-initialised
    ``from_unet`` stand-in would run the whole evaluation unconditioned, with plausible-looking images."""
    from genima_amd.agent import SDControlNetAgent, SDXLControlNetAgent

    real_sd = str(tmp_path / "sd")  # any directory: sd_ckpt is not read before the ControlNet resolves
    os.makedirs(real_sd)
    for cls in (SDControlNetAgent, SDXLControlNetAgent):
        with pytest.raises(FileNotFoundError, match="diffusion_ckpt"):
            cls(_eval_cfg(sd_ckpt=real_sd, diffusion_ckpt=str(tmp_path / "missing")))
        with pytest.raises(FileNotFoundError, match="no ControlNet checkpoint"):
            os.makedirs(str(tmp_path / "run" / "checkpoint-5"), exist_ok=True)
            cls(_eval_cfg(sd_ckpt=real_sd, diffusion_ckpt=str(tmp_path / "run")))
        with pytest.raises(FileNotFoundError, match="diffusion_ckpt"):
            cls(_eval_cfg(sd_ckpt=real_sd, diffusion_ckpt=""))


def _write_component(root, sub, cls_cfg, sch_fn, seed):
    from genima_amd.weights import save_diffusers_dir, synth_state_dict

    name = "model.safetensors" if "text" in sub else "diffusion_pytorch_model.safetensors"
    save_diffusers_dir(os.path.join(root, sub), cls_cfg, synth_state_dict(sch_fn(cls_cfg), seed), torch.float16, name)


def test_real_weights_need_the_real_tokenizer_and_sdxl_force_upcast(tmp_path):
    from genima_amd.pipeline import HashTokenizer, StableDiffusionControlNetPipeline, StableDiffusionXLControlNetPipeline
    from genima_amd.tokenizer import BOS, EOS, CLIPTokenizer, bytes_to_unicode

    root = str(tmp_path / "sd")
    fam = configs.family("tiny")
    _write_component(root, "unet", fam["unet"], schema.unet_schema, 1)
    _write_component(root, "vae", fam["vae"], schema.vae_schema, 2)
    _write_component(root, "text_encoder", fam["text"], schema.clip_text_schema, 3)
    _cfg_dir(os.path.join(root, "scheduler"), {})
    os.rename(os.path.join(root, "scheduler", "config.json"), os.path.join(root, "scheduler", "scheduler_config.json"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(configs.SD_TURBO_SCHEDULER, f)
    with pytest.raises(FileNotFoundError, match="real tokenizer"):
        StableDiffusionControlNetPipeline.from_pretrained(root)
    pipe = StableDiffusionControlNetPipeline.from_pretrained(root, allow_hash_tokenizer=True)
    assert isinstance(pipe.tokenizer, HashTokenizer)
    # with a BPE model in tokenizer/ the pipeline tokenises with it
    alphabet = sorted(bytes_to_unicode().values())
    vocab = {c: i for i, c in enumerate(alphabet)}
    for c in alphabet:
        vocab[c + "</w>"] = len(vocab)
    vocab["open</w>"] = len(vocab)
    vocab[BOS], vocab[EOS] = len(vocab), len(vocab) + 1
    os.makedirs(os.path.join(root, "tokenizer"))
    with open(os.path.join(root, "tokenizer", "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(root, "tokenizer", "merges.txt"), "w") as f:
        f.write("#version: 0.2\no p\nop e\nope n</w>\n")
    with open(os.path.join(root, "tokenizer", "special_tokens_map.json"), "w") as f:
        json.dump({"pad_token": "!"}, f)
    pipe = StableDiffusionControlNetPipeline.from_pretrained(root)
    assert isinstance(pipe.tokenizer, CLIPTokenizer)
    ids = pipe.encode_ids(["open"])[0].tolist()
    assert ids[:3] == [vocab[BOS], vocab["open</w>"], vocab[EOS]] and ids[3:] == [vocab["!"]] * 74
    # SDXL: a VAE that asks for the fp32 decode gets the stream-scaled f16 decode unless the caller vouches for plain f16
    xl = str(tmp_path / "xl")
    famx = configs.family("tiny-xl")
    _write_component(xl, "unet", famx["unet"], schema.unet_schema, 1)
    _write_component(xl, "vae", dict(famx["vae"], force_upcast=True), schema.vae_schema, 2)
    _write_component(xl, "text_encoder", famx["text"], schema.clip_text_schema, 3)
    _write_component(xl, "text_encoder_2", famx["text_2"], schema.clip_text_schema, 4)
    os.makedirs(os.path.join(xl, "scheduler"))
    with open(os.path.join(xl, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(configs.SD_TURBO_SCHEDULER, _class_name="EulerAncestralDiscreteScheduler"), f)
    p1 = StableDiffusionXLControlNetPipeline.from_pretrained(xl, allow_hash_tokenizer=True)
    assert p1.vae.stream_scale == 1.0 / 64.0  # force_upcast: the residual stream is carried scaled, not run in plain f16
    p2 = StableDiffusionXLControlNetPipeline.from_pretrained(xl, allow_hash_tokenizer=True, allow_fp16_vae=True)
    assert isinstance(p2.tokenizer_2, HashTokenizer) and p2.vae.stream_scale == 1.0


def test_taesd_schema_and_prediction_type():
    from genima_amd.scheduler import EulerDiscreteScheduler

    dec = schema.taesd_schema(configs.TAESD, encoder=False)
    enc = schema.taesd_schema(configs.TAESD, decoder=False)
    assert schema.param_count(dec) == 1_222_531 and schema.param_count(enc) == 1_222_532  # madebyollin/taesd: 1.22 M each way
    assert "decoder.layers.18.bias" in dec and "decoder.layers.6.bias" not in dec and "decoder.layers.5.weight" not in dec
    assert "decoder.layers.2.conv.4.weight" in dec and "encoder.layers.14.weight" in enc and "encoder.layers.2.bias" not in enc
    with pytest.raises(NotImplementedError, match="prediction_type"):
        EulerDiscreteScheduler(prediction_type="v_prediction")
