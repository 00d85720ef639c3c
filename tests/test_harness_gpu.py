"""-m gpu: BASELINE.json configs[0] without the simulator (SURVEY.md section 8d "Config 1"): one synthetic observation dict through
the whole control step of controller/eval_genima.py:162-248 -- tiling, the diffusion-agent plugin (``infer``), untiling, the
observation overwrite and the ACT controller plugin (``act``) -- checking shapes, dtypes, ordering and the call contract."""
import types

import numpy as np
import pytest
import torch

from genima_amd import configs, harness, schema, weights
from genima_amd.act import GenimaACT, act_schema
from genima_amd.agent import SDControlNetAgent
from genima_amd.tiling import CROP_ORDER

pytestmark = pytest.mark.gpu

CAMERAS = ["front", "left_shoulder", "right_shoulder", "wrist"]  # a RoboBase camera list (any order: the tile follows it)


def _agents():
    cfg = types.SimpleNamespace(diffusion_ckpt="", sd_ckpt="synthetic:tiny", device="cuda", image_resolution=512, vae_slicing=False,
                                upcast_vae=False, fused_projections=True, enable_xformers_memory_efficient_attention=True,
                                show_diffusion_progress=False, torch_compile=False, autoencoder="")
    dagent = SDControlNetAgent(cfg)
    acfg, ccfg = configs.ACT_POLICY, dict(configs.TINY_ACT_CLIP_TEXT, projection_dim=512)
    sd = weights.round_to(weights.synth_state_dict(act_schema(acfg), 31), torch.float16)
    csd = weights.round_to(weights.synth_state_dict(schema.clip_text_schema(ccfg), 32), torch.float16)
    return dagent, GenimaACT(acfg, sd, ccfg, csd, device="cuda"), ccfg


def _obs(ccfg, fs=1):
    obs = {}
    for i, cam in enumerate(CAMERAS):
        obs[f"{cam}_rgb"] = weights.counter_bytes(40 + i, "harness", fs * 3 * 256 * 256).reshape(fs, 3, 256, 256)
    obs["low_dim_state"] = np.linspace(-1, 1, fs * 8, dtype=np.float32).reshape(fs, 8)
    toks = np.zeros((fs, 1, 77), dtype=np.int32)
    toks[:, 0, :5] = [ccfg["vocab_size"] - 2, 5, 6, 7, ccfg["vocab_size"] - 1]
    obs["lang_tokens"] = toks
    return obs


def test_agent_resolves_the_latest_checkpoint_in_natural_order(tmp_path):
    """controller/agent/sd_controlnet_agent.py:21-35: ``<diffusion_ckpt>/checkpoint-<max>/controlnet`` with natsort, so
    checkpoint-10 wins over checkpoint-9 (a lexicographic sort would pick 9)."""
    from genima_amd.host import ControlNetModel

    fam = configs.family("tiny")
    first = None
    for step, seed in ((9, 101), (10, 102), (2, 103)):
        sd = weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), seed)
        ControlNetModel(fam["controlnet"], sd).save_pretrained(str(tmp_path / f"checkpoint-{step}" / "controlnet"))
        if step == 10:
            first = next(iter(sd.items()))
    cfg = types.SimpleNamespace(diffusion_ckpt=str(tmp_path), sd_ckpt="synthetic:tiny", device="cuda", image_resolution=512,
                                show_diffusion_progress=False)
    agent = SDControlNetAgent(cfg)
    got = agent.pipe.controlnet.state_dict()[first[0]].float().cpu()
    assert torch.equal(got, first[1].float().cpu()), "the agent must load checkpoint-10"
    from PIL import Image
    assert agent.transform_to_half_resolution(Image.new("RGB", (256, 256))).size == (256, 256)
    assert agent.transform_to_resolution(Image.new("RGB", (640, 512))).size == (512, 512)


def test_control_step_contract():
    dagent, cagent, ccfg = _agents()
    obs, calls = _obs(ccfg), []
    real_infer = dagent.infer

    def spy(*a, **kw):
        assert not a, "the evaluation loop calls infer with keywords only"
        calls.append(kw)
        return real_infer(**kw)

    dagent.infer = spy
    gen = [torch.Generator(device="cuda").manual_seed(2)]  # diffusion_seed (controller/cfgs/eval_genima.yaml:32)
    actions, obs_after, tiled_in, tiled_out = harness.control_step(dagent, cagent, obs, "open the box", CAMERAS, 1, gen, 5, 0.0, "cuda")
    kw = calls[0]
    assert kw["prompts"] == ["tiled perspectives of a robot arm executing 'open the box'"] and len(kw["negative_prompts"]) == 1
    assert kw["num_inference_steps"] == 5 and kw["guidance_scale"] == 0.0
    assert len(kw["generator"]) == len(tiled_in) == 1 and kw["generator"][0] is gen[0], "the same Generator object, once per image"
    # tiling: camera k of the list sits at CROP_ORDER[k]
    tin = np.asarray(tiled_in[0])
    assert tin.shape == (512, 512, 3)
    for k, cam in enumerate(CAMERAS):
        l, t, r, b = CROP_ORDER[k]
        assert np.array_equal(tin[t:b, l:r], obs[f"{cam}_rgb"][0].transpose(1, 2, 0)), cam
    # pipeline output surface: list of PIL 512x512 RGB
    assert len(tiled_out) == 1 and tiled_out[0].size == (512, 512) and tiled_out[0].mode == "RGB"
    tout = np.asarray(tiled_out[0])
    # the controller saw the generated views, by camera name, with a leading batch axis, on the device
    for k, cam in enumerate(CAMERAS):
        v = obs_after[f"{cam}_rgb"]
        assert v.shape == (1, 1, 3, 256, 256) and v.dtype == torch.uint8 and v.is_cuda
        l, t, r, b = CROP_ORDER[k]
        assert np.array_equal(v[0, 0].permute(1, 2, 0).cpu().numpy(), tout[t:b, l:r]), cam
    assert obs_after["low_dim_state"].shape == (1, 1, 8) and obs_after["lang_tokens"].shape == (1, 1, 1, 77)
    assert actions.shape == (20, 8) and actions.dtype == np.float32 and np.isfinite(actions).all()
    # a fresh generator with the same seed reproduces the step bit for bit; the generated views differ from the input views
    gen2 = [torch.Generator(device="cuda").manual_seed(2)]
    actions2, _, _, tiled_out2 = harness.control_step(dagent, cagent, obs, "open the box", CAMERAS, 1, gen2, 5, 0.0, "cuda")
    assert np.array_equal(actions, actions2) and np.array_equal(np.asarray(tiled_out2[0]), tout)
    assert not np.array_equal(tout, tin)
